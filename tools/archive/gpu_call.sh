#!/bin/bash
# One parametrised driver for the GPU calls of a round (replaces the per-call scripts of round 3):
#   gpurun --timeout N -- 'bash tools/gpu_call.sh <recipe> [args]'
# Every step runs under its own `timeout`, writes under gpurun_out/<recipe>/ and never stops the recipe on failure.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
RECIPE=${1:-help}; shift || true
O=$R/gpurun_out/$RECIPE
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python bench.py --cpu-baseline off --parity off --live2fps off --more-configs off"
val() { grep -o "\"$2\": [0-9.]*" $1 | head -1 | cut -d' ' -f2; }

case $RECIPE in
golden)      # the fixture-based 7B / 2B tests (no HF forward on the box)
  timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q --timeout 600 "$@" > $O/golden.log 2>&1; tail -n 15 $O/golden.log ;;
pmc_l2)      # L2 / TCP / TA counters of the 8-wave GEMM's DMA ring
  bash tools/pmc_gemm_l2.sh $O > $O/pmc_l2.log 2>&1; tail -n 70 $O/pmc_l2.log ;;
t192)        # the 192-row tile of the 8-wave GEMM: bit-identity with the 256-row tile, kernel tests, micro-benchmark (gate/up + down_proj)
  for V in 3 13; do timeout 200 python tools/gemm_checksum.py $V > $O/sum_v$V.txt 2>$O/sum_v$V.err; done
  cmp $O/sum_v3.txt $O/sum_v13.txt && echo "variant 13: CHECKSUMS IDENTICAL to variant 3" || { echo "variant 13 DIFFERS"; paste $O/sum_v3.txt $O/sum_v13.txt; tail -n 3 $O/sum_v13.err; }
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -k "gemm_tiled and 13" > $O/gemm_tests.log 2>&1; tail -n 4 $O/gemm_tests.log
  for V in 3 13 2; do timeout 120 python tools/bench_gemm_diag.py $V down 2>/dev/null | grep '^{' | tee -a $O/gemm_192_bench.txt; timeout 120 python tools/bench_gemm_diag.py $V 2>/dev/null | grep '^{' | tee -a $O/gemm_192_bench.txt; done ;;
mid)         # weight-streaming GEMV for 17-64 rows: kernel tests, then the 32-stream bench with the new and the round-3 routing
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -k "17_to_64" > $O/mid_tests.log 2>&1; tail -n 6 $O/mid_tests.log
  timeout 600 python -m pytest tests/test_gpu_facade.py -m gpu -q --timeout 500 -k "20_streams or batch" > $O/facade.log 2>&1; tail -n 4 $O/facade.log
  ( timeout 500 $B --steps 1 --warmup 1 --streams-per-gpu 32 --share8 off ) > $O/bench_32s_mid.log 2>&1; tail -n 1 $O/bench_32s_mid.log | cut -c1-1600 ;;
trace)       # kernel-trace breakdown of a multi-stream replay: bash tools/gpu_call.sh trace <streams>
  N=${1:-8}; shift || true; X="$*"; cd /tmp      # extra bench flags after the stream count, e.g. trace 32 --skinny-rows 16
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t$N -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-prefetch --streams-per-gpu $N --cpu-baseline off --parity off --share8 off --live2fps off --more-configs off $X > $O/bench_${N}s_under_rocprof.json 2> $O/trace_$N.err
  T=$(find $O/t$N -name '*kernel_trace.csv' | head -1); python $R/tools/trace_breakdown.py $T 28 > $O/step_breakdown_${N}streams_noprefetch.json 2>> $O/trace_$N.err; rm -rf $O/t$N
  [ -n "$X" ] && cp $O/step_breakdown_${N}streams_noprefetch.json "$O/step_breakdown_${N}streams_noprefetch_$(echo $X | tr -d ' -').json"
  python - <<PY
import json
d = json.load(open("$O/step_breakdown_${N}streams_noprefetch.json"))
print("decode step", d["avg_step_us"], "us; kernels", d["avg_kernel_time_per_step_us"], "gaps", d["avg_gap_per_step_us"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["us_per_step"])[:12]:
    print("  %-56s calls %5.1f avg %7.2f us  per step %8.1f" % (k[:56], v["calls_per_step"], v["avg_us"], v["us_per_step"]))
PY
  cd $R ;;
pmc_gemv)    # counters of the weight-streaming gate/up GEMV at M = 8 / 32 / 64 (one pass per group)
  cd /tmp; i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum" "FETCH_SIZE"; do
    i=$((i+1)); timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o gemv -- python $R/tools/pmc_target.py --gemv-rows > $O/p$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -n 1 $O/p$i.log)"
    find $O/p$i -name '*kernel_trace.csv' -delete
  done
  python - <<PY > $O/gemv_rows_counters.json
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "gemv_skinny_kernel" in kn:
            key = kn.split("(")[0].replace("void lcc::", "")[:60]
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[key]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}, indent=1))
PY
  cat $O/gemv_rows_counters.json; rm -rf $O/p*/; cd $R ;;
pmc_attn)    # counters of the decode attention (attn_decode_kernel + combine) at L = 6.5k, one stream: one pass per group
  cd /tmp; i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    i=$((i+1)); timeout 100 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o attn -- python $R/tools/pmc_target.py --attn-decode > $O/p$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -n 1 $O/p$i.log)"
    find $O/p$i -name '*kernel_trace.csv' -delete
  done
  python - <<PY > $O/attn_decode_counters.json
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "attn_decode" in kn:
            key = kn.split("(")[0].replace("void lcc::", "").replace("lcc::", "")[:60]
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[key]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print(json.dumps({k: {c: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for c, v in cs.items()} for k, cs in res.items()}, indent=1))
PY
  cat $O/attn_decode_counters.json; rm -rf $O/p*/; cd $R ;;
b72)         # BASELINE.json configs[4]: Qwen2-VL-72B shapes, fp8 weights, the configs[1] protocol (60 frames), one stream on one GPU
  ( timeout 900 $B --config qwen2vl-72b --weights fp8 --steps 1 --warmup 1 --share8 off ) > $O/bench_72b_fp8.log 2>$O/bench_72b_fp8.err; tail -n 1 $O/bench_72b_fp8.log | cut -c1-2200; tail -n 3 $O/bench_72b_fp8.err ;;
ab)          # A/B of an environment switch on the multi-stream bench: bash tools/gpu_call.sh ab <streams> VAR=a VAR=b ...
  N=$1; shift
  for KV in "$@"; do ( env $KV timeout 500 $B --steps 2 --warmup 1 --streams-per-gpu $N --share8 off ) > $O/bench_${N}s_$KV.log 2>&1; echo "== $N streams $KV: $(val $O/bench_${N}s_$KV.log value) tok/s, $(val $O/bench_${N}s_$KV.log frames_per_s) frames/s"; tail -n 2 $O/bench_${N}s_$KV.log | grep -v '^{' | cut -c1-300; done ;;
vitq)        # q|k|v projection of the vision tower with RoPE + V transpose in its epilogue: bit-identity vs the separate launches, then the
             # tower / HF parity tests that run through it, then the 8-stream bench A/B
  timeout 600 python -m pytest tests/test_gpu_vit_fused.py -m gpu -q --timeout 500 > $O/vitq.log 2>&1; tail -n 12 $O/vitq.log; cp gpurun_out/parity_report.json $O/parity_vitq.json; grep -A12 vit_tower_fused $O/parity_vitq.json | head -40
  timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --timeout 500 -k "vit_rope or vit_features or prefetch or teacher" > $O/vitq2.log 2>&1; tail -n 4 $O/vitq2.log
  if [ "${1:-}" = "bench" ]; then
    for KV in LCC_VIT_FUSED_QKV=0 LCC_VIT_FUSED_QKV=1 LCC_VIT_FUSED_QKV=0 LCC_VIT_FUSED_QKV=1; do ( env $KV timeout 500 $B --steps 2 --warmup 1 --streams-per-gpu 8 --share8 off ) > $O/bench_8s_$KV.log 2>&1; echo "== 8 streams $KV: $(val $O/bench_8s_$KV.log value) tok/s"; done
    for KV in LCC_VIT_FUSED_QKV=0 LCC_VIT_FUSED_QKV=1; do ( env $KV timeout 500 $B --steps 2 --warmup 1 --share8 off ) > $O/bench_1s_$KV.log 2>&1; echo "== 1 stream $KV: $(val $O/bench_1s_$KV.log value) tok/s"; done
  fi ;;
vitprof)     # kernel stats of the vision tower alone, fused and separate q|k|v epilogue: bash tools/gpu_call.sh vitprof <streams>
  N=${1:-8}; cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$N -o tower -- python $R/tools/vit_tower_target.py $N > $O/tower_$N.log 2> $O/tower_$N.err
  S=$(find $O/s$N -name '*kernel_stats.csv' | head -1); cp $S $O/vit_tower_${N}streams_kernel_stats.csv; rm -rf $O/s$N
  python - <<PY
import csv
rows = list(csv.DictReader(open("$O/vit_tower_${N}streams_kernel_stats.csv")))
for r in rows[:16]:
    print("%-110s calls %5s avg %9.2f us total %9.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  cd $R ;;
final)       # end-of-round validation of the final tree, most important first: GPU tier, smoke, the driver's bench line, kernel stats of the bench
  ( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/test_full.log 2>&1; echo "tests rc=$?" >> $O/test_full.log; tail -n 8 $O/test_full.log
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
  ( time timeout 1200 python bench.py ) > $O/bench_default.log 2>$O/bench_default.err; grep '^{' $O/bench_default.log | tail -n 1 | cut -c1-2500
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 1 --warmup 1 --cpu-baseline off --parity off --share8 off > $O/bench_under_rocprof.json 2> $O/stats.err
  S=$(find $O/stats -name '*kernel_stats.csv' | head -1); [ -n "$S" ] && cp $S $O/bench_kernel_stats.csv; rm -rf $O/stats; head -n 8 $O/bench_kernel_stats.csv | cut -c1-200; cd $R ;;
variants)    # the bench variants of DESIGN section 6 that the vision-tower change moves: 8 and 32 streams, one stream without the prefetch
  ( $B --steps 1 --warmup 1 --streams-per-gpu 8 --share8 off ) > $O/bench_8streams.log 2>&1; echo "8 streams: $(val $O/bench_8streams.log value)"
  ( $B --steps 1 --warmup 0 --streams-per-gpu 32 --share8 off ) > $O/bench_32streams.log 2>&1; echo "32 streams: $(val $O/bench_32streams.log value)"
  ( $B --steps 2 --warmup 1 --no-prefetch --share8 off ) > $O/bench_noprefetch.log 2>&1; echo "1 stream, no prefetch: $(val $O/bench_noprefetch.log value)" ;;
r5a)         # round 5, call 1: LDS-DMA through inline asm (counted lgkmcnt ladders) vs the round-4 build; raster / L2 run-ahead knobs
  NEW=$R/livecc_amd/_C/liblivecc_amd.so; OLD=$R/livecc_amd/_C_r4/liblivecc_amd.so
  LCC_LIB_PATH=$OLD timeout 200 python tools/gemm_checksum.py > $O/sum_r4.txt 2>$O/sum_r4.err
  for KV in "X=0" "LCC_GEMM_RASTER=8" "LCC_GEMM_L2PF=2" "LCC_GEMM_L2PF=3" "LCC_GEMM_RASTER=4 LCC_GEMM_L2PF=2"; do
    T=$(echo $KV | tr -d ' ='); env $KV timeout 200 python tools/gemm_checksum.py > $O/sum_$T.txt 2>$O/sum_$T.err
    cmp -s $O/sum_r4.txt $O/sum_$T.txt && echo "checksums [$KV]: IDENTICAL to the round-4 build" || { echo "checksums [$KV] DIFFER"; paste $O/sum_r4.txt $O/sum_$T.txt; tail -n 3 $O/sum_$T.err; }
  done
  LCC_LIB_PATH=$OLD timeout 300 python tools/r5_bench_gemm.py r4 2>$O/g_r4.err | tee -a $O/gemm_ab.jsonl | cut -c1-200
  for KV in "X=0" "LCC_GEMM_RASTER=8" "LCC_GEMM_RASTER=4" "LCC_GEMM_L2PF=2" "LCC_GEMM_L2PF=3" "LCC_GEMM_RASTER=8 LCC_GEMM_L2PF=2" "LCC_GEMM_RASTER=4 LCC_GEMM_L2PF=3"; do
    T=$(echo $KV | tr -d ' ='); env $KV timeout 300 python tools/r5_bench_gemm.py "$T" 2>$O/g_$T.err | tee -a $O/gemm_ab.jsonl | cut -c1-200
  done
  LCC_LIB_PATH=$OLD timeout 300 python tools/r5_bench_gemm.py r4_again gate_up_M3088,down_M3088,vit_fc1_P11648 2>>$O/g_r4.err | tee -a $O/gemm_ab.jsonl | cut -c1-200
  LCC_LIB_PATH=$OLD timeout 300 python tools/bench_attn.py --quick 2>$O/attn_r4.err | sed 's/^/r4 /' | tee -a $O/attn_ab.txt | grep '"variant": 3' | cut -c1-220
  timeout 300 python tools/bench_attn.py --quick 2>$O/attn_new.err | sed 's/^/new /' | tee -a $O/attn_ab.txt | grep '"variant": 3' | cut -c1-220
  LCC_LIB_PATH=$OLD timeout 300 python tools/r5_tower.py r4 2>$O/tower_r4.err | tee -a $O/tower_ab.jsonl
  timeout 300 python tools/r5_tower.py new 2>$O/tower_new.err | tee -a $O/tower_ab.jsonl
  timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit_fused.py -m gpu -q -x --timeout 500 > $O/ops_tests.log 2>&1; tail -n 5 $O/ops_tests.log ;;
r5b)         # round 5, call 2: SQ counters of the MFMA-bound kernels (where do the wave cycles go?) + the new bench legs end to end
  cd /tmp; i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
    i=$((i+1)); timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o k -- python $R/tools/r5_pmc_target.py > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -n 2 $O/p$i.log)"
    find $O/p$i -name '*kernel_trace.csv' -delete
  done
  python - <<PY > $O/sq_counters.json
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if any(t in kn for t in ("gemm_big_kernel", "attn_gqa32", "attn_vit32", "layernorm_kernel", "gemm_glds")):
            key = kn.split("(")[0].replace("void lcc::", "").replace("lcc::", "")[:60] + " grid=" + r["Grid_Size"]
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[key]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, launches=len(cs["duration_us"])) for k, cs in res.items()}
print(json.dumps(out, indent=1))
PY
  python - <<PY
import json
d = json.load(open("$O/sq_counters.json"))
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("duration_us", 0) * kv[1].get("launches", 0))[:9]:
    w = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-70s %8.1f us x%3d | of wave cycles: wait_any %.2f wait_inst %.2f active_any %.2f valu %.2f lds %.2f vmem %.2f | mfma_busy/busy %.2f | vmem_inst_cycles/wave %.2f ta_addr_full %.3f lds_cmd_full %.3f bank_conf/idx %.2f | insts valu %.0f mfma %.0f lds %.0f vmem %.0f" % (
        k[:70], c.get("duration_us", 0), c.get("launches", 0), c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w,
        c.get("SQ_ACTIVE_INST_VALU", 0) / w, c.get("SQ_ACTIVE_INST_LDS", 0) / w, c.get("SQ_ACTIVE_INST_VMEM", 0) / w,
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1), c.get("SQ_INST_CYCLES_VMEM", 0) / w, c.get("SQ_VMEM_TA_ADDR_FIFO_FULL", 0) / w,
        c.get("SQ_LDS_CMD_FIFO_FULL", 0) / w, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_MFMA", 0),
        c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0)))
PY
  rm -rf $O/p*/; cd $R
  ( time timeout 1500 python bench.py --steps 1 --warmup 1 --cpu-baseline off --parity off ) > $O/bench_legs.log 2>$O/bench_legs.err; tail -n 4 $O/bench_legs.err
  python - <<PY
import json
l = [x for x in open("$O/bench_legs.log") if x.startswith("{")]
d = json.loads(l[-1]) if l else {}
print("value", d.get("value"), "share", (d.get("configs2_share") or {}).get("value"), (d.get("configs2_share") or {}).get("error"))
print("live2fps", json.dumps(d.get("live2fps"))[:1500])
print("configs3", json.dumps(d.get("configs3_oneshot480"))[:1200])
print("configs4", json.dumps(d.get("configs4_72b_fp8"))[:1200])
PY
  ;;
r5c)         # round 5, call 3: variable-height GEMM tiles + raster default + attention cross-region prefetch: bit-identity, tests, A/B
  OLD=$R/livecc_amd/_C_r4/liblivecc_amd.so
  LCC_LIB_PATH=$OLD timeout 200 python tools/gemm_checksum.py > $O/sum_r4.txt 2>$O/sum_r4.err
  for KV in "X=0" "LCC_GEMM_VH=0 LCC_GEMM_RASTER=0"; do
    T=$(echo $KV | tr -d ' ='); env $KV timeout 200 python tools/gemm_checksum.py > $O/sum_$T.txt 2>$O/sum_$T.err
    cmp -s $O/sum_r4.txt $O/sum_$T.txt && echo "checksums [$KV]: IDENTICAL to the round-4 build" || { echo "checksums [$KV] DIFFER"; paste $O/sum_r4.txt $O/sum_$T.txt; tail -n 3 $O/sum_$T.err; }
  done
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit_fused.py -m gpu -q -x --timeout 800 > $O/ops_tests.log 2>&1; tail -n 5 $O/ops_tests.log
  SH=gate_up_M3088,gate_up_M1131,gate_up_M24058,down_M3088,qkv_M3088,o_M3088,vit_fc1_P11648,vit_fc2_P11648,vit_qkv_P11648,vit_proj_P11648
  LCC_LIB_PATH=$OLD timeout 300 python tools/r5_bench_gemm.py r4 $SH 2>$O/g_r4.err | tee -a $O/gemm_ab.jsonl | cut -c1-160
  for KV in "X=0" "LCC_GEMM_VH=0" "LCC_GEMM_RASTER=0" "LCC_GEMM_RASTER=8" "X=1"; do
    T=$(echo $KV | tr -d ' ='); env $KV timeout 300 python tools/r5_bench_gemm.py "$T" $SH 2>$O/g_$T.err | tee -a $O/gemm_ab.jsonl | cut -c1-160
  done
  for P in 0 1 0 1; do LCC_ATTN32_PIPE=$P timeout 300 python tools/bench_attn.py --quick 2>$O/attn_$P.err | grep '"variant": 3' | grep '"nsplit": 1,' | sed "s/^/pipe$P /" | tee -a $O/attn_ab.txt | cut -c1-200; done
  for P in 0 1 0 1; do LCC_ATTN32_PIPE=$P timeout 300 python tools/r5_tower.py pipe$P 2>$O/tower_$P.err | tee -a $O/tower_ab.jsonl; done
  for L in r4 new r4 new; do
    if [ $L = r4 ]; then export LCC_LIB_PATH=$OLD; else unset LCC_LIB_PATH; fi
    ( timeout 500 $B --steps 2 --warmup 1 --streams-per-gpu 8 --share8 off ) > $O/bench_8s_$L.log 2>&1; echo "== 8 streams $L: $(val $O/bench_8s_$L.log value) tok/s"
  done
  for L in r4 new; do
    if [ $L = r4 ]; then export LCC_LIB_PATH=$OLD; else unset LCC_LIB_PATH; fi
    ( timeout 500 $B --steps 2 --warmup 1 --share8 off ) > $O/bench_1s_$L.log 2>&1; echo "== 1 stream $L: $(val $O/bench_1s_$L.log value) tok/s"
  done
  unset LCC_LIB_PATH ;;
r5d)         # round 5, call 4: the prefetched vision tower under a CU budget (persistent walks): bit-identity, then the bench at 1 / 8 streams per cap
  timeout 600 python -m pytest tests/test_gpu_vit_fused.py -m gpu -q -x --timeout 500 > $O/cap_tests.log 2>&1; tail -n 4 $O/cap_tests.log
  grep -A3 "vit_tower_grid_cap" gpurun_out/parity_report.json | grep -E "grid_cap|tower_ms" | paste - - | cut -c1-160
  for CAP in 0 64 96 128 160 0; do
    ( LCC_VIT_PREFETCH_CAP=$CAP timeout 500 $B --steps 2 --warmup 1 --streams-per-gpu 8 --share8 off ) > $O/bench_8s_cap$CAP.log 2>&1
    echo "== 8 streams, prefetch cap $CAP: $(val $O/bench_8s_cap$CAP.log value) tok/s  step $(grep -o '"avg_step_us": [0-9.]*' $O/bench_8s_cap$CAP.log | head -1)"
  done
  for CAP in 0 32 48 64 96 0; do
    ( LCC_VIT_PREFETCH_CAP=$CAP timeout 500 $B --steps 2 --warmup 1 --share8 off ) > $O/bench_1s_cap$CAP.log 2>&1
    echo "== 1 stream, prefetch cap $CAP: $(val $O/bench_1s_cap$CAP.log value) tok/s  step $(grep -o '"avg_step_us": [0-9.]*' $O/bench_1s_cap$CAP.log | head -1)"
  done ;;
r5e)         # round 5, call 5: the long-stream fixture with its new fp32 leg + the 72B full-depth test again; GEMV-under-GEMM overlap probe
  timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_layer_parity.py -m gpu -q --timeout 500 -k "long480_stream or full_depth" > $O/fixtures.log 2>&1; tail -n 3 $O/fixtures.log
  grep -A12 '"livecc7b_long480_stream_vs_committed_golden"' gpurun_out/parity_report.json | tr -d '\n' | cut -c1-700; echo
  timeout 300 python tools/r5_overlap_probe.py 2>$O/overlap.err | tee $O/overlap_probe.jsonl ;;
tests)       # the whole GPU tier, serially, as the driver runs it
  timeout ${1:-1500} python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -n 25 $O/tests.log ;;
bench)       # the driver's default line
  timeout 1700 python bench.py "$@" > $O/bench.log 2>$O/bench.err; tail -n 3 $O/bench.log | cut -c1-3000; tail -n 5 $O/bench.err ;;
r5f)         # round 5, call 6 (RECORD ONLY: the variant measured null and was removed, the switch no longer exists): the k-tile barrier 3 steps before the end of the tile (LCC_GEMM_EARLY_BARRIER=1): bit-identity (3 runs), A/B
  for E in 0 1 1 1; do LCC_GEMM_EARLY_BARRIER=$E timeout 200 python tools/gemm_checksum.py "$@" > $O/sum_e$E.txt 2>$O/sum_e$E.err; cmp $O/sum_e0.txt $O/sum_e$E.txt && echo "early barrier $E: CHECKSUMS IDENTICAL" || { echo "early barrier $E DIFFERS"; paste $O/sum_e0.txt $O/sum_e$E.txt; tail -n 3 $O/sum_e$E.err; }; done
  for E in 0 1 0 1; do LCC_GEMM_EARLY_BARRIER=$E timeout 300 python tools/r5_bench_gemm.py early$E gate_up_M3088,gate_up_M1131,down_M3088,vit_fc1_P11648,vit_fc2_P11648,vit_qkv_P11648 2>/dev/null | grep '^{' | tee -a $O/gemm_early_barrier_ab.jsonl | cut -c1-200; done ;;
r5g)         # round 5, call 7: what bounds the 8-wave GEMM now?  no-DMA / no-MFMA diagnostics of gemm_big_kernel<256> + the row-stride probe
  for D in 0 2 3 0 2 3; do LCC_GEMM_VH=0 LCC_GEMM_DIAG=$D timeout 120 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | tee -a $O/gemm_diag.jsonl; done
  timeout 120 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | sed 's/^/vh /' | tee -a $O/gemm_diag.jsonl
  timeout 300 python tools/r5_lda_probe.py "$@" 2>$O/lda.err | tee -a $O/lda_probe.jsonl; tail -n 3 $O/lda.err ;;
r5h)         # round 5, call 8: the row-stride probe again, A/B/A/B order: bash tools/gpu_call.sh r5h <pads> <shapes>
  timeout 600 python tools/r5_lda_probe.py "$@" 2>$O/lda.err | tee -a $O/lda_probe_abab.jsonl; tail -n 2 $O/lda.err ;;
r5i)         # round 5, call 9 (RECORD ONLY: 5-7 % slower, kernel removed again, LCC_GEMM_RING no longer exists): the 4-stage half-tile ring (gemm_big4_kernel) on the asm LDS-DMA vs gemm_big_kernel<256>: bit-identity, sustained A/B
  for RING in 2 4 4; do LCC_GEMM_VH=0 LCC_GEMM_RING=$RING timeout 200 python tools/gemm_checksum.py > $O/sum_ring$RING.txt 2>$O/sum_ring$RING.err; cmp $O/sum_ring2.txt $O/sum_ring$RING.txt && echo "ring $RING: CHECKSUMS IDENTICAL" || { echo "ring $RING DIFFERS"; paste $O/sum_ring2.txt $O/sum_ring$RING.txt; tail -n 3 $O/sum_ring$RING.err; }; done
  for RING in 2 4 2 4; do LCC_GEMM_VH=0 LCC_GEMM_RING=$RING timeout 120 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | sed "s/^/ring$RING /" | tee -a $O/gemm_ring4_asm_dma.txt; done ;;
r5j)         # round 5, call 10: random vs zero-filled operands through the same GEMM launches (is the kernel clock / power bound?)
  timeout 300 python tools/r5_power_probe.py 2>$O/power.err | tee $O/gemm_operand_toggling.jsonl; tail -n 2 $O/power.err ;;
r5k)         # round 5, call 11: shader clock + socket power under the GEMM on random vs zero-filled operands (and under the HBM-bound GEMV)
  ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40 > $O/hwmon_files.txt
  timeout 200 python tools/r5_clock_probe.py 2>$O/clock.err | tee $O/gemm_clock_power.jsonl; tail -n 2 $O/clock.err ;;
r5l)         # round 5, call 12: effective clock of the GEMM dispatches on random vs zero-filled operands: GRBM_GUI_ACTIVE cycles / duration
  cd /tmp; timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_clock -o clk -- python $R/tools/r5_power_probe.py > $O/power_under_pmc.jsonl 2>$O/pmc.err
  cd $R; python tools/r5_clock_from_pmc.py $O/pmc_clock | tee $O/gemm_effective_clock.jsonl; tail -n 2 $O/pmc.err; rm -rf $O/pmc_clock ;;
r5m)         # round 5, call 13: cache policy of the LDS-DMA instruction (needs the extra builds livecc_amd/_C_{nt,sc0,sc1}: make OUT=../_C_nt CXXFLAGS="... -DLCC_GLDS_POLICY='\" nt\"'"; skipped when absent): bit-identity + A/B
  for P in "" _nt _sc0 _sc1; do
    L=$R/livecc_amd/_C$P/liblivecc_amd.so; [ -f $L ] || continue
    LCC_LIB_PATH=$L timeout 200 python tools/gemm_checksum.py > $O/sum$P.txt 2>$O/sum$P.err; cmp $O/sum.txt $O/sum$P.txt > /dev/null && echo "policy[$P]: CHECKSUMS IDENTICAL" || echo "policy[$P] DIFFERS"
    LCC_LIB_PATH=$L timeout 120 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | sed "s/^/policy[$P] /" | tee -a $O/glds_policy_ab.txt
    LCC_LIB_PATH=$L LCC_GEMM_VH=0 LCC_GEMM_DIAG=3 timeout 120 python tools/bench_gemm_diag.py 2>/dev/null | grep '^{' | sed "s/^/policy[$P] dma-only /" | tee -a $O/glds_policy_ab.txt
    LCC_LIB_PATH=$L timeout 200 python tools/r5_bench_gemm.py "policy$P" gate_up_M386_tall,vit_fc2_P1456,vit_fc1_P1456,down_M3088 2>/dev/null | grep '^{' | tee -a $O/glds_policy_ab.txt | cut -c1-180
    LCC_LIB_PATH=$L timeout 200 python tools/r5_tower.py "policy$P" 2>/dev/null | grep '^{' | tee -a $O/glds_policy_ab.txt
  done ;;
r5n)         # round 5, call 14: the prefill attention kernel on random vs zero-filled q / K / V (clock-throttled like the GEMMs?)
  for Z in "" --zeros "" --zeros; do timeout 200 python tools/bench_attn.py --only32 $Z 2>/dev/null | grep '^{' | tee -a $O/attn_operand_toggling.jsonl | cut -c1-200; done ;;
r5o)         # round 5, call 15: a full-size (16.6 GB, sharded, HF 4.5x keys) safetensors checkpoint through weights.from_pretrained, bf16 and fp8 arenas
  df -h /tmp | tail -n 1 > $O/disk.txt
  timeout 900 python tools/full_size_loader_check.py 2>$O/loader.err | tee $O/full_size_loader_check.jsonl
  timeout 900 python tools/full_size_loader_check.py --fp8 2>>$O/loader.err | tee -a $O/full_size_loader_check.jsonl; tail -n 3 $O/loader.err ;;
r5p)         # round 5, call 16: soak of the live-paced server leg: N streams x 60 s of video (KV history to ~26k keys per stream), p99 per 10-s window
  timeout 400 python tools/r5_live_soak.py ${1:-48} ${2:-60} 2>$O/soak.err | tee -a $O/live2fps_soak.jsonl | cut -c1-900; tail -n 2 $O/soak.err ;;
r5q)         # round 5, call 17: the small variable-height tile class (128 / 144-row tiles, 6 / 6 / 4 splits at M = 386): tests, micro-benchmark, 1-stream A/B
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 500 -k "small_variable or auto_choice or variable_height" > $O/ops_tests.log 2>&1; tail -n 4 $O/ops_tests.log
  timeout 200 python tools/r5_bench_splitk.py 2>$O/splitk.err | tee $O/gemm_splitk_small_vh.jsonl | cut -c1-160; tail -n 2 $O/splitk.err
  for V in 0 1 0 1; do ( LCC_GEMM_VH_SMALL=$V timeout 300 $B --steps 2 --warmup 1 --share8 off ) > $O/bench_small$V.log 2>&1; echo "LCC_GEMM_VH_SMALL=$V: $(val $O/bench_small$V.log value) tok/s" | tee -a $O/bench_small_vh_ab.txt; done ;;
r5r)         # round 5, call 18: MfmaUtil of the M = 386 split-K projections on the small variable-height class (one PMC pass, reduced by summarize_pmc.py)
  mkdir -p $O/p; cd /tmp
  timeout 150 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/p/pmc_gemm_MfmaUtil -o gemm -- python $R/tools/pmc_target.py --gemm > $O/pmc.log 2>&1
  find $O/p -name '*kernel_trace.csv' -delete; cd $R
  python tools/summarize_pmc.py $O/p 2>/dev/null | python -c "import sys, json; d = json.load(sys.stdin); print(json.dumps(d['gemm_counters_mean_per_dispatch'], indent=1))" | tee $O/mfma_util_small_vh.json; rm -rf $O/p ;;
*) echo "recipes: golden pmc_l2 tests bench r5a..r5f (see the case statement)";;
esac
