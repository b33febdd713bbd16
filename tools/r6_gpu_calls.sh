#!/bin/bash
# Round-6 GPU calls, one recipe each:   gpurun --timeout N -- 'bash tools/r6_gpu_calls.sh <recipe>'
# Every recipe writes under gpurun_out/r6_<recipe>/; what is judged is copied into profiles/r06/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; RECIPE=${1:-help}; O=$R/gpurun_out/r6_$RECIPE; mkdir -p $O; cd $R
export TMPDIR=/tmp LCC_PARITY_OUT=$O
QUIET="--cpu-baseline off --parity off --share8 off --live2fps off --more-configs off"
step() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict($1, tokens_per_s=d['value'], ms_per_replay=d['ms_per_step'], decode_step_us=r['decode_step']['avg_step_us'], us_per_layer=r['decode_step']['us_per_layer'], gate_up_us=r['avg_launch_us'])))"; }
case $RECIPE in
tier)          # the whole GPU tier as the driver runs it (pytest.ini adds -rs) + smoke()
  ( time python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40 ) > $O/pytest_gpu_tail.txt 2>&1
  ( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke_tail.txt 2>&1
  tail -45 $O/pytest_gpu_tail.txt; cat $O/smoke_tail.txt ;;
bench)         # the driver's default line + rocprofv3 kernel stats / step breakdown of a short run of the same command
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err ) 2> $O/bench_time.txt      # the driver's own command line (BENCH_rNN.json)
  cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $R/bench.py --steps 1 --warmup 1 --no-prefetch $QUIET > $O/bench_under_rocprof.json 2> $O/rocprof.err
  cd $R; cp $O/prof/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
  python tools/trace_breakdown.py $(ls $O/prof/*kernel_trace.csv | head -1) > $O/step_breakdown_1streams_noprefetch.json 2> $O/breakdown.err
  rm -rf $O/prof; tail -c 600 $O/bench_default_n1.json; cat $O/bench_time.txt; head -8 $O/bench_kernel_stats.csv ;;
pmc)           # T2 evidence: MfmaUtil of all four LLM GEMMs at M = 3088 / 386 -> profiles/roofline_traffic.json (copy it back!)
  cd /tmp; timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/pmc -o g -- python $R/tools/r6_pmc_llm_gemms.py $O/pmc/manifest.json > $O/pmc.log 2>&1
  cd $R; python tools/r6_summarize_llm_gemm_pmc.py $O/pmc > $O/llm_gemm_mfma_util.json 2> $O/summ.err
  cp profiles/roofline_traffic.json $O/roofline_traffic.json; find $O/pmc -name "*.csv" -size +2000k -delete
  python -c "import json; d=json.load(open('$O/llm_gemm_mfma_util.json')); print({k: v['mfma_util_pct'] for k, v in d['per_gemm'].items()}, d['per_layer'])"; tail -3 $O/summ.err ;;
pmc_gemv)      # HBM bytes per launch of the dominant kernel (FETCH_SIZE / WRITE_SIZE in separate passes) -> roofline_traffic.json
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p/pmc_$C -o gemv -- python $R/tools/pmc_target.py > $O/pmc_$C.log 2>&1; done
  cd $R; python tools/summarize_pmc.py $O/p > $O/roofline_traffic_gemv.json 2> $O/summ.err; find $O/p -name "*.csv" -size +2000k -delete
  python -c "import json; d=json.load(open('$O/roofline_traffic_gemv.json')); print(d['gemv_gate_up_hbm_bytes_per_launch'], d['algorithmic_bytes_per_launch'], d['kernel_source_sha16'])" ;;
attn_map)      # XCD-chunked block order / (row, head) pair packing of the 32x32x16 attention kernels, A/B/A/B
  for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do set -- $cfg
    LCC_ATTN32_XCD=$1 LCC_ATTN32_PACK=$2 python tools/bench_attn.py --only32 2>/dev/null | sed "s/^{/{\"xcd\": $1, \"pack\": $2, /" >> $O/attn_map_ab.jsonl
    LCC_ATTN32_XCD=$1 python tools/r5_tower.py "xcd$1" >> $O/tower_xcd_ab.jsonl 2>/dev/null; done
  cat $O/attn_map_ab.jsonl $O/tower_xcd_ab.jsonl ;;
attn_rowmajor) # RECORD of a removed variant (LCC_ATTN32_ROWMAJOR no longer exists; attn32.hip keeps the result): row-major (coalesced LDS-DMA)
               # vs fragment-order staging of the 32x32x16 attention kernels: same bits first, then A/B/A/B
  for m in 0 3; do LCC_ATTN32_ROWMAJOR=$m timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit_fused.py -q -x -k "attn or vit" 2>&1 | tail -2 | sed "s/^/rowmajor=$m: /" >> $O/attn_rowmajor_tests.txt; done
  for m in 0 3 0 3; do
    LCC_ATTN32_ROWMAJOR=$m python tools/bench_attn.py --only32 2>/dev/null | sed "s/^{/{\"rowmajor\": $m, /" >> $O/attn_rowmajor_ab.jsonl
    LCC_ATTN32_ROWMAJOR=$m python tools/r5_tower.py "rowmajor$m" >> $O/tower_rowmajor_ab.jsonl 2>/dev/null; done
  cat $O/attn_rowmajor_tests.txt $O/attn_rowmajor_ab.jsonl $O/tower_rowmajor_ab.jsonl ;;
attn_vgprform) # decode attention after __launch_bounds__(64, 2) (VGPR-form MFMAs, no v_accvgpr copies in the key loop): correctness, the cold-cache
               # split sweep of profiles/r06/attn_decode_split_sweep.jsonl again, and the decode step of the replay
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_decode_v2.py -q -x -k "attn or decode" 2>&1 | tail -2 > $O/tests.txt
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o s -- python $R/tools/r6_attn_decode_splits.py > $O/sweep.log 2>&1
  cd $R; python tools/r6_attn_decode_splits.py --reduce $(ls $O/tr/*kernel_trace.csv | head -1) > $O/attn_decode_split_sweep_vgprform.jsonl; rm -rf $O/tr
  for i in 1 2; do python bench.py --steps 3 --warmup 1 --no-prefetch $QUIET 2>/dev/null | step "vgpr_form=1" >> $O/decode_step_vgprform.jsonl; done
  python bench.py --steps 2 --warmup 1 --cpu-baseline off --parity off --share8 on --live2fps off --more-configs off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(vgpr_form=1, tokens_per_s=d['value'], configs2_share=d['configs2_share'])))" >> $O/share8_vgprform.jsonl
  cat $O/tests.txt $O/attn_decode_split_sweep_vgprform.jsonl $O/decode_step_vgprform.jsonl; cut -c1-900 $O/share8_vgprform.jsonl ;;
attn_tall)     # query tiles of 8 x 32 / G rows (36 at 7 heads per KV head: the pair packing fills all eight waves) vs 32-row tiles
  timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_baseline_configs.py tests/test_gpu_e2e.py tests/test_gpu_golden.py tests/test_gpu_torch_ops.py -q -x 2>&1 | tail -3 > $O/tests.txt
  for i in 1 2; do python tools/bench_attn.py --only32 2>/dev/null >> $O/attn_tall_tiles_ab.jsonl; done
  for r in 32 0 32 0; do LCC_ATTN32_TILE_ROWS=$r python bench.py --steps 3 --warmup 1 --no-prefetch $QUIET 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict(tile_rows_env=$r, tokens_per_s=d['value'], ms_per_replay=d['ms_per_step'], decode_step_us=r['decode_step']['avg_step_us'], prefill=r.get('chunk_prefill'))))" >> $O/replay_tall_tiles_ab.jsonl; done
  cat $O/tests.txt $O/attn_tall_tiles_ab.jsonl; cut -c1-400 $O/replay_tall_tiles_ab.jsonl ;;
tower4)        # tower attention after the AGPR-copy fixes: default routing vs the 4-wave 32x32x16 kernel for every grid (LCC_VIT32_MIN_BLOCKS4=0)
  timeout 900 python -m pytest tests/test_gpu_vit_fused.py tests/test_gpu_ops.py -q -x -k "vit or tower" 2>&1 | tail -2 > $O/tests.txt
  LCC_VIT32_MIN_BLOCKS4=0 timeout 900 python -m pytest tests/test_gpu_vit_fused.py tests/test_gpu_ops.py -q -x -k "vit or tower" 2>&1 | tail -2 | sed "s/^/min_blocks4=0: /" >> $O/tests.txt
  for m in default 0 default 0; do
    if [ $m = default ]; then python tools/r5_tower.py "routing=default" >> $O/tower_4wave_vgprform_ab.jsonl 2>/dev/null
    else LCC_VIT32_MIN_BLOCKS4=$m python tools/r5_tower.py "min_blocks4=$m" >> $O/tower_4wave_vgprform_ab.jsonl 2>/dev/null; fi; done
  cat $O/tests.txt $O/tower_4wave_vgprform_ab.jsonl ;;
tower_prof)    # kernel stats of the one-chunk tower (2 frames 392x728 = 1,456 patches... x 2 temporal = one slice)
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $R/tools/r5_tower.py prof 1 > $O/tower.log 2>&1
  cd $R; cp $O/prof/*kernel_stats.csv $O/tower_one_chunk_kernel_stats.csv; rm -rf $O/prof; head -30 $O/tower_one_chunk_kernel_stats.csv | cut -c1-200; cat $O/tower.log | tail -2 ;;
pingpong)      # RECORD of a removed variant (LCC_ATTN32_PINGPONG no longer exists; attn32.hip keeps the result): 8-wave attention kernels: matrix phase / vector phase in opposite order on the two waves of a SIMD (LCC_ATTN32_PINGPONG bit 0 LLM, bit 1 tower)
  for m in 0 3; do LCC_ATTN32_PINGPONG=$m timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit_fused.py tests/test_gpu_baseline_configs.py -q -x -k "attn or vit or attention" 2>&1 | tail -1 | sed "s/^/pingpong=$m: /" >> $O/tests.txt; done
  for m in 0 1 0 1; do LCC_ATTN32_PINGPONG=$m python tools/bench_attn.py --only32 2>/dev/null | sed "s/^{/{\"pingpong\": $m, /" >> $O/attn_pingpong_ab.jsonl; done
  for m in 0 2 0 2; do LCC_ATTN32_PINGPONG=$m python tools/r5_tower.py "pingpong=$m" 8 >> $O/tower_pingpong_ab.jsonl 2>/dev/null; done
  cat $O/tests.txt; grep -v "nsplit\": 5" $O/attn_pingpong_ab.jsonl | cut -c1-200; cat $O/tower_pingpong_ab.jsonl ;;
vit_lsum)      # tower attention: softmax denominator from the P.V MFMAs (ones rows 80..95 of V^T) vs vector adds
  for m in 0 1; do LCC_VIT32_LSUM_MFMA=$m timeout 1200 python -m pytest tests/test_gpu_vit_fused.py tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_golden.py tests/test_gpu_layer_parity.py -q -x -k "not 72b" 2>&1 | tail -1 | sed "s/^/lsum_mfma=$m: /" >> $O/tests.txt; done
  for m in 0 1 0 1; do LCC_VIT32_LSUM_MFMA=$m python tools/r5_tower.py "lsum_mfma=$m" >> $O/tower_lsum_mfma_ab.jsonl 2>/dev/null; done
  cat $O/tests.txt $O/tower_lsum_mfma_ab.jsonl ;;
pmc_attn)      # MfmaUtil of the 32x32x16 attention kernels (LLM prefill shapes of tools/bench_attn.py --only32, then the tower at 8 / 1 chunks)
  cd /tmp; timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/p1 -o a -- python $R/tools/bench_attn.py --only32 > $O/bench_attn_under_pmc.jsonl 2>$O/p1.err
  timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/p2 -o t -- python $R/tools/r5_tower.py pmc > $O/tower_under_pmc.jsonl 2>$O/p2.err
  cd $R; python tools/r6_summarize_attn_pmc.py $O/p1 > $O/attn_mfma_util.jsonl; python tools/r6_summarize_attn_pmc.py $O/p2 >> $O/attn_mfma_util.jsonl
  rm -rf $O/p1 $O/p2; cat $O/attn_mfma_util.jsonl ;;
attn_direct)   # decode attention with the stream state by value (LCC_ATTN_DIRECT), whole replay without the tower prefetch, A/B/A/B
  for d in 0 1 0 1; do LCC_ATTN_DIRECT=$d python bench.py --steps 3 --warmup 1 --no-prefetch $QUIET 2>/dev/null | step "attn_direct=$d" >> $O/attn_direct_ab.jsonl; done
  cat $O/attn_direct_ab.jsonl ;;
attn_tps)      # key tiles per split of the decode attention with the split cap raised
  for cfg in "4 64" "2 128" "3 128" "4 64" "2 128" "3 128"; do set -- $cfg
    LCC_ATTN_TPS=$1 LCC_ATTN_MAXSPLIT=$2 python bench.py --steps 3 --warmup 1 --no-prefetch $QUIET 2>/dev/null | step "tps=$1, maxsplit=$2" >> $O/attn_tps_maxsplit_ab.jsonl; done
  cat $O/attn_tps_maxsplit_ab.jsonl ;;
tall)          # one-chunk GEMMs: tall kernel schedule A/B
  for s in 0 1 2 0 1 2; do LCC_TALL_SCHED=$s python tools/r6_tall_sched.py "tall_sched$s" 2>/dev/null >> $O/m386_gemms_ab.jsonl; done; cat $O/m386_gemms_ab.jsonl ;;
rccl)          # can RCCL run two ranks on the box's one GPU?
  timeout 150 python tools/r6_rccl_same_device_probe.py | tee $O/rccl_probe.json ;;
*) echo "recipes: tier bench pmc pmc_gemv attn_map attn_rowmajor attn_vgprform attn_tall tower4 tower_prof pingpong vit_lsum pmc_attn attn_direct attn_tps tall rccl" ;;
esac
