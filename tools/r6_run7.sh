# round 6, GPU call 7: decode attention with the stream state by value (LCC_ATTN_DIRECT) -- A/B/A/B of the whole replay without the tower prefetch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g; mkdir -p $O
for d in 0 1 0 1; do
  LCC_ATTN_DIRECT=$d python bench.py --steps 3 --warmup 1 --no-prefetch --cpu-baseline off --parity off --share8 off --live2fps off --more-configs off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict(attn_direct=$d, tokens_per_s=d['value'], ms_per_replay=d['ms_per_step'], decode_step_us=r['decode_step']['avg_step_us'], us_per_layer=r['decode_step']['us_per_layer'], gate_up_us=r['avg_launch_us'])))" >> $O/attn_direct_ab.jsonl
done
python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -3
cat $O/attn_direct_ab.jsonl
