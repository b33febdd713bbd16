cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_fused8; mkdir -p $O
QUIET="--cpu-baseline off --parity off --share8 off --live2fps off --more-configs off"
for cfg in "256 4" "512 4" "1024 4" "256 8" "512 8" "256 4" "512 4"; do set -- $cfg
  LCC_ATTN_FUSED_BLOCKS=$1 LCC_ATTN_FUSED_WAVES=$2 python bench.py --streams-per-gpu 8 --steps 2 --warmup 1 --no-prefetch $QUIET 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict(fused_blocks=$1, fused_waves=$2, tokens_per_s=d['value'], decode_step_us=r['decode_step']['avg_step_us'], us_per_layer=r['decode_step']['us_per_layer'])))" >> $O/fused_attn_blocks_8streams_ab.jsonl
done
cat $O/fused_attn_blocks_8streams_ab.jsonl
