# round 6, GPU call 8: key tiles per split of the decode attention WITH the split cap raised (the round-5 A/B of LCC_ATTN_TPS ran under the cap of 64
# splits: at 6.5k keys = 204 tiles every setting below 4 tiles per split was clamped back to 4)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6h; mkdir -p $O
for cfg in "4 64" "2 128" "3 128" "4 64" "2 128" "3 128"; do
  set -- $cfg
  LCC_ATTN_TPS=$1 LCC_ATTN_MAXSPLIT=$2 python bench.py --steps 3 --warmup 1 --no-prefetch --cpu-baseline off --parity off --share8 off --live2fps off --more-configs off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict(tps=$1, maxsplit=$2, tokens_per_s=d['value'], decode_step_us=r['decode_step']['avg_step_us'], us_per_layer=r['decode_step']['us_per_layer'])))" >> $O/attn_tps_maxsplit_ab.jsonl
done
cat $O/attn_tps_maxsplit_ab.jsonl
