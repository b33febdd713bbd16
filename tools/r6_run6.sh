# round 6, GPU call 6: static wave priority A/B on the 32x32x16 attention kernels (LCC_ATTN32_PRIO = 0 / 1 / 2), A/B/A/B order
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6f; mkdir -p $O
for p in 0 1 2 0 1 2; do
  LCC_ATTN32_PRIO=$p python tools/bench_attn.py --only32 2>/dev/null | sed "s/^{/{\"prio\": $p, /" >> $O/attn_prio_ab.jsonl
  LCC_ATTN32_PRIO=$p python tools/r5_tower.py "prio$p" >> $O/tower_prio_ab.jsonl 2>/dev/null
done
grep -h "chunk_8streams\|first_turn\|4096" $O/attn_prio_ab.jsonl | grep '"nsplit": 1' | cut -c1-200; cat $O/tower_prio_ab.jsonl
