python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
LCC_PREFILL_QKV_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -2
B="python bench.py --cpu-baseline off --parity off --steps 3 --warmup 1"
for cfg in "LCC_PREFILL_QKV_SPLIT=0" "LCC_PREFILL_QKV_SPLIT=1"; do
  echo "== prefetch   $cfg $(env $cfg $B 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | tr '\n' ' ')"
  echo "== noprefetch $cfg $(env $cfg $B --no-prefetch 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | tr '\n' ' ')"
done
