# round 6, GPU call 2: attention work-mapping A/B (XCD-chunked block order, (row, head) pair packing) + the tests they touch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b; mkdir -p $O; export LCC_PARITY_OUT=$GRAFT_REPO_ROOT/$O
( python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -5 ) > $O/tests.log 2>&1
( python -m pytest tests/test_gpu_server.py tests/test_gpu_torch_ops.py -m gpu -x -q 2>&1 | tail -8 ) >> $O/tests.log 2>&1
( python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "plugin" 2>&1 | tail -5 ) >> $O/tests.log 2>&1
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do
  set -- $cfg
  LCC_ATTN32_XCD=$1 LCC_ATTN32_PACK=$2 python tools/bench_attn.py --only32 2>/dev/null | sed "s/^{/{\"xcd\": $1, \"pack\": $2, /" >> $O/attn_map_ab.jsonl
  LCC_ATTN32_XCD=$1 LCC_ATTN32_PACK=$2 python - >> $O/attn_map_ab.jsonl 2>/dev/null <<PY
import json, sys, os
sys.path.insert(0, os.getcwd()); sys.argv = ["x", "--quick"]
import torch
from livecc_amd import _lib, ops
import importlib.util
spec = importlib.util.spec_from_file_location("ba", "tools/bench_attn.py")
# chunk_1stream with the engine's split (4) only
src = open("tools/bench_attn.py").read().split("cases = [")[0]
exec(src)
kv = ops.KvArena(8, 1, Hkv, 8192, dev)
kv.buf.copy_((torch.randn(kv.buf.shape, device=dev) * 0.7).to(torch.bfloat16))
for name, segs, ns in (("chunk_1stream", [(0, 386, 6200)], 4), ("chunk_1stream", [(0, 386, 6200)], 3), ("chunk_2streams", [(0, 386, 6200), (1, 386, 6200)], 2)):
    us, pf, out = run(kv, segs, 3, 32, ns, iters=40)
    print(json.dumps(dict(xcd=$1, pack=$2, case=name, nsplit=ns, us=round(us, 1), pflops=round(pf, 3))), flush=True)
PY
  LCC_ATTN32_XCD=$1 python tools/r5_tower.py "xcd$1" >> $O/tower_xcd_ab.jsonl 2>/dev/null
done
cat $O/tests.log; cat $O/attn_map_ab.jsonl; cat $O/tower_xcd_ab.jsonl
