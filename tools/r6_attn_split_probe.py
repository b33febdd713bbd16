import json, os, sys
sys.path.insert(0, os.getcwd())
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_attn", "tools/bench_attn.py")
src = open("tools/bench_attn.py").read()
# reuse run()/tables() of the tool without its sweep: cut the file at the case list
head = src[:src.index("cases = [")]
ns = {"__file__": os.path.abspath("tools/bench_attn.py"), "__name__": "bench_attn_head"}
exec(compile(head, "bench_attn_head", "exec"), ns)
ops, torch = ns["ops"], ns["torch"]
kv = ops.KvArena(8, 1, ns["Hkv"], 24576 + 8192, ns["dev"])
kv.buf.copy_((torch.randn(kv.buf.shape, device=ns["dev"]) * 0.7).to(torch.bfloat16))
for name, segs in (("chunk_8streams", [(s, 386, 6200) for s in range(8)]), ("chunk_4streams", [(s, 386, 6200) for s in range(4)]),
                   ("chunk_8streams_12k", [(s, 386, 12000) for s in range(8)]), ("chunk_2streams", [(s, 386, 6200) for s in range(2)])):
    for rep in range(2):
        for tr, nsplit in ((32, 1), (36, 1), (36, 2), (32, 2), (36, 3), (36, 4)):
            us, pf, out = ns["run"](kv, segs, 3, tr, nsplit)
            print(json.dumps(dict(case=name, tile_rows=tr, nsplit=nsplit, us=round(us, 1), pflops=round(pf, 3))), flush=True)
