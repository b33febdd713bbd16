cd $GRAFT_REPO_ROOT
QUIET="--cpu-baseline off --parity off --share8 off --live2fps off --more-configs off"
python bench.py --steps 2 --warmup 1 $QUIET 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict(tokens_per_s=d['value'], live=dict(avg_launch_us=r['avg_launch_us'], frac=r['frac'], n=r['launches_timed']), without_tower=r.get('without_tower_overlap'), step=r['decode_step']['avg_step_us'])))"
python bench.py --steps 2 --warmup 1 --config qwen2vl-72b --weights fp8 $QUIET 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps(dict(tokens_per_s=d['value'], live=dict(avg_launch_us=r['avg_launch_us'], frac=r['frac']))))"
python -m pytest tests/test_gpu_decode_v2.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
