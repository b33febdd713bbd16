cd $GRAFT_REPO_ROOT; O=gpurun_out/r6d; mkdir -p $O; export LCC_PARITY_OUT=$GRAFT_REPO_ROOT/$O
( time python -m pytest tests/test_gpu_ops.py tests/test_gpu_resize.py tests/test_gpu_rccl.py tests/test_gpu_server.py tests/test_gpu_torch_ops.py tests/test_gpu_vit_fused.py tests/test_gpu_zz_tier.py -x -q -m gpu 2>&1 | tail -30 ) > $O/pytest_tail.txt 2>&1
tail -32 $O/pytest_tail.txt
