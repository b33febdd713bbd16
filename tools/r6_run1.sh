set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
export LCC_PARITY_OUT=$GRAFT_REPO_ROOT/gpurun_out/r6a
( time python -m pytest tests/test_gpu_server.py tests/test_gpu_torch_ops.py "tests/test_gpu_ops.py::test_unmodified_hf_model_on_the_gpu_with_every_plugin_matches_hf_cpu" tests/test_gpu_ops.py -k "server or torch_ops or plugin or sampler or eos or threshold" -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r6a/tests.log 2>&1
( time python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "eos_stops or stepped_threshold or teacher or forced" 2>&1 | tail -8 ) >> gpurun_out/r6a/tests.log 2>&1
timeout 150 python tools/r6_rccl_same_device_probe.py > gpurun_out/r6a/rccl_probe.json 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6a/pmc -o g -- python $GRAFT_REPO_ROOT/tools/r6_pmc_llm_gemms.py $GRAFT_REPO_ROOT/gpurun_out/r6a/pmc/manifest.json > $GRAFT_REPO_ROOT/gpurun_out/r6a/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/r6_summarize_llm_gemm_pmc.py gpurun_out/r6a/pmc > gpurun_out/r6a/llm_gemm_mfma_util.json 2>gpurun_out/r6a/summ.err
cp profiles/roofline_traffic.json gpurun_out/r6a/roofline_traffic.json
find gpurun_out/r6a/pmc -name "*.csv" -size +2000k -delete
tail -30 gpurun_out/r6a/tests.log; cat gpurun_out/r6a/rccl_probe.json; tail -5 gpurun_out/r6a/pmc.log; head -50 gpurun_out/r6a/llm_gemm_mfma_util.json; cat gpurun_out/r6a/summ.err | tail -5
