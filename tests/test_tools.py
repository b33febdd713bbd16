"""Host-side tools: the kernel-trace breakdown (tools/trace_breakdown.py) on a synthetic rocprofv3 CSV."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_breakdown_finds_decode_steps_and_gaps(tmp_path):
    spec = importlib.util.spec_from_file_location("trace_breakdown", os.path.join(ROOT, "tools", "trace_breakdown.py"))
    tb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tb)
    hdr = '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp"\n'
    t, lines = 1000, []

    def k(name, dur, gap=100, queue=1):
        nonlocal t
        lines.append(f'"KERNEL_DISPATCH","Agent 2",{queue},0,1,1,1,"{name}",1,{t},{t + dur}\n')
        t += dur + gap

    k("lcc::seen_set_kernel(unsigned int*, int)", 3000)                                        # a prefill call: marks ids, GEMMs, samples
    k("void lcc::gemm_big_kernel<128, 4, 1, false>(unsigned short const*, int)", 100000)
    k("lcc::sample_final_kernel(float const*, int)", 2000)
    for _ in range(2):
        k("lcc::decode_step_begin_kernel(int const*, int const*)", 3000)
        for _layer in range(2):
            k("void lcc::dgemv_kernel<1, 1, 3, 8, 4, 2>(lcc::DgArgs)", 10000)
            k("void lcc::dgemv_kernel<2, 1, 1, 4, 1, 2>(lcc::DgArgs)", 40000)
        k("lcc::sample_partial_kernel(unsigned short const*, int)", 8000)
        k("lcc::sample_final_kernel(float const*, int)", 2000, gap=5000)
    p = tmp_path / "t_kernel_trace.csv"
    p.write_text(hdr + "".join(lines))
    out = tb.breakdown(tb.load(str(p)), n_layers=2)
    assert out["decode_steps"] == 2
    assert out["avg_kernel_time_per_step_us"] == 3 + 2 * (10 + 40) + 8 + 2
    assert out["avg_gap_per_step_us"] == 0.6                       # 6 gaps of 100 ns inside a step
    assert out["kernels"]["dgemv_kernel<2, 1, 1, 4, 1, 2>"] == dict(calls_per_step=2.0, avg_us=40.0, us_per_step=80.0)
    assert "gemm_big_kernel<128, 4, 1, false>" not in out["kernels"]
