"""CPU tier: the TORCH_LIBRARY registration of the operator-level entry points (csrc/torch_ops.cpp -> _C/liblivecc_torch_ops.so) builds
against the installed PyTorch-ROCm, loads without a GPU, registers every op `livecc_amd.torch_ops.OP_NAMES` lists with a schema, and has
NO CPU kernels (a CPU tensor raises: the product has no CPU path).  The GPU parity of the ops is tests/test_gpu_torch_ops.py."""
import pytest
import torch


@pytest.fixture(scope="module")
def ns(built_lib):
    from livecc_amd import build, torch_ops
    if not torch_ops.available():
        build.build_torch_ops(verbose=False)
    return torch_ops.load()


def test_every_operator_is_registered_with_a_schema(ns):
    from livecc_amd import torch_ops
    for name in torch_ops.OP_NAMES:
        op = getattr(ns, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"livecc_amd::{name}("), schema
    assert "Tensor(a!) kv_buf" in str(ns.rope_kv_append.default._schema), "the in-place KV append is annotated as a mutation"
    assert "Tensor(a!) seen" not in str(ns.rmsnorm.default._schema)


def test_cpu_tensors_are_refused(ns):
    x, w = torch.zeros(2, 64, dtype=torch.bfloat16), torch.zeros(64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        ns.rmsnorm(x, w, 1e-6)
    with pytest.raises(NotImplementedError):
        ns.swiglu(x, x)
    with pytest.raises(RuntimeError):          # schema check: wrong argument type
        ns.rmsnorm(x, w, "eps")


def test_ops_module_routes_through_the_dispatcher_only_when_allowed(ns, monkeypatch):
    from livecc_amd import _lib, ops, torch_ops
    assert torch_ops.try_load() is ns
    with pytest.raises(_lib.LccError):          # the wrapper's own validation still speaks first: "no CPU path"
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.zeros(64, dtype=torch.bfloat16))
    monkeypatch.setattr(torch_ops, "_ns", None)
    monkeypatch.setattr(torch_ops, "_tried", False)
    monkeypatch.setenv("LCC_TORCH_OPS", "0")
    assert torch_ops.try_load() is None and torch_ops.op("rmsnorm") is None
