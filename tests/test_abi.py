"""The C-ABI library loads without a GPU and exports every symbol include/livecc_amd.h declares."""
import ctypes
import os
import subprocess

import pytest

from livecc_amd import _lib


@pytest.fixture(scope="module")
def lib(built_lib):
    return built_lib


def test_every_declared_symbol_is_exported(lib):
    names = _lib.declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_every_declared_symbol_has_a_ctypes_signature(lib):
    sig = lib._lcc_signatures
    assert sorted(sig) == _lib.declared_symbols()


def test_no_torch_or_cxx_types_in_the_header():
    import re
    src = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)   # code only, comments stripped
    for bad in ("torch", "at::", "std::", "template", "class "):
        assert bad not in src, bad
    # compiles as plain C
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", _lib.HEADER_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_version_and_error_paths_without_gpu(lib):
    assert b"gfx950" in lib.lcc_version()
    # argument validation happens before any device work
    assert lib.lcc_gemm_bf16(None, 0, None, 0, 0, None, None, 0, None, 0, 1, 16, 8, 0, None, 0, None) != 0
    assert b"null" in lib.lcc_last_error()
    assert lib.lcc_engine_create(None, None) is None
    cfg = _lib.ModelConfig(2048, 256, 512, 2, 2, 1, 64, 1e-6, 16, 24, 24, 2, 160, 2, 640, 1176, 2)   # head_dim 64: rejected
    lim = _lib.EngineLimits(1, 1024, 256, 1024, 16)
    assert lib.lcc_engine_create(ctypes.byref(cfg), ctypes.byref(lim)) is None
    assert b"head_dim" in lib.lcc_last_error()


def test_engine_sizing_queries_work_without_gpu(lib):
    from livecc_amd.config import livecc_7b
    c = livecc_7b()
    cfg = _lib.ModelConfig(c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                           c.num_key_value_heads, 128, 1e-6, 16, 24, 24, c.vit_depth, c.vit_embed_dim, c.vit_num_heads,
                           c.vit_mlp_dim, 1176, 2)
    lim = _lib.EngineLimits(8, 32768, 4096, 16384, 32)
    e = lib.lcc_engine_create(ctypes.byref(cfg), ctypes.byref(lim))
    assert e
    # 57,344 B per token x 32768 tokens
    assert lib.lcc_engine_kv_bytes_per_slot(e) == 57344 * 32768 == c.kv_bytes_per_token * 32768
    assert lib.lcc_engine_workspace_bytes(e) > 0 and lib.lcc_engine_meta_bytes(e) > 0
    lib.lcc_engine_destroy(e)


def test_product_has_no_cpu_fallback():
    """ops/engine must fail loudly without a GPU instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from livecc_amd import ops
    from livecc_amd.config import tiny
    from livecc_amd.engine import Engine
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(_lib.LccError):
        ops.linear(x, x)
    with pytest.raises((_lib.LccError, RuntimeError, AssertionError)):
        Engine(tiny(), None)


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "livecc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_no_kernel_spills_to_scratch():
    """Every gfx950 kernel of the library must fit its register budget: a scratch spill is a silent 3-5x slowdown
    (tools/check_resources.py compiles with -Rpass-analysis=kernel-resource-usage; no GPU needed)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "check_resources.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-500:]
    assert p.stdout.count("vgpr=") > 50


def test_production_attention_kernels_keep_their_accumulators_in_vgprs():
    """hipcc selects the AGPR form of the MFMAs whenever a kernel may run one wave per SIMD, and then copies the attention accumulators to the
    vector registers and back around every softmax (round 6: 88-144 v_accvgpr_read / write per key tile).  The kernels on the default path
    are built with a minimum of two waves per SIMD; tools/audit_agpr_copies.py reads the gfx950 assembly (no GPU needed)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_agpr_copies.py"), "attention.hip", "attn32.hip"], capture_output=True, text=True)
    assert p.returncode == 0 and "AGPR copies inside an MFMA block" in p.stdout, p.stdout[-1500:] + p.stderr[-500:]
    flagged = [line for line in p.stdout.splitlines() if line.startswith(("attention.hip", "attn32.hip"))]
    for name in ("attn_decode_kernel<", "attn_decode_fused_kernel<", "attn_gqa32_kernel<", "attn_vit32_kernel<8>"):
        assert not [line for line in flagged if name in line], (name, flagged)
