"""CPU: documentation drift guards and the GPU tier's time-budget logic (tests/conftest.py)."""
import glob
import os
import re
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def test_every_environment_switch_is_documented_and_every_documented_switch_exists():
    """INTEGRATION.md section 4c lists the LCC_* environment switches; the sources are the truth."""
    used = set()
    for f in glob.glob(os.path.join(ROOT, "livecc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "livecc_amd", "csrc", "*.h")):
        used |= set(re.findall(r'getenv\("(LCC_[A-Z0-9_]+)"\)', open(f).read()))
    for f in (glob.glob(os.path.join(ROOT, "livecc_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "*.py")) +
              [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]):
        if os.path.basename(f) == os.path.basename(__file__):
            continue
        used |= set(re.findall(r'environ(?:\.get)?[\[\(]"(LCC_[A-Z0-9_]+)"', open(f).read()))
    used.discard("LCC_BENCH_SELF_LAUNCHED")          # bench.py's own re-launch marker, not a user switch
    doc = _read("INTEGRATION.md")
    sec = doc[doc.index("## 4c."):doc.index("## 5. Build")]
    documented = set(re.findall(r"`(LCC_[A-Z0-9_]+)`", sec))
    assert used - documented == set(), f"switches read by the code but missing from INTEGRATION.md 4c: {sorted(used - documented)}"
    assert documented - used == set(), f"switches documented in INTEGRATION.md 4c that no source reads: {sorted(documented - used)}"


def test_header_symbol_count_matches_what_build_reports():
    """include/livecc_amd.h is the single declaration of the C-ABI: every `int lcc_*(` / `const char* lcc_*(` prototype is exported and
    bound (tests/test_abi.py checks the library); here: the README / INTEGRATION text does not name a symbol the header lacks."""
    hdr = _read("include", "livecc_amd.h")
    declared = set(re.findall(r"\b(lcc_[a-z0-9_]+)\s*\(", hdr))
    types = set(re.findall(r"\b(lcc_[a-z0-9_]+)\b", hdr)) - declared      # struct / enum / typedef names
    for doc in ("INTEGRATION.md", "README.md"):
        named = set(re.findall(r"`(lcc_[a-z0-9_]+)`", _read(doc)))
        missing = {n for n in named if n not in declared and n not in types}
        assert not missing, f"{doc} names C-ABI symbols the header does not declare: {sorted(missing)}"


def test_slow_tests_run_first_budget_skips_are_recorded_and_the_sentinel_fails_on_them(monkeypatch):
    from tests import conftest as C
    from tests import test_gpu_zz_tier as Z

    class Item:
        def __init__(self, nodeid):
            self.nodeid = nodeid
            self.keywords = {}
    items = [Item("tests/test_gpu_ops.py::test_a"), Item("tests/test_gpu_zz_tier.py::" + C.SENTINEL),
             Item("tests/test_gpu_e2e.py::test_baseline_config0_qwen2vl_2b_8frame_clip_vs_cpu_reference"),
             Item("tests/test_gpu_baseline_configs.py::test_greedy_tokens_are_exact_on_decisive_weights[livecc-7b]"), Item("tests/test_abi.py::test_b"),
             Item("tests/test_gpu_layer_parity.py::test_every_layer_at_livecc_7b_shapes_matches_hf_on_the_oracles_input")]
    order = [it.nodeid.split("::")[1][:20] for it in sorted(items, key=C._slow_rank)]
    assert order == ["test_every_layer_at_", "test_greedy_tokens_a", "test_baseline_config", "test_a", "test_b", C.SENTINEL[:20]]
    # the live twins of the committed fixtures are opt-in; each names a fixture test that exists
    src = _read("tests", "test_gpu_golden.py")
    for live, twin in C.LIVE_TWINS.items():
        assert C._is_live_twin("x::" + live) and f"def {twin}(" in src, (live, twin)
    assert not C._is_live_twin("tests/test_gpu_layer_parity.py::test_every_layer_at_livecc_7b_shapes_matches_hf_on_the_oracles_input")
    # budget: a test that needs 300 s does not start 600 s into a session whose slow block must end by 780 s -- and the skip is recorded
    class Req:
        node = Item("tests/x.py::test_needs_300")
    check = C.slow_budget.__wrapped__(Req()) if hasattr(C.slow_budget, "__wrapped__") else None
    if check is None:
        pytest.skip("fixture internals not reachable in this pytest version")
    monkeypatch.setattr(C, "_SESSION_T0", time.time() - 600.0)
    monkeypatch.setattr(C, "BUDGET_SKIPPED", [])
    monkeypatch.delenv("LCC_SLOW_DEADLINE_S", raising=False)
    monkeypatch.delenv("LCC_ALLOW_BUDGET_SKIPS", raising=False)
    Z.test_zz_no_parity_test_was_skipped_by_the_time_budget()           # nothing skipped yet: green
    with pytest.raises(pytest.skip.Exception):
        check(300)
    assert C.BUDGET_SKIPPED == ["tests/x.py::test_needs_300"]
    with pytest.raises(AssertionError):
        Z.test_zz_no_parity_test_was_skipped_by_the_time_budget()       # a budget skip fails the tier
    monkeypatch.setenv("LCC_ALLOW_BUDGET_SKIPS", "1")
    Z.test_zz_no_parity_test_was_skipped_by_the_time_budget()
    assert check(100) >= 600.0
    monkeypatch.setenv("LCC_SLOW_DEADLINE_S", "0")
    assert check(10_000) >= 600.0


@pytest.mark.parametrize("tag,start,end", [("r04", "`r04/` (round 4)", "`r03/` (round 3)"), ("r03", "`r03/` (round 3)", "`r02/` (round 2)")])
def test_profiles_readme_names_only_files_that_exist(tag, start, end):
    """profiles/README.md, sections of rounds 3 and 4: every file they name is committed under profiles/<round>/ (globs and {a,b} groups
    allowed; the final-tree set of the CURRENT round is written by the last GPU call and checked once it exists)."""
    doc = _read("profiles", "README.md")
    sec = doc[doc.index(start):doc.index(end)]
    have = set(os.listdir(os.path.join(ROOT, "profiles", tag)))
    if tag == "r04" and "bench_kernel_stats.csv" not in have:        # before the round's final validation call: skip that row
        sec = "\n".join(r for r in sec.splitlines() if "tools/validate_on_gpu.sh r04` |" not in r)
    missing = []
    for row in sec.splitlines():
        if not row.startswith("| `"):
            continue
        first = row.split("|")[1]
        for name in re.findall(r"`([A-Za-z0-9_*{},.\-]+\.(?:json|jsonl|csv|txt))`", first):
            pats = [name]
            m = re.search(r"\{([^}]*)\}", name)
            if m:
                pats = [name[:m.start()] + alt + name[m.end():] for alt in m.group(1).split(",")]
            for p in pats:
                rx = re.compile("^" + re.escape(p).replace(r"\*", ".*") + "$")
                if not any(rx.match(h) for h in have):
                    missing.append(p)
    assert not missing, f"profiles/README.md names files that are not under profiles/{tag}/: {missing}"


def test_roofline_traffic_is_reported_only_for_the_kernel_version_that_was_profiled(tmp_path):
    """bench.py's `roofline.traffic` comes from a committed PMC pass; the JSON records the sha256 of the kernel's source file and a
    mismatch nulls the field (the constant could silently go stale)."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    p = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    d = json.load(open(p))
    assert d.get("kernel_source") == "livecc_amd/csrc/decode_v2.hip" and len(d.get("kernel_source_sha16", "")) == 16
    stale = dict(d, kernel_source_sha16="0" * 16)
    q = tmp_path / "stale.json"
    q.write_text(json.dumps(stale))
    assert bench.pmc_traffic_if_current(str(q)) is None
    fresh = bench.pmc_traffic_if_current(p)
    assert fresh is None or abs(fresh / d["algorithmic_bytes_per_launch"] - 1.0) < 0.05      # None: kernel edited since the last PMC pass
