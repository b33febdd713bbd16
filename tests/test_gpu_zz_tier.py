"""-m gpu, collected LAST (tests/conftest.py sorts it behind everything): a green GPU tier must mean that every parity test RAN.
VERDICT r3 weak #4: the time-budget guards of conftest.py turn a test into a skip instead of running the tier into the driver's
20-minute kill; this test turns any such skip back into a failure (LCC_ALLOW_BUDGET_SKIPS=1: a deliberately shortened local run)."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_zz_no_parity_test_was_skipped_by_the_time_budget():
    from tests import conftest as C
    if os.environ.get("LCC_ALLOW_BUDGET_SKIPS") == "1":
        return
    assert not C.BUDGET_SKIPPED, (f"{len(C.BUDGET_SKIPPED)} GPU test(s) were skipped by the tier's time budget and did NOT run: "
                                  f"{C.BUDGET_SKIPPED[:8]} -- the tier is not green (slower box? raise the budget or move a live-oracle "
                                  f"test to a committed fixture)")


def test_zz_every_skip_of_the_gpu_tier_is_on_the_allow_list():
    """the tier reported "1 skipped" and no log named the test.  Every skipped GPU test is recorded with its reason
    (conftest.pytest_runtest_logreport; `-rs` in pytest.ini prints them too); anything that is not a known hardware-capacity / opt-in skip
    FAILS the tier, so green means: every comparison ran, or the log says exactly which one did not and why."""
    import torch
    from tests import conftest as C
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    unknown = [(n, r) for (n, r) in C.GPU_SKIPPED if not any(a in r for a in C.ALLOWED_SKIP_REASONS)]
    if os.environ.get("LCC_ALLOW_BUDGET_SKIPS") == "1":
        unknown = [(n, r) for (n, r) in unknown if "time budget" not in r and "time limit" not in r]
    print(f"GPU tier skips: {C.GPU_SKIPPED}")
    assert not unknown, f"{len(unknown)} GPU test(s) skipped for a reason that is not on the allow-list: {unknown[:8]}"
