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
