import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# what the product library refuses BY DESIGN (include/f110.h: F110_ERR_STATE "... experimental build only"): the
# layouts, step forms and switches that were measured and not adopted.  Only these turn into a skip.
LAB_ONLY = (r"step_groups > 2 is available in the experimental build only",
            r"f110_exp_set\([a-z_0-9]+\) is available in the experimental build only")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """A test that asks the product library (the default) for one of the LAB_ONLY features is skipped, not failed: the same
    test runs for real under F110_LIB_VARIANT=experimental (tests/test_gpu_round3.py::test_gpu_suite_on_the_experimental_build,
    which asserts how many ran).  Any OTHER "experimental build only" refusal — a product code path that wrongly answers
    with it — stays a failure."""
    import re
    outcome = yield
    if outcome.excinfo is not None:
        from f1tenth_gym_amd._ffi import ExperimentalOnly
        ex = outcome.excinfo[1]
        if isinstance(ex, ExperimentalOnly) and any(re.search(p, str(ex)) for p in LAB_ONLY):
            outcome.force_exception(pytest.skip.Exception("experimental build only: %s" % ex, _use_item_location=True))
