import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The product picks the scan kernel by the table (BatchedPSRS._streams_apply: a table with a hot state takes the window kernel).  The
    # small tables of this suite nearly all have one, so the suite FORCES the row-packed kernel wherever it applies -- it is the headline
    # kernel -- and runs the window kernel through the variant matrix (test_gpu_edges.py) and the tests that switch to it; the automatic
    # choice has its own test (test_gpu_round3.py).
    os.environ.setdefault("OFFSIM_SCAN_ROWS", "1")
    # Likewise the shape of its launch: the product spreads a launch of few rollouts over the CUs (one to three chain wavefronts per
    # workgroup instead of four, offsim_eval_mc_streams), which is what every small test would get; the suite keeps the headline's shape
    # (four chain wavefronts + four helpers per workgroup) and runs the others through the variant matrix and the "auto" member.
    os.environ.setdefault("OFFSIM_ROWS_WAVES", "4")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
