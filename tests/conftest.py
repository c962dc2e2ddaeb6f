import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is many small torch ops: with torch's default of one thread per core it runs ~10 x SLOWER on the GPU box's
    # 128-core host than on 16 threads (scripts/probes/oracle_threads.py, profiles/r05_oracle_threads.txt: the 64-row code predictor
    # 37.6 s at 128 threads, 3.3 s at 16) -- most of the GPU suite's 16 minutes was that.  Results do not depend on the count.
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
