import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multiprocess: launches child processes (torch.distributed.run, train_net.py, bench.py); collected last")


# Collection order under ``-x``: the operator parity tests - one kernel against the oracle, deterministic -
# run FIRST, then the trained-regime file, then everything else, the end-to-end file after that, and every test that launches
# child processes LAST.  The least deterministic tests must never stand between ``-x`` and an operator test.
_FILE_RANK = {"test_gpu_parity.py": 0, "test_gpu_trained.py": 1, "test_gpu_e2e.py": 3}


def pytest_collection_modifyitems(config, items):
    def key(item):
        multi = item.get_closest_marker("multiprocess") is not None
        return (1 if multi else 0, _FILE_RANK.get(os.path.basename(str(item.fspath)), 2))
    items.sort(key=key)                 # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return load
