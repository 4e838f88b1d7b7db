import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Achieved parity errors of this run -> gpurun_out/parity_errors_<pid>.json (tests/tolerance.py)."""
    import tolerance
    tolerance.dump()


class Golden(dict):
    def t(self, key, device=None):
        v = self.get(key)
        if v is None:
            return None
        out = torch.from_numpy(np.asarray(v))
        return out if device is None else out.to(device)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return Golden({k: z[k] for k in z.files})


@pytest.fixture
def golden():
    return load_golden


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))
