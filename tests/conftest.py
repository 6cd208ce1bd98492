import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "aa-clip-iqm_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_tiny():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "tiny.npz"))


@pytest.fixture(scope="session")
def golden_full():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "full.npz"))


# Measured parity errors of the GPU tests (name -> numbers), written to gpurun_out/parity_errors.json at session end and
# committed per round under profiles/ (bench.py's `parity_vs_north_star` reads the committed copy).
PARITY_ERRORS = {}


@pytest.fixture(scope="session", autouse=True)
def _dump_parity_errors():
    yield
    if not PARITY_ERRORS:
        return
    import json
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_errors.json")
    merged = {}
    if os.environ.get("AACLIP_PARITY_MERGE") and os.path.exists(path):   # several pytest invocations, one record
        try:
            with open(path) as f:
                merged = json.load(f)
        except (OSError, ValueError):
            merged = {}
    merged.update(PARITY_ERRORS)
    with open(path, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)
