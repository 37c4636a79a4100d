import os
import sys

import pytest

# the suites drive the library's development overrides (SPH_CELL_ORDER, SPH_TILE_SKIP, ... for A/B comparisons of layouts that must give the
# same bits); the library honours them only with SPH_DEV=1 (include/sph_mi355x.h: sph_overrides)
os.environ.setdefault("SPH_DEV", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kats.json")) as f:
        return json.load(f)
