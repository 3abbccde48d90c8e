import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(GOLDEN / name, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_mesh(g):
    return {k[5:]: v for k, v in g.items() if k.startswith("mesh_")}


def rel_l2(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    den = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0)


def mask_corners(a):
    """Interior + edge ghosts (the four corner ghosts are never written by BCs)."""
    a = np.array(a, dtype=float, copy=True)
    a[0, 0] = a[0, -1] = a[-1, 0] = a[-1, -1] = 0.0
    return a


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
