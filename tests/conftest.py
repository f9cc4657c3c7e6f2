"""pytest configuration: marker registration + import paths.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol presence (no GPU).
``-m gpu``: parity tests proper; they call the HIP path through the C-ABI library.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "motion-policy-networks_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "geometry_golden.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def loss_golden():
    path = os.path.join(ROOT, "tests", "golden", "loss_golden.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def model_golden():
    """Vectors made by running the reference's own model.py (tests/golden/gen_model_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "model_golden.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def dense_golden():
    """The reference's model.py run on clouds built to OVERFLOW both ball queries (tests/golden/gen_dense_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "dense_golden.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def caller_golden():
    """Vectors made by running the reference's data_loader.py / run_inference.py (tests/golden/gen_caller_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "caller_golden.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def metrics_golden():
    """Vectors made by running the PyBullet-free methods of the reference's Evaluator (tests/golden/gen_metrics_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "metrics_golden.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc
