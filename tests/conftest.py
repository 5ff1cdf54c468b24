"""Shared fixtures.  `-m "not gpu"` runs here (no GPU): oracle vs golden vectors,
host logic, ABI/export checks.  `-m gpu` runs on a real MI355X through the C ABI."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))   # tests/fp8_ref.py

import __graft_entry__ as graft  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle("vit_b_16")


@pytest.fixture(scope="session")
def weights(oracle):
    """Synthetic ViT-B/16 weights, seed_base 0 (the set the golden vectors were made with)."""
    return oracle.synth_weights(0)


@pytest.fixture(scope="session")
def golden_full():
    return np.load(GOLDEN / "b16_full.npz")


@pytest.fixture(scope="session")
def golden_stages():
    return np.load(GOLDEN / "b16_stages.npz")


@pytest.fixture(scope="session")
def device(pkg):
    """Initialise HIP device 0 or fail loudly (gpu-marked tests only)."""
    L = pkg.lib()
    rc = L.vh_init(0)
    assert rc == 0, f"vh_init(0) failed: {L.vh_last_error().decode()}"
    return 0
