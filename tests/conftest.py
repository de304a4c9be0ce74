import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def golden_inputs():
    """The shared uint8 input batch of tests/golden (regenerated from NumPy seeds, hash-checked)."""
    import hashlib
    import json
    import numpy as np
    from oracle import scoring as osc
    faces = osc.synthetic_faces(2, 2, seed=1234).reshape(4, 112, 112, 3)
    noise = np.random.default_rng(99).integers(0, 256, size=(4, 112, 112, 3), dtype=np.uint8)
    u8 = np.concatenate([faces, noise], 0)
    meta = json.load(open(os.path.join(GOLDEN, "inputs.json")))
    assert hashlib.sha256(u8.tobytes()).hexdigest() == meta["sha256"], "golden input batch drifted"
    return u8


@pytest.fixture(scope="session")
def golden_x(golden_inputs):
    import torch
    from oracle import scoring as osc
    return torch.stack([osc.to_model_input(im) for im in golden_inputs])
