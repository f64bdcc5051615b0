import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


@pytest.fixture(scope="session")
def golden():
    return load_golden


def synth_state_dict(module, seed, prefix):
    """state dict of `module` (a cremage_amd module built on CPU) filled with the name-keyed synthetic weights"""
    from cremage_amd.synth import synth_fill_
    synth_fill_(module, seed, prefix=prefix)
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
