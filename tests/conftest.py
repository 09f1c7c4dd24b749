import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def pkg(sub=None):
    """The product package; its directory name has a hyphen, so import it by string."""
    name = "sahs-deformable-nerf_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def weights_mod():
    return pkg("weights")


@pytest.fixture(scope="session")
def flat_weights(weights_mod):
    cache = {}

    def get(seed=0, density_bias=0.0, density_gain=1.0):
        key = (int(seed), float(density_bias), float(density_gain))
        if key not in cache:
            cache[key] = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(*key))
        return cache[key]

    return get


def golden_rand(d):
    """Random tensors captured from the reference, in draw order."""
    return [(k.split("_")[-1], d[k]) for k in sorted(k for k in d if k.startswith("rand_"))]
