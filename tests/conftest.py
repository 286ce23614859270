import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One committed fixture: ``meta`` (dict) + named arrays, with torch views on demand."""

    def __init__(self, name):
        with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
            self.arrays = {k: f[k] for k in f.files if k != "meta"}
            self.meta = json.loads(str(f["meta"]))

    def t(self, key, device=None):
        x = torch.from_numpy(self.arrays[key])
        return x if device is None else x.to(device)

    def has(self, key):
        return key in self.arrays

    def state(self, prefix, device=None, strip=True):
        """state-dict entries stored as ``sd.<prefix><key>``."""
        full = "sd." + prefix
        return {(k[len(full):] if strip else k[3:]): self.t(k, device)
                for k in self.arrays if k.startswith(full)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
