"""Shared pytest plumbing: marker registration, repo-root imports, fixture loading."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """A loaded .npz fixture: `.meta` is the JSON header, item access returns arrays."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.meta = json.loads(str(self._z["meta"]))

    def __getitem__(self, key):
        return self._z[key]

    def __contains__(self, key):
        return key in self._z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def table_callable(arr):
    """Density profile callable x in [0,1) -> float backed by a stored per-site table."""
    L = len(arr)
    return lambda x: float(arr[int(np.clip(np.round(x * L), 0, L - 1))])
