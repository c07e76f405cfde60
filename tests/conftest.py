import gzip
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_cache = {}


def load_golden(name):
    if name not in _cache:
        with gzip.open(os.path.join(GOLDEN, name + ".json.gz")) as f:
            _cache[name] = json.load(f)
    return _cache[name]


@pytest.fixture(scope="session")
def golden():
    return load_golden
