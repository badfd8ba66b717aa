import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def prim():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "primitives.json")))


@pytest.fixture(scope="session")
def golden_proofs():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))


@pytest.fixture(scope="session")
def hiplib():
    """The product library; building it needs hipcc only (no GPU)."""
    from rofl_project_code_amd import build, api
    build.build()
    return api.lib()
