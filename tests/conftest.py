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


class _FullSizeOracle:
    """Whole full-size proofs from the oracle, computed ONCE per session and shared by the tests that compare against them (each costs
    ~35 s of four host threads: the GPU suite has a time limit).  case(name) -> dict(vals, bl, seed, nb, P, fp, opr, ocm)."""
    SPECS = {"cfg2": (25000, 32, 4, (32, 7), 250004, b"\x51" * 32), "cfg4": (55000, 32, 4, (32, 7), 550004, b"\x52" * 32)}

    def __init__(self):
        self._cache = {}

    @staticmethod
    def inputs(name):
        import numpy as np
        d, nb, P, fp, rs, seed = _FullSizeOracle.SPECS[name]
        rng = np.random.default_rng(rs)
        mx = np.float32(((1 << (nb - 1)) - 1) / float(1 << fp[1]))      # get_clip_bounds(nb) (conversion32.rs:56-60), drawn from the half-open interval
        vals = np.clip(rng.uniform(-mx, mx, size=d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
        bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
        return dict(vals=vals, bl=bl, seed=seed, nb=nb, P=P, fp=fp, d=d)

    def case(self, name):
        if name not in self._cache:
            import orc
            os.environ.setdefault("OMP_NUM_THREADS", "4")
            c = self.inputs(name)
            rc, opr, ocm = orc.create_rangeproof(c["vals"], c["bl"], c["nb"], c["P"], c["fp"][0], c["fp"][1], seed=c["seed"])
            assert rc == 0
            c["opr"], c["ocm"] = opr, ocm
            self._cache[name] = c
        return self._cache[name]


@pytest.fixture(scope="session")
def full_oracle():
    return _FullSizeOracle()
