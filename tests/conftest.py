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


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gold():
    return np.load(os.path.join(GOLDEN, "small.npz"))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def case_input(orc, spec):
    if spec[0] == "rnd":
        return orc.rnd(spec[1], spec[2], spec[3])
    if spec[0] == "grad":
        return orc.grad(spec[1], spec[2])
    if spec[0] == "imgl":
        return orc.imgl(spec[1], spec[2], spec[3], spec[4])
    raise ValueError(spec)


def case_palette(orc, spec):
    if spec[0] == "U":
        return orc.generate_uniform_palette(spec[1])
    if spec[0] == "palr":
        return orc.palr(spec[1], spec[2] if len(spec) > 2 else 7)
    if spec[0] == "list":
        return [tuple(c) for c in spec[1]]
    raise ValueError(spec)
