import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native libraries are built in-tree by __graft_entry__.build(); tests never fall back."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import maxwell_amd as mx
    c = mx.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle_pillbox12():
    from oracle import yee
    s = yee.pillbox_sim(12)
    return s, s.wave_ops()


@pytest.fixture(scope="session")
def oracle_pillbox16():
    from oracle import yee
    s = yee.pillbox_sim(16)
    return s, s.wave_ops()


@pytest.fixture(scope="session")
def oracle_periodic6():
    from oracle import yee
    s = yee.YeeSim(6, strict=False)
    return s, s.wave_ops()


@pytest.fixture
def rng():
    return np.random.default_rng(12345)
