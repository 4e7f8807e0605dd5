import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def smpl_model():
    import hpe_amd

    return hpe_amd.synthetic.make_smpl_model()


@pytest.fixture(scope="session")
def oracle_smpl(smpl_model):
    from oracle import hmr_oracle as O

    return O.SMPL(smpl_model)
