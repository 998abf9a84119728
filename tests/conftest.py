import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-detector_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU restatement (test infrastructure).  Builds oracle/liboracle_psroialign.so on demand."""
    from oracle import lighthead_oracle as O
    O.build_c_oracle()
    return O


@pytest.fixture(scope='session')
def lh_weights():
    from xdet import weights as W
    return W.make_lighthead_weights(1234)
