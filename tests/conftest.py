import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def pkg():
    import denseASPP  # noqa: F401  (loads the hyphenated package directory)
    return sys.modules['semantic_segmentation_amd']


TINY_CFG = dict(bn_size=4, drop_rate=0, growth_rate=8, num_init_features=16, block_config=(2, 2, 2, 2),
                dropout0=0.1, dropout1=0.1, d_feature0=32, d_feature1=16)
