import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'yolov3v4-modelcompression-multidatasettraining-multibackbone_amd')
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg_root():
    return PKG


@pytest.fixture(scope='session')
def cfg_dir():
    return os.path.join(PKG, 'cfg')
