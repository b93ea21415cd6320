import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIG_DIR = os.path.join(ROOT, 'config_jsons')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')


def load_json(name):
    with open(os.path.join(CONFIG_DIR, name), 'rt') as f:
        return json.load(f)


@pytest.fixture(scope='session')
def student_cfg():
    return load_json('parallel_wavenet.json')


@pytest.fixture(scope='session')
def gauss_student_cfg():
    return load_json('parallel_wavenet_gauss.json')
