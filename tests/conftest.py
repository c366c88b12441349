import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # A/B builds (tools/ab_variants.py): DS2_TEST_LIB=path/to/libds2hip_<name>.so runs the suite against a variant library
    if os.environ.get("DS2_TEST_LIB"):
        from deepspeech.pytorch_amd import _lib
        assert os.path.exists(os.environ["DS2_TEST_LIB"]), os.environ["DS2_TEST_LIB"]
        _lib.LIB_PATH = os.environ["DS2_TEST_LIB"]
