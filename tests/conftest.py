import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture(autouse=True)
def _default_precision():
    """every test starts from the default operand formats (base bf16, inference stages per ops.POLICIES["mixed"]);
    tests that want one format everywhere call ops.set_compute_dtype themselves"""
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.reset_precision()
    yield
    ops.reset_precision()
