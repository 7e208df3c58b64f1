import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def tiny_bundle():
    from mars5_tts_amd import synth
    return synth.make_bundle("tiny", seed=0)


@pytest.fixture(scope="session")
def full_bundle():
    """Seeded random checkpoints of the real MARS5 geometry (takes ~20 s of host time: built once per session)."""
    from mars5_tts_amd import synth
    return synth.make_bundle("full", seed=0)
