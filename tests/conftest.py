import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def scene_path(name):
    return os.path.join(SCENES, name + ".dae")


@pytest.fixture(scope="session")
def device():
    """One GPU context for the whole GPU test session."""
    import pathtracer_amd as pt
    dev = pt.Device(0)
    yield dev
    dev.close()
