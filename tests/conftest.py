import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def space_scene(built):
    from safemotionsrisk_amd.scene import build_scene, SPACE_ENV_CONFIG
    return build_scene(SPACE_ENV_CONFIG, auto_reset=False)


@pytest.fixture(scope="session")
def ball_scene(built):
    from safemotionsrisk_amd.scene import build_scene, BALL_ENV_CONFIG
    return build_scene(BALL_ENV_CONFIG, auto_reset=False)


@pytest.fixture(scope="session")
def bm_scene(built):
    """the space scene on robot_ball_machine.urdf (what the shipped backup_networks/space/params.json asks for)"""
    from safemotionsrisk_amd.scene import build_scene, SPACE_ENV_CONFIG
    return build_scene(dict(SPACE_ENV_CONFIG, ball_machine_mode=True), auto_reset=False)
