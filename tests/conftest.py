import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "multiagent-particle-envs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a fixture captured from the reference by tools/gen_golden.py."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


GOLDEN_CASES = {
    "spread3": ("simple_spread", dict(A=3, L=3)),
    "tag": ("simple_tag", {}),
    "world_comm": ("simple_world_comm", {}),
    "spread64": ("simple_spread", dict(A=64, L=64)),
}


# simple, simple_adversary, simple_push: fixtures also hold the goal landmark index per env
CLASSIC_CASES = ["simple", "simple_adversary", "simple_push"]

# scenarios with immovable and / or speaking agents; fixtures hold the drawn landmark indices (goal),
# the comm state, the agents' own rewards before sharing and the callbacks on the reset state
COMM_CASES = ["simple_speaker_listener", "simple_reference", "simple_crypto"]


@pytest.fixture(scope="session")
def golden():
    return {k: load_golden(k) for k in GOLDEN_CASES}
