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


def decided_with_margin(size, pos, margin, ordered=()):
    """Envs [B] (bool) whose integer outcomes cannot depend on a position error below `margin`.

    Every contact, forest, pick-up and hand-over test of the scenarios is ||p_a - p_b|| < size_a + size_b
    for some pair of entities: an env passes when every pair's distance is further than `margin` from that
    sum.  `ordered` lists (entity, candidates) groups whose distances the scenario sorts (collector's
    nearest-first lists): there the distances from the entity to two candidates must differ by more than
    `margin`, unless the two candidates sit on the very same point (an exact tie in any precision, which
    the scenario breaks by index)."""
    pos = np.asarray(pos, np.float64)
    size = np.asarray(size, np.float64)
    d = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(axis=3))            # [B, E, E]
    gap = np.abs(d - (size[:, None] + size[None, :]))
    E = pos.shape[1]
    ok = np.all(gap[:, ~np.eye(E, dtype=bool)] > margin, axis=1)
    for e, cand in ordered:
        cand = list(cand)
        for x in range(len(cand)):
            for y in range(x + 1, len(cand)):
                same_point = np.all(pos[:, cand[x]] == pos[:, cand[y]], axis=1)
                ok &= (np.abs(d[:, e, cand[x]] - d[:, e, cand[y]]) > margin) | same_point
    return ok


# ---- the float32 bar ----------------------------------------------------------------------------
# Trajectories of the float32 kernels must stay within 1e-4 of the reference's float64 ones after a 25-step
# episode: |got - ref| <= 1e-4 * max(1, |ref|), element by element.  Every comparison made through within_fp32_bar
# also records the largest |got - ref| / max(1, |ref|) it saw under its label; at the end of a GPU session the
# table goes to gpurun_out/fp32_margins.json (how far inside the bar each case really is).
FP32_BAR = 1e-4
FP32_MARGINS = {}


def within_fp32_bar(got, ref, label, factor=1.0):
    """True when every element of `got` is within factor * 1e-4 * max(1, |ref|) of `ref`; records the worst ratio.
    `factor` > 1 is for derived quantities whose error is a known multiple of the state's (stated where it is used)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if got.size == 0:
        return True
    ratio = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    worst = float(ratio.max()) if not np.isnan(ratio).any() else float("nan")
    rec = FP32_MARGINS.setdefault(label, {"worst_ratio": 0.0, "bar": FP32_BAR * factor, "elements": 0})
    rec["worst_ratio"] = worst if np.isnan(worst) else max(rec["worst_ratio"], worst)
    rec["elements"] += int(got.size)
    return bool(worst <= FP32_BAR * factor)


def pytest_sessionfinish(session, exitstatus):
    if not FP32_MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fp32_margins.json"), "w") as f:
            json.dump({"bar": "|got - ref| <= factor * 1e-4 * max(1, |ref|); worst_ratio = max |got - ref| / max(1, |ref|)",
                       "cases": dict(sorted(FP32_MARGINS.items()))}, f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden():
    return {k: load_golden(k) for k in GOLDEN_CASES}
