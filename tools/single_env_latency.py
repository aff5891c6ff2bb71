#!/usr/bin/env python3
"""Latency of the single-env drop-in: microseconds per MultiAgentEnv.step through make_env(), beside the numpy
oracle's single-env step on the same host (the reference itself does not travel to the GPU box; in the build
container it measures 1.35x the oracle's time, profiles/r01_cpu_baselines_container.json).

    python3 tools/single_env_latency.py [--out profiles/r02_single_env_latency.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def time_env(name, benchmark, episodes, steps=25):
    import numpy as np
    from make_env import make_env
    env = make_env(name, benchmark=benchmark)
    env.seed(0)
    rng = np.random.RandomState(0)

    def action():
        out = []
        for sp in env.action_space:
            out.append(tuple(int(rng.randint(0, s.n)) for s in sp.spaces) if hasattr(sp, "spaces") else int(rng.randint(0, sp.n)))
        return out
    for _ in range(3):
        env.reset()
        for _ in range(steps):
            env.step(action())
    acts = [[action() for _ in range(steps)] for _ in range(episodes)]
    t_reset = t_step = 0.0
    for ep in range(episodes):
        t0 = time.perf_counter()
        env.reset()
        t1 = time.perf_counter()
        for a in acts[ep]:
            env.step(a)
        t2 = time.perf_counter()
        t_reset += t1 - t0
        t_step += t2 - t1
    return {"us_per_step": 1e6 * t_step / (episodes * steps), "us_per_reset": 1e6 * t_reset / episodes, "fused": env._scenario is not None}


def time_oracle(episodes, steps=25):
    import numpy as np
    import mpe_oracle as orc
    spec = orc.spec_spread()
    rng = np.random.RandomState(0)
    pos, vel, _ = orc.reset(spec, rng, 1)
    t0 = time.perf_counter()
    for _ in range(episodes):
        pos, vel, _ = orc.reset(spec, rng, 1)
        for _ in range(steps):
            pos, vel, *_ = orc.env_step(spec, pos, vel, rng.randint(0, 5, (1, 3)))
    return 1e6 * (time.perf_counter() - t0) / (episodes * steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--episodes", type=int, default=100)
    args = ap.parse_args()
    res = {"unit": "microseconds per MultiAgentEnv.step (one env, host objects in and out)", "episodes": args.episodes, "scenarios": {}}
    for name in ("simple_spread", "simple_tag", "simple_world_comm", "simple_reference", "collector"):
        for bench in (False, True):
            try:
                res["scenarios"]["%s%s" % (name, "+benchmark_data" if bench else "")] = time_env(name, bench, args.episodes)
            except AttributeError:      # the scenario defines no benchmark_data
                pass
    res["oracle_simple_spread_us_per_step"] = time_oracle(args.episodes)
    res["reference_simple_spread_us_per_step_estimate"] = 1.35 * res["oracle_simple_spread_us_per_step"]
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(os.path.join(ROOT, args.out), "w"), indent=1)


if __name__ == "__main__":
    main()
