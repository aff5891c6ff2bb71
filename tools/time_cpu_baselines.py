#!/usr/bin/env python3
"""CPU baselines of the single-env loop, timed in the build container (the only place the reference exists).

The same loop is timed three ways on simple_spread (3 agents): reset() + 25 x step(uniform random int
actions), 200 episodes after 10 warm-up episodes, time.perf_counter:
  reference   hepengli/multiagent-particle-envs imported by path (MultiAgentEnv.step, environment.py:83-115)
  oracle      oracle/mpe_oracle.py, one env, one process
  oracle x N  one env per process, N = os.cpu_count() processes (subprocess workers of bench.py)

    python3 tools/time_cpu_baselines.py > profiles/r01_cpu_baselines_container.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

EPISODE, EPISODES, WARM = 25, 200, 10


def time_reference():
    import gen_golden as gg
    ref = gg.load_reference()
    sc, world, env = gg.make_env(ref, "simple_spread")
    env.seed(0)
    rng = np.random.RandomState(0)

    def episodes(n):
        for _ in range(n):
            env.reset()
            for t in range(EPISODE):
                env.step([int(a) for a in rng.randint(0, 5, 3)])
    episodes(WARM)
    t0 = time.perf_counter()
    episodes(EPISODES)
    return time.perf_counter() - t0


def time_oracle():
    import mpe_oracle as orc
    spec = orc.spec_spread()
    rng = np.random.RandomState(0)

    def episodes(n):
        for _ in range(n):
            pos, vel, _ = orc.reset(spec, rng, 1)
            for t in range(EPISODE):
                pos, vel, *_ = orc.env_step(spec, pos, vel, rng.randint(0, 5, (1, 3)))
    episodes(WARM)
    t0 = time.perf_counter()
    episodes(EPISODES)
    return time.perf_counter() - t0


def main():
    import bench
    out = {"loop": "reset + %d x step, %d episodes after %d warm-up, simple_spread 3 agents, uniform int actions" % (EPISODE, EPISODES, WARM),
           "host": "build container, %d CPUs" % os.cpu_count()}
    for name, fn in (("reference", time_reference), ("oracle", time_oracle)):
        dt = fn()
        out[name] = {"env_steps_per_s": EPISODES * EPISODE / dt, "agent_steps_per_s": 3 * EPISODES * EPISODE / dt,
                     "us_per_step": dt / (EPISODES * EPISODE) * 1e6, "cores": 1}
    out["oracle_n_process"] = bench.cpu_baseline_processes(os.cpu_count(), 6.0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
