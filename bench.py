#!/usr/bin/env python3
"""Throughput of the batched particle-world step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N == 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one MultiAgentEnv.step of every env in the batch.  The workload is the one
BASELINE.json's metric is quoted on: simple_spread with 3 agents / 3 landmarks, 65 536 envs per
GPU, float32, 25-step episodes with pre-generated uniform random int actions (synthetic), every
env reset on device at each episode boundary.  Envs shard by index across GPUs with no data-path
collective (weak scaling: the per-GPU batch is fixed).

Launch modes (--mode):
  rollout  one mpe_episode launch per 25-step episode: on-device reset + 25 steps (default: the
           headline number)
  step     one mpe_step launch per step, issued eagerly from Python (host-bound)
  graph    one mpe_step launch per step, the 25 launches of an episode replayed as one hipGraph
           (what an RL loop with a captured policy between steps would issue)

Rank 0 prints ONE JSON line.  Besides the driver's keys it carries
  roofline      HBM roofline of the dominant kernel; achieved = algorithmic bytes per launch /
                average launch duration from HIP event pairs inside the timed region, each
                bracketing 4 consecutive launches (every 4th group by default, --event-every)
  cpu_baseline  the single-env numpy oracle (oracle/mpe_oracle.py) timed on this host, 1 core; its
                "n_process" entry is the same loop in --cpu-procs processes, one env each
  step_mode     the same workload issued as one mpe_step launch per step (hipGraph), for comparison
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))

EPISODE = 25
EVENT_GROUP = 4               # launches bracketed by one HIP event pair in rollout mode
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak

# Algorithmic bytes per env-step, float32 (s = 4), int32 actions, 1-byte done.
#   step launch:    reads  agent pos+vel 4A*s, landmark pos 2L*s, actions 4*n_act
#                   writes agent pos+vel 4A*s, obs obs_total*s, rewards A*s, done 1
#   rollout launch: state is read once and written once per launch of EPISODE steps, so per
#                   env-step only actions are read and obs, rewards, done written.
#   episode launch: as rollout, but the initial state is drawn on the device, not read.
def algorithmic_bytes(A, L, n_act, obs_total, s, steps_per_launch, reads_state=True):
    state_rd = (4 * A * s + 2 * L * s) if reads_state else 0
    state_wr = 4 * A * s
    per_step = 4 * n_act + obs_total * s + A * s + 1
    return per_step + (state_rd + state_wr) / float(steps_per_launch)


def cpu_baseline(seconds=12.0):
    """Single-env reset + 25 x step loop of the numpy oracle, one process, one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import mpe_oracle as orc
    spec = orc.spec_spread()
    rng = np.random.RandomState(0)
    episodes, steps = 0, 0
    for _ in range(3):   # warm-up episodes
        pos, vel, _ = orc.reset(spec, rng, 1)
        for t in range(EPISODE):
            pos, vel, *_ = orc.env_step(spec, pos, vel, rng.randint(0, 5, (1, 3)))
    t0 = time.perf_counter()
    while True:
        pos, vel, _ = orc.reset(spec, rng, 1)
        for t in range(EPISODE):
            pos, vel, comm, obs, rew, done = orc.env_step(spec, pos, vel, rng.randint(0, 5, (1, 3)))
        episodes += 1
        steps += EPISODE
        el = time.perf_counter() - t0
        if el >= seconds and episodes >= 20:
            break
    return {"value": steps * 3 / el, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": "%d episodes x %d steps of one simple_spread env (3 agents), numpy oracle, %.1f s" % (episodes, EPISODE, el)}


def _cpu_worker(seconds):
    """One process of the N-process CPU figure: prints "<env steps> <seconds>"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import mpe_oracle as orc
    spec = orc.spec_spread()
    rng = np.random.RandomState(os.getpid())
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        pos, vel, _ = orc.reset(spec, rng, 1)
        for t in range(EPISODE):
            pos, vel, *_ = orc.env_step(spec, pos, vel, rng.randint(0, 5, (1, 3)))
        steps += EPISODE
    print(steps, time.perf_counter() - t0)


def cpu_baseline_processes(n_procs, seconds=6.0):
    """The same loop as cpu_baseline, one env per process in n_procs fresh Python processes (no GPU use)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds)],
                              stdout=subprocess.PIPE, env=env) for _ in range(n_procs)]
    steps, longest = 0, 0.0
    for p in procs:
        o, _ = p.communicate(timeout=seconds * 10 + 120)
        if p.returncode != 0:
            return None
        n, el = o.split()
        steps, longest = steps + int(n), max(longest, float(el))
    return {"value": steps * 3 / longest, "unit": "agent-steps/s", "cores": n_procs,
            "sample": "%d processes x one simple_spread env, %.1f s each" % (n_procs, seconds)}


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(float(sys.argv[2]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2500)
    ap.add_argument("--warmup", type=int, default=250)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--scenario", default="simple_spread")
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--landmarks", type=int, default=3)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step", "graph"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a one-GPU box cannot host the real N > 1 run)")
    ap.add_argument("--episodes-per-launch", type=int, default=1,
                    help="rollout mode: consecutive episodes (each with its own on-device reset) per mpe_episodes launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=min(16, os.cpu_count() or 1),
                    help="processes of the N-process CPU figure (0 = skip); a one-GPU box's CPU share is 16")
    ap.add_argument("--no-step-mode", action="store_true")
    ap.add_argument("--event-every", type=int, default=4,
                    help="time every k-th group of 4 episode launches with one HIP event pair (a pair costs ~7 us on the "
                         "stream, so pairs around every launch would throttle submission and inflate the average)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import mpe_hip
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    kw = dict(n_agents=args.agents, n_landmarks=args.landmarks) if args.scenario == "simple_spread" else {}
    env = mpe_hip.BatchedEnv(args.scenario, args.envs, dtype=tdt, device=dev, **kw)
    from mpe_hip.dist import EpisodeStats, shard_range
    B, A, n_act = args.envs, env.n_agents, env.n_act
    first_env, last_env = shard_range(world * B, rank, world)    # this rank's slice of the global env index
    assert last_env - first_env == B

    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    K = max(1, args.episodes_per_launch) if args.mode == "rollout" else 1
    actions = torch.randint(0, 5, (K * EPISODE, B, n_act), generator=gen, device=dev, dtype=torch.int32)
    for col, n in enumerate(env.action_dims):        # comm words are uniform over dim_c
        if n != 5:
            actions[:, :, col] = torch.randint(0, n, (K * EPISODE, B), generator=gen, device=dev, dtype=torch.int32)
    out = None
    pairs = []                                  # (start, end, steps) events around dominant-kernel launches
    graphs = {}

    def episode_graph():
        """hipGraph of one episode's 25 mpe_step launches (captured once)."""
        if "g" not in graphs:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                env.step(actions[0])
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for t in range(EPISODE):
                    env.step(actions[t])
            graphs["g"] = g
        return graphs["g"]

    def run(n_steps, mode, record):
        """n_steps env steps in episodes of EPISODE; returns the number of dominant-kernel launches."""
        nonlocal out
        done_steps, launches, ep = 0, 0, 0
        open_pair = None
        while done_steps < n_steps:
            n = min(EPISODE, n_steps - done_steps)
            k_ep = 1
            if mode == "rollout" and K > 1 and n == EPISODE:
                k_ep = max(1, min(K, (n_steps - done_steps) // EPISODE))
                n = k_ep * EPISODE
            if mode != "rollout":
                env.reset(seed=1234 + first_env)
            if mode == "rollout":
                # one event pair brackets EVENT_GROUP consecutive launches (an event pair costs
                # about 7 us on the stream, a tenth of one launch), every event_every-th group
                grp = launches // EVENT_GROUP
                rec = record and n == K * EPISODE and (grp % args.event_every == 0)
                if rec and launches % EVENT_GROUP == 0:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    open_pair = e0
                # mpe_episode: on-device reset of every env + n steps in one launch
                out = env.rollout(actions[:n], out if (out is not None and out["obs"].shape[0] == n) else None,
                                  reset_seed=1234 + first_env, episodes=k_ep)
                if rec and launches % EVENT_GROUP == EVENT_GROUP - 1 and open_pair is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    pairs.append((open_pair, e1, n * EVENT_GROUP))
                    open_pair = None
                launches += 1
            elif mode == "graph" and n == EPISODE:
                g = episode_graph()
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                g.replay()
                if record:
                    e1.record()
                    pairs.append((e0, e1, n))
                launches += n
            else:
                for t in range(n):
                    if record:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    env.step(actions[t])
                    if record:
                        e1.record()
                        pairs.append((e0, e1, 1))
                    launches += 1
            done_steps += n
            ep += 1
        return launches

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, mode, record):
        barrier()
        t0 = time.perf_counter()
        launches = run(n_steps, mode, record)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, launches

    run(args.warmup, args.mode, False)
    elapsed, launches = timed(args.steps, args.mode, True)
    steps_per_launch = K * EPISODE if args.mode == "rollout" else 1
    # average duration of one dominant-kernel launch and the env steps it covers (a graph replay
    # holds EPISODE single-step launches, so its time is divided by the steps it ran)
    tot_ms = sum(a.elapsed_time(b) for a, b, _ in pairs)
    if args.mode == "rollout":
        kern_ms, avg_steps = tot_ms / max(1, len(pairs) * EVENT_GROUP), float(K * EPISODE)
    else:
        kern_ms, avg_steps = tot_ms / max(1, sum(n for _, _, n in pairs)), 1.0

    if not pairs or kern_ms <= 0.0:          # too few launches for an event group: fall back to the wall clock
        kern_ms = elapsed * 1e3 / max(1, launches)
    s = 4 if args.dtype == "f32" else 8
    lo = env.layout
    # an episode's state write-back is once per launch; every episode of a launch is counted as if it wrote it
    bytes_env_step = algorithmic_bytes(A, env.n_landmarks, n_act, lo.obs_total, s, EPISODE if args.mode == "rollout" else 1,
                                       reads_state=args.mode != "rollout")
    bytes_launch = bytes_env_step * B * avg_steps
    achieved = bytes_launch / (kern_ms * 1e-3) / 1e9

    step_mode = None
    if args.mode == "rollout" and not args.no_step_mode:
        pairs_saved = list(pairs)
        pairs.clear()
        n2 = min(args.steps, 1000)
        n2 -= n2 % EPISODE
        run(min(args.warmup, 100), "graph", False)
        dt2, _ = timed(n2, "graph", True)
        k2 = sum(a.elapsed_time(b) for a, b, _ in pairs) / max(1, sum(n for _, _, n in pairs))
        b2 = algorithmic_bytes(A, env.n_landmarks, n_act, lo.obs_total, s, 1) * B
        step_mode = {"value": world * B * A * n2 / dt2, "unit": "agent-steps/s", "steps": n2, "ms_per_step": dt2 * 1e3 / n2,
                     "launch": "mpe_step, 1 step/launch, %d launches per hipGraph replay" % EPISODE, "launch_ms_hip_events": k2,
                     "roofline_frac": b2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_env_step": b2 / B}
        pairs[:] = pairs_saved

    # optional statistics all-reduce (RCCL), outside every timed region: return of agent 0 over the last episode
    stats = EpisodeStats(dev)
    stats.add_episode(out["rew"][-EPISODE:] if out is not None else env.rew)
    tot = stats.reduce()
    mean_rew0 = tot["return_agent0"] / max(1.0, tot["episodes"])

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "%s_%s_B%d_%s" % (args.scenario, args.dtype, B, args.mode)
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    if rank == 0:
        cpu = None if args.no_cpu_baseline else cpu_baseline()
        if cpu is not None and args.cpu_procs > 1:
            cpu["n_process"] = cpu_baseline_processes(args.cpu_procs)
        value = world * B * A * args.steps / elapsed
        line = {
            "metric": "agent-steps/sec (whole node) batched simple_spread N=3" if args.scenario == "simple_spread" and A == 3
                      else "agent-steps/sec (whole node) batched %s" % args.scenario,
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s A=%d L=%d, %d envs/GPU, %d-step episodes, uniform random int actions, on-device reset each episode"
                                   % (args.scenario, A, env.n_landmarks, B, EPISODE),
                       "envs_per_gpu": B, "global_envs": world * B, "episode_steps": EPISODE,
                       "launch": {"rollout": ("mpe_episode (on-device reset + %d steps) per launch" % EPISODE) if K == 1 else
                                             ("mpe_episodes: %d x (on-device reset + %d steps) per launch" % (K, EPISODE)), "step": "mpe_step, 1 step/launch, eager",
                                  "graph": "mpe_step, 1 step/launch, %d launches per hipGraph replay" % EPISODE}[args.mode],
                       "parallelism": "env-index shards, no data-path collective"},
            "env_steps_per_s": value / A,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "%s<%s, %s, %s>" % ("consensus_kernel" if args.scenario == "consensus" else ("wave_kernel" if env.env_major else "small_kernel"),
                                                       "float" if s == 4 else "double", args.scenario, "rollout" if args.mode == "rollout" else "step"),
                         "bytes_per_env_step": bytes_env_step, "bytes_per_launch": bytes_launch,
                         "launch_ms_hip_events": kern_ms, "launches": launches},
            "cpu_baseline": cpu,
            "step_mode": step_mode,
            "mean_episode_return_agent0": mean_rew0,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
