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
  rollout  one mpe_episode launch per 25-step episode: on-device reset, the reset state's observation and
           25 steps (default: the headline number)
  step     one mpe_step launch per step, issued eagerly from Python (host-bound)
  graph    one mpe_step launch per step, the 25 launches of an episode replayed as one hipGraph
           (what an RL loop with a captured policy between steps would issue)

--steps K / --warmup W: the timed region always holds whole 25-step episodes and at least
TARGET_STEPS env steps, so that it is tens of milliseconds of GPU work and not one launch between two
host synchronisations: the K-step block is repeated R times inside ONE timed region (timing_plan
below); the line reports "steps" = R * K, the env steps that were actually timed, "steps_requested" = K
and "repeats" = R.  Warm-up is W steps rounded up to whole episodes (at least two).

--gpus N without WORLD_SIZE in the environment starts the N rank processes itself (fresh Python
processes, before anything touches the GPU); under torch.distributed.run it joins the ranks it is given.

Rank 0 prints ONE JSON line.  Besides the driver's keys it carries
  roofline      HBM roofline of the dominant kernel; achieved = algorithmic bytes per launch /
                average launch duration from HIP event pairs inside the timed region, each
                bracketing 4 consecutive launches (every 4th group by default, --event-every);
                traffic / valu = the rocprofv3 counter figures stored for this very build (null otherwise)
  cpu_baseline  the single-env numpy oracle (oracle/mpe_oracle.py) timed on this host, 1 core; its
                "n_process" entry is the same loop in --cpu-procs processes, one env each
  step_mode     the same workload issued as one mpe_step launch per step (hipGraph), for comparison
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))

EPISODE = 25
EVENT_GROUP = 4               # launches bracketed by one HIP event pair in rollout mode
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
TARGET_STEPS = 25000          # env steps a timed region holds at least (about 60 ms of the headline launch)
STEP_MODE_STEPS = 1000        # env steps of the mpe_step / hipGraph comparison run
STEP_MODE_WARMUP = 100


def timing_plan(steps, warmup, episode=EPISODE, target=TARGET_STEPS):
    """How a request for `steps` timed and `warmup` untimed env steps is run.

    The block of `steps` env steps is repeated `repeats` times in one timed region.  While the
    request is shorter than `target`, repeats is the smallest count that reaches `target` AND makes
    the region a whole number of episodes (every launch is then the full reset + `episode` steps
    launch); a request of `target` steps or more runs once, its last launch a partial episode when
    `steps` is not a multiple of `episode`.  Warm-up is rounded up to whole episodes, two at least.
    """
    if steps < 1:
        raise ValueError("--steps must be >= 1")
    if warmup < 0:
        raise ValueError("--warmup must be >= 0")
    if steps >= target:
        repeats = 1
    else:
        unit = episode // math.gcd(steps, episode)      # repeats must be a multiple of this
        repeats = -(-target // steps)
        repeats = -(-repeats // unit) * unit
    timed = repeats * steps
    return {"steps_requested": steps, "repeats": repeats, "timed_steps": timed,
            "full_launches": timed // episode, "tail_steps": timed % episode,
            "warmup_requested": warmup, "warmup_steps": max(2, -(-warmup // episode)) * episode}


def source_hash():
    """sha256 over the kernel sources and the C header: identifies the build a stored counter figure belongs to."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multiagent-particle-envs_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    for fn in files + [os.path.join(ROOT, "include", "mpe_hip.h")]:
        h.update(os.path.basename(fn).encode() + b"\0")
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def stored_traffic(key):
    """HBM bytes per launch from the rocprofv3 counter passes (tools/collect_profiles.py), or None when
    the stored figure was not measured on this build of the kernels."""
    try:
        entry = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
    except (OSError, KeyError, ValueError):
        return None
    if entry.get("source_hash") != source_hash():
        return None
    return entry.get("hbm_bytes_per_launch")

def stored_valu(key):
    """VALU figures of the dominant kernel from the rocprofv3 counter pass (profiles/*valu*.json, written by
    tools/collect_profiles.py), or None when none was measured on this build of the kernels."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*valu*.json")), reverse=True):
        try:
            entry = json.load(open(fn)).get(key)
        except (OSError, ValueError, AttributeError):
            continue
        if isinstance(entry, dict) and entry.get("source_hash") == source_hash() and "valu_insts_per_wave" in entry:
            return {"insts_per_wave": entry["valu_insts_per_wave"],
                    "issue_fraction_of_wave_lifetime": entry["valu_issue_fraction_of_wave_lifetime"],
                    "cycles_per_inst_while_issuing": entry["cycles_per_valu_inst_while_issuing"],
                    "source": "profiles/" + os.path.basename(fn)}
    return None


# Algorithmic bytes per env-step, float32 (s = 4), int32 actions, 1-byte done.
#   step launch:    reads  agent pos+vel 4A*s, landmark pos 2L*s, actions 4*n_act
#                   writes agent pos+vel 4A*s, obs obs_total*s, rewards A*s, done 1
#   rollout launch: state is read once and written once per launch of EPISODE steps, so per
#                   env-step only actions are read and obs, rewards, done written.
#   episode launch: as rollout, but the initial state is drawn on the device, not read, and the
#                   observation of the reset state (what env.reset() returns) is written once per episode.
#   consensus keeps ten past states per agent as state (hist_rows = 240 values per env): a launch that starts from
#   a given state reads them all and every single step writes the newest slot (4A values); an episode launch reads
#   none and keeps only the last slots, which are not counted.
def algorithmic_bytes(A, L, n_act, obs_total, s, steps_per_launch, reads_state=True, reset_obs=False, hist_rows=0):
    state_rd = (4 * A * s + 2 * L * s + hist_rows * s) if reads_state else 0
    state_wr = 4 * A * s + (4 * A * s if (hist_rows and reads_state) else 0)
    per_step = 4 * n_act + obs_total * s + A * s + 1
    return per_step + (state_rd + state_wr + (obs_total * s if reset_obs else 0)) / float(steps_per_launch)


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


def parse_args(argv=None):
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
                    help="rehearsal only: every rank uses GPU 0 and the ranks talk over gloo (a one-GPU box cannot host the real N > 1 run)")
    ap.add_argument("--dry", action="store_true",
                    help="rehearsal of the multi-process path without a GPU: ranks rendezvous (use --backend gloo), shard the env "
                         "index, all-reduce synthetic statistics and rank 0 prints the plan; nothing is simulated or timed")
    ap.add_argument("--episodes-per-launch", type=int, default=1,
                    help="rollout mode: consecutive episodes (each with its own on-device reset) per mpe_episodes launch")
    ap.add_argument("--target-steps", type=int, default=TARGET_STEPS,
                    help="the timed region holds at least this many env steps (the --steps block is repeated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=min(16, os.cpu_count() or 1),
                    help="processes of the N-process CPU figure (0 = skip); a one-GPU box's CPU share is 16")
    ap.add_argument("--no-step-mode", action="store_true")
    ap.add_argument("--event-every", type=int, default=4,
                    help="time every k-th group of 4 episode launches with one HIP event pair (a pair costs ~7 us on the "
                         "stream, so pairs around every launch would throttle submission and inflate the average)")
    return ap.parse_args(argv)


def spawn_ranks(n, argv):
    """Starts the n rank processes of a --gpus n run and waits for them; returns the exit code.

    Called before torch is imported: this process never touches the GPU and replaces nothing, it
    only starts fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what
    torch.distributed.run would set) and lets them inherit stdout, on which rank 0 prints the line.
    """
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    code = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and code == 0:
                code = rc
                for q in pending:           # a rank failed: the others would wait in a barrier for ever
                    q.terminate()
        time.sleep(0.05)
    return code


def dry_run(args, world, rank):
    """The multi-process path without a GPU: rendezvous, shards, the statistics all-reduce, one line."""
    import torch
    import torch.distributed as dist
    from mpe_hip.dist import EpisodeStats, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.backend == "nccl" else args.backend, rank=rank, world_size=world)
    first_env, last_env = shard_range(world * args.envs, rank, world)
    stats = EpisodeStats("cpu")
    # rewards derived from the global env index so that the global sum is known: -sum(range(world * envs))
    stats.add_episode(-torch.arange(first_env, last_env, dtype=torch.float64).view(1, -1, 1).expand(1, -1, args.agents))
    total = stats.reduce()
    shards = [None] * world
    if world > 1:
        dist.all_gather_object(shards, (first_env, last_env))
        dist.barrier()
        dist.destroy_process_group()
    else:
        shards = [(first_env, last_env)]
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "plan": timing_plan(args.steps, args.warmup, target=args.target_steps),
                          "shards": shards, "stats": total}))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(float(sys.argv[2]))
    args = parse_args()
    plan = timing_plan(args.steps, args.warmup, target=args.target_steps)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist

    dev_index = 0 if args.share_device else local_rank
    if args.share_device and args.backend == "nccl":
        args.backend = "gloo"       # RCCL refuses two ranks on one device
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
    outs = {}                                   # rollout output buffers by step count (full and partial launches)
    last = {"out": None}
    pairs = []                                  # (start, end, launches or steps) events around dominant-kernel launches
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
        done_steps, launches = 0, 0
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
                # one event pair brackets EVENT_GROUP consecutive full launches (an event pair costs
                # about 7 us on the stream, a tenth of one launch), every event_every-th group
                grp = launches // EVENT_GROUP
                rec = record and n == K * EPISODE and (grp % args.event_every == 0)
                if rec and launches % EVENT_GROUP == 0:
                    open_pair = torch.cuda.Event(enable_timing=True)
                    open_pair.record()
                elif not rec:
                    open_pair = None
                # mpe_episode: on-device reset of every env + n steps in one launch
                # (with the reset state's observation: everything "obs = env.reset(); 25 x env.step(a)" produces)
                outs[n] = env.rollout(actions[:n], outs.get(n), reset_seed=1234 + first_env, episodes=k_ep, obs0=True)
                last["out"] = outs[n]
                if rec and launches % EVENT_GROUP == EVENT_GROUP - 1 and open_pair is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    pairs.append((open_pair, e1, EVENT_GROUP))
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
                    rec = record and (launches % 8 == 0)
                    if rec:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    env.step(actions[t])
                    if rec:
                        e1.record()
                        pairs.append((e0, e1, 1))
                    launches += 1
            done_steps += n
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
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, launches

    def event_average():
        """Mean duration in ms of what one recorded unit covers (a launch, or a step of a graph replay)."""
        units = sum(n for _, _, n in pairs)
        if units == 0:
            return None
        ms = sum(a.elapsed_time(b) for a, b, _ in pairs) / units
        return ms if ms > 0.0 else None

    run(plan["warmup_steps"], args.mode, False)
    elapsed, launches = timed(plan["timed_steps"], args.mode, True)
    timed_steps = plan["timed_steps"]
    # average duration of one dominant-kernel launch and the env steps it covers (a graph replay
    # holds EPISODE single-step launches, so its time is divided by the steps it ran)
    avg_steps = float(K * EPISODE) if args.mode == "rollout" else 1.0
    kern_ms, kern_src = event_average(), "hip_events"
    if kern_ms is None:         # no complete event group (a request of a few launches only): the wall clock
        kern_ms, kern_src = elapsed * 1e3 / max(1.0, timed_steps / avg_steps), "wall_clock"
    s = 4 if args.dtype == "f32" else 8
    lo = env.layout
    # an episode's state write-back is once per launch; every episode of a launch is counted as if it wrote it
    bytes_env_step = algorithmic_bytes(A, env.n_landmarks, n_act, lo.obs_total, s, EPISODE if args.mode == "rollout" else 1,
                                       reads_state=args.mode != "rollout", reset_obs=args.mode == "rollout", hist_rows=lo.n_hist_rows)
    bytes_launch = bytes_env_step * B * avg_steps
    achieved = bytes_launch / (kern_ms * 1e-3) / 1e9

    step_mode = None
    if args.mode == "rollout" and not args.no_step_mode:
        pairs_saved = list(pairs)
        pairs.clear()
        run(STEP_MODE_WARMUP, "graph", False)
        dt2, _ = timed(STEP_MODE_STEPS, "graph", True)
        k2 = event_average() or dt2 * 1e3 / STEP_MODE_STEPS
        b2 = algorithmic_bytes(A, env.n_landmarks, n_act, lo.obs_total, s, 1, hist_rows=lo.n_hist_rows) * B
        step_mode = {"value": world * B * A * STEP_MODE_STEPS / dt2, "unit": "agent-steps/s", "steps": STEP_MODE_STEPS,
                     "ms_per_step": dt2 * 1e3 / STEP_MODE_STEPS,
                     "launch": "mpe_step, 1 step/launch, %d launches per hipGraph replay" % EPISODE, "launch_ms_hip_events": k2,
                     "roofline_frac": b2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_env_step": b2 / B}
        pairs[:] = pairs_saved

    # optional statistics all-reduce (RCCL), outside every timed region: return of agent 0 over the last episode
    stats = EpisodeStats(dev)
    stats.add_episode(last["out"]["rew"][-EPISODE:] if last["out"] is not None else env.rew)
    tot = stats.reduce()
    mean_rew0 = tot["return_agent0"] / max(1.0, tot["episodes"])

    if rank == 0:
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()      # a one-GPU figure: rank 0 at N = 1 only
        if cpu is not None and args.cpu_procs > 1:
            cpu["n_process"] = cpu_baseline_processes(args.cpu_procs)
        value = world * B * A * timed_steps / elapsed
        line = {
            "metric": "agent-steps/sec (whole node) batched simple_spread N=3" if args.scenario == "simple_spread" and A == 3
                      else "agent-steps/sec (whole node) batched %s" % args.scenario,
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": timed_steps, "warmup": plan["warmup_steps"],
            "steps_requested": args.steps, "warmup_requested": args.warmup, "repeats": plan["repeats"],
            "elapsed_s": elapsed, "ms_per_step": elapsed * 1e3 / timed_steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s A=%d L=%d, %d envs/GPU, %d-step episodes, uniform random int actions, on-device reset each episode"
                                   % (args.scenario, A, env.n_landmarks, B, EPISODE),
                       "envs_per_gpu": B, "global_envs": world * B, "episode_steps": EPISODE,
                       "launch": {"rollout": ("mpe_episode (on-device reset, its observation, %d steps) per launch" % EPISODE) if K == 1 else
                                             ("mpe_episodes: %d x (on-device reset, its observation, %d steps) per launch" % (K, EPISODE)), "step": "mpe_step, 1 step/launch, eager",
                                  "graph": "mpe_step, 1 step/launch, %d launches per hipGraph replay" % EPISODE}[args.mode],
                       "timed_region": "the %d-step request repeated %d times between one pair of barrier + synchronize" % (args.steps, plan["repeats"]),
                       "parallelism": "env-index shards, no data-path collective"},
            "env_steps_per_s": value / A,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # counters are not collected in a timing run: the figure is the stored rocprofv3 measurement
                         # of this very build of the kernels, null otherwise
                         "traffic": stored_traffic("%s_%s_B%d_%s" % (args.scenario, args.dtype, B, args.mode)),
                         # VALU use of the same kernel, from the same kind of stored counter pass (null otherwise)
                         "valu": stored_valu("%s_%s_B%d_%s" % (args.scenario, args.dtype, B, args.mode)),
                         "kernel": "%s<%s, %s, %s>" % ("consensus_kernel" if args.scenario == "consensus" else ("wave_kernel" if env.env_major else "small_kernel"),
                                                       "float" if s == 4 else "double", args.scenario, "rollout" if args.mode == "rollout" else "step"),
                         "bytes_per_env_step": bytes_env_step, "bytes_per_launch": bytes_launch,
                         "launch_ms": kern_ms, "launch_ms_source": kern_src, "event_pairs": len(pairs), "launches": launches},
            "cpu_baseline": cpu,
            "step_mode": step_mode,
            "mean_episode_return_agent0": mean_rew0,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
