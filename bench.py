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

Rank 0 prints ONE JSON line.  "steps" and "warmup" are the values asked for; "timed_steps", "warmup_steps" and
"repeats" say what was run.  Besides the driver's keys the line carries
  roofline      HBM roofline of the dominant kernel; achieved = algorithmic bytes per launch /
                average launch duration from HIP event pairs inside the timed region, each
                bracketing 4 consecutive launches (every 4th group by default, --event-every);
                traffic / valu = the rocprofv3 counter figures stored for this very build (null otherwise)
  cpu_baseline  the single-env numpy oracle (oracle/mpe_oracle.py) timed on this host, 1 core; its
                "n_process" entry is the same loop in --cpu-procs processes, one env each
  step_mode     the same workload issued as one mpe_step launch per step (hipGraph), for comparison
  step_auto     the same as mpe_step_auto launches (episode bookkeeping and time-limit reset on the device): the
                call a trainer's vectorised env makes per step (mpe_hip.train.BatchedVecEnv)
  sustained     (N = 1) the headline launch again after the card has been kept busy: >= 1 s of untimed launches, then
                >= 2 s timed -- what a training loop that never idles would see
  configs       (N = 1) one record per other configuration BASELINE.json names (simple_tag 131 072 envs,
                simple_world_comm 262 144, simple_spread 64/64 x 2 048, simple_spread 3/3 in float64) and the headline
                scenario at 1 048 576 envs, whose output no longer fits the 256 MiB Infinity Cache: launch_ms from HIP
                events over >= 200 ms of episode launches, algorithmic bytes, fraction of 8 TB/s, agent-steps/s
  ranks         (N > 1) what torch.distributed reports: backend, world size, and every rank's (env_base, seed, device)
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
RESET_SEED = 1234             # one seed for the whole job: rank r's envs draw as envs [r * B, (r + 1) * B) of the global batch
SUSTAINED_WARM_S = 1.0        # the sustained figure: this long untimed, then ...
SUSTAINED_TIMED_S = 2.0       # ... this long timed
CONFIG_TIMED_S = 0.25         # timed launches of every other BASELINE configuration
# (name, scenario, envs per GPU, dtype, scenario keywords): BASELINE.json's configs 3-5, config 2's float64 run, and
# the headline scenario at a batch whose output (4.4 GB) cannot sit in the 256 MiB Infinity Cache
OTHER_CONFIGS = [
    ("simple_tag B=131072 f32 (1M envs over 8 GPUs: the per-GPU shard)", "simple_tag", 131072, "f32", {}),
    ("simple_world_comm B=262144 f32", "simple_world_comm", 262144, "f32", {}),
    ("simple_spread 64/64 B=2048 f32 (16 384 envs over 8 GPUs: the per-GPU shard)", "simple_spread", 2048, "f32", dict(n_agents=64, n_landmarks=64)),
    ("simple_spread 3/3 B=65536 f64 (parity variant)", "simple_spread", 65536, "f64", {}),
    ("simple_spread 3/3 B=1048576 f32 (output beyond the Infinity Cache)", "simple_spread", 1048576, "f32", {}),
]


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
#   mpe_step_auto: a step launch that also reads and writes the env's step counter (4 + 4 B) and running return
#                   (2 * A * s), reads its episode number (4 B), and once per episode of `auto_period` steps writes the
#                   finished return (A * s) and the episode number (4 B).
def algorithmic_bytes(A, L, n_act, obs_total, s, steps_per_launch, reads_state=True, reset_obs=False, hist_rows=0, auto_period=0):
    state_rd = (4 * A * s + 2 * L * s + hist_rows * s) if reads_state else 0
    state_wr = 4 * A * s + (4 * A * s if (hist_rows and reads_state) else 0)
    per_step = 4 * n_act + obs_total * s + A * s + 1
    if auto_period:
        per_step += 12 + 2 * A * s + (A * s + 4) / float(auto_period)
    return per_step + (state_rd + state_wr + (obs_total * s if reset_obs else 0)) / float(steps_per_launch)


def cpu_baseline(seconds=10.0):
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
    ap.add_argument("--no-step-mode", action="store_true", help="skip the step_mode and step_auto records")
    ap.add_argument("--no-configs", action="store_true", help="skip the records of the other BASELINE configurations")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained figure")
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

    # The port is found by binding port 0 and handed to the ranks by number; SO_REUSEADDR on the probe lets rank 0 bind
    # the port again at once.  (Under torch.distributed.run, which is how the driver starts N > 1, the launcher owns
    # the rendezvous and this function is not used.)
    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
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
    mine = {"rank": rank, "env_base": first_env, "envs": last_env - first_env, "seed": RESET_SEED}
    ranks = [None] * world
    if world > 1:
        dist.all_gather_object(ranks, mine)
        info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks, info = [mine], {"backend": None, "world_size": 1}
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "plan": timing_plan(args.steps, args.warmup, target=args.target_steps),
                          "shards": [(r["env_base"], r["env_base"] + r["envs"]) for r in ranks], "stats": total,
                          "ranks": dict(info, ranks=ranks)}))


def random_actions(torch, env, steps, gen):
    """Uniform random int actions [steps, B, n_act]: moves 0..4, comm words uniform over dim_c."""
    a = torch.randint(0, 5, (steps, env.n_envs, env.n_act), generator=gen, device=env.device, dtype=torch.int32)
    for col, n in enumerate(env.action_dims):
        if n != 5:
            a[:, :, col] = torch.randint(0, n, (steps, env.n_envs), generator=gen, device=env.device, dtype=torch.int32)
    return a


def time_episode_launches(torch, env, actions, timed_s, warm_s=0.0, event_every=4):
    """Episode launches (on-device reset, its observation, EPISODE steps) of `env` back to back: about `warm_s` seconds
    untimed, then about `timed_s` seconds between two synchronisations, with one HIP event pair around every
    event_every-th group of EVENT_GROUP launches.  Returns launch_ms (HIP events), wall seconds, launches."""
    out = {"o": None}

    def launch():
        out["o"] = env.rollout(actions, out["o"], reset_seed=RESET_SEED, obs0=True)

    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        launch()
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 8, 1e-6)          # seconds per launch, for sizing the loops
    for _ in range(int(warm_s / est)):
        launch()
    n = max(4 * EVENT_GROUP, -(-int(timed_s / est) // EVENT_GROUP) * EVENT_GROUP)
    pairs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        grp, pos = divmod(k, EVENT_GROUP)
        rec = grp % event_every == 0
        if rec and pos == 0:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        launch()
        if rec and pos == EVENT_GROUP - 1:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            pairs.append((e0, e1))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sum(a.elapsed_time(b) for a, b in pairs) / (len(pairs) * EVENT_GROUP)
    return ms, wall, n


def config_record(torch, mpe_hip, dev, name, scenario, B, dtype, kw, timed_s=CONFIG_TIMED_S, warm_s=0.0):
    """One BASELINE configuration as episode launches on this GPU: launch_ms by HIP events, algorithmic bytes of the
    launch, fraction of the 8 TB/s roofline, agent-steps/s by the wall clock of the timed launches."""
    tdt = torch.float32 if dtype == "f32" else torch.float64
    env = mpe_hip.BatchedEnv(scenario, B, dtype=tdt, device=dev, **kw)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    actions = random_actions(torch, env, EPISODE, gen)
    ms, wall, n = time_episode_launches(torch, env, actions, timed_s, warm_s)
    lo, s = env.layout, (4 if dtype == "f32" else 8)
    per_step = algorithmic_bytes(env.n_agents, env.n_landmarks, env.n_act, lo.obs_total, s, EPISODE, reads_state=False, reset_obs=True,
                                 hist_rows=lo.n_hist_rows)
    bytes_launch = per_step * B * EPISODE
    return {"name": name, "scenario": scenario, "agents": env.n_agents, "landmarks": env.n_landmarks, "envs": B, "dtype": dtype,
            "launch": "mpe_episode (on-device reset, its observation, %d steps)" % EPISODE, "launch_ms": ms,
            "bytes_per_env_step": per_step, "bytes_per_launch": bytes_launch, "frac": bytes_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "agent_steps_per_s": B * env.n_agents * EPISODE * n / wall, "launches": n, "seconds": wall}


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
    from mpe_hip.dist import EpisodeStats, shard_range
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    kw = dict(n_agents=args.agents, n_landmarks=args.landmarks) if args.scenario == "simple_spread" else {}
    B = args.envs
    first_env, last_env = shard_range(world * B, rank, world)    # this rank's slice of the global env index
    assert last_env - first_env == B
    # env_base = the shard's first env: with ONE seed on every rank the job simulates exactly the worlds a single
    # batch of world * B envs would (the draws are keyed by the global env index)
    env = mpe_hip.BatchedEnv(args.scenario, B, dtype=tdt, device=dev, env_base=first_env, **kw)
    A, n_act = env.n_agents, env.n_act

    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    K = max(1, args.episodes_per_launch) if args.mode == "rollout" else 1
    actions = random_actions(torch, env, K * EPISODE, gen)
    outs = {}                                   # rollout output buffers by step count (full and partial launches)
    last = {"out": None}
    pairs = []                                  # (start, end, launches or steps) events around dominant-kernel launches
    graphs = {}
    auto_env = {}

    def stepper(which):
        """The env a step-per-launch mode steps: the main one, or (mpe_step_auto) one with the episode bookkeeping."""
        if which != "auto":
            return env
        if "e" not in auto_env:
            auto_env["e"] = mpe_hip.BatchedEnv(args.scenario, B, dtype=tdt, device=dev, env_base=first_env, max_episode_steps=EPISODE,
                                               seed=RESET_SEED, **kw)
            auto_env["e"].reset(seed=RESET_SEED, observe=False)
        return auto_env["e"]

    def episode_graph(which):
        """hipGraph of one episode's 25 mpe_step (or mpe_step_auto) launches (captured once)."""
        if which not in graphs:
            e = stepper(which)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e.step(actions[0])
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for t in range(EPISODE):
                    e.step(actions[t])
            graphs[which] = g
        return graphs[which]

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
            if mode in ("step", "graph"):
                env.reset(seed=RESET_SEED)          # (mpe_step_auto resets itself, inside the step that ends an episode)
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
                outs[n] = env.rollout(actions[:n], outs.get(n), reset_seed=RESET_SEED, episodes=k_ep, obs0=True)
                last["out"] = outs[n]
                if rec and launches % EVENT_GROUP == EVENT_GROUP - 1 and open_pair is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    pairs.append((open_pair, e1, EVENT_GROUP))
                    open_pair = None
                launches += 1
            elif mode in ("graph", "auto") and n == EPISODE:
                g = episode_graph(mode)
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                g.replay()
                if record:
                    e1.record()
                    pairs.append((e0, e1, n))
                launches += n
            else:
                e = stepper(mode)
                for t in range(n):
                    rec = record and (launches % 8 == 0)
                    if rec:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    e.step(actions[t])
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
    n_pairs = len(pairs)
    s = 4 if args.dtype == "f32" else 8
    lo = env.layout
    # an episode's state write-back is once per launch; every episode of a launch is counted as if it wrote it
    bytes_env_step = algorithmic_bytes(A, env.n_landmarks, n_act, lo.obs_total, s, EPISODE if args.mode == "rollout" else 1,
                                       reads_state=args.mode != "rollout", reset_obs=args.mode == "rollout", hist_rows=lo.n_hist_rows)
    bytes_launch = bytes_env_step * B * avg_steps
    achieved = bytes_launch / (kern_ms * 1e-3) / 1e9

    def per_step_record(mode, what, auto_period=0):
        """The workload as one launch per step, EPISODE launches per hipGraph replay."""
        pairs.clear()
        run(STEP_MODE_WARMUP, mode, False)
        dt2, _ = timed(STEP_MODE_STEPS, mode, True)
        k2 = event_average() or dt2 * 1e3 / STEP_MODE_STEPS
        b2 = algorithmic_bytes(A, env.n_landmarks, n_act, lo.obs_total, s, 1, hist_rows=lo.n_hist_rows, auto_period=auto_period) * B
        return {"value": world * B * A * STEP_MODE_STEPS / dt2, "unit": "agent-steps/s", "steps": STEP_MODE_STEPS,
                "ms_per_step": dt2 * 1e3 / STEP_MODE_STEPS, "launch": "%s, 1 step/launch, %d launches per hipGraph replay" % (what, EPISODE),
                "launch_ms_hip_events": k2, "roofline_frac": b2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_env_step": b2 / B}

    step_mode = step_auto = None
    if args.mode == "rollout" and not args.no_step_mode:
        step_mode = per_step_record("graph", "mpe_step")
        # what a trainer's vectorised env issues per step (mpe_hip.train.BatchedVecEnv): the step plus the episode
        # bookkeeping, and the on-device reset in the step that reaches the time limit
        step_auto = per_step_record("auto", "mpe_step_auto (time limit %d)" % EPISODE, auto_period=EPISODE)
        auto_env.clear()
        graphs.pop("auto", None)

    # optional statistics all-reduce (RCCL), outside every timed region: return of agent 0 over the last episode
    stats = EpisodeStats(dev)
    stats.add_episode(last["out"]["rew"][-EPISODE:] if last["out"] is not None else env.rew)
    tot = stats.reduce()
    mean_rew0 = tot["return_agent0"] / max(1.0, tot["episodes"])

    ranks = None
    if world > 1:
        mine = {"rank": rank, "env_base": first_env, "envs": B, "seed": RESET_SEED, "device": torch.cuda.get_device_name(dev)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": gathered}

    sustained = None
    if world == 1 and args.mode == "rollout" and K == 1 and not args.no_sustained:
        ms, wall, n = time_episode_launches(torch, env, actions, SUSTAINED_TIMED_S, SUSTAINED_WARM_S, args.event_every)
        sustained = {"launch_ms": ms, "frac": bytes_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "agent_steps_per_s": B * A * EPISODE * n / wall,
                     "launches": n, "seconds": wall, "warmup_seconds": SUSTAINED_WARM_S,
                     "what": "the headline launch after >= %.0f s of untimed launches, timed for >= %.0f s" % (SUSTAINED_WARM_S, SUSTAINED_TIMED_S)}

    configs = None
    if world == 1 and args.mode == "rollout" and not args.no_configs:
        del env, outs, last, actions
        graphs.clear()
        torch.cuda.empty_cache()
        configs = []
        for name, scn, b_c, dt_c, kw_c in OTHER_CONFIGS:
            configs.append(config_record(torch, mpe_hip, dev, name, scn, b_c, dt_c, kw_c))
            torch.cuda.empty_cache()

    if rank == 0:
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()      # a one-GPU figure: rank 0 at N = 1 only
        if cpu is not None and args.cpu_procs > 1:
            cpu["n_process"] = cpu_baseline_processes(args.cpu_procs)
        value = world * B * A * timed_steps / elapsed
        line = {
            "metric": "agent-steps/sec (whole node) batched simple_spread N=3" if args.scenario == "simple_spread" and A == 3
                      else "agent-steps/sec (whole node) batched %s" % args.scenario,
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_steps": timed_steps, "warmup_steps": plan["warmup_steps"], "repeats": plan["repeats"],
            "elapsed_s": elapsed, "ms_per_step": elapsed * 1e3 / timed_steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s A=%d L=%d, %d envs/GPU, %d-step episodes, uniform random int actions, on-device reset each episode"
                                   % (args.scenario, A, lo.n_landmarks, B, EPISODE),
                       "envs_per_gpu": B, "global_envs": world * B, "episode_steps": EPISODE, "reset_seed": RESET_SEED,
                       "launch": {"rollout": ("mpe_episode (on-device reset, its observation, %d steps) per launch" % EPISODE) if K == 1 else
                                             ("mpe_episodes: %d x (on-device reset, its observation, %d steps) per launch" % (K, EPISODE)), "step": "mpe_step, 1 step/launch, eager",
                                  "graph": "mpe_step, 1 step/launch, %d launches per hipGraph replay" % EPISODE}[args.mode],
                       "timed_region": "the %d-step request repeated %d times between one pair of barrier + synchronize" % (args.steps, plan["repeats"]),
                       "parallelism": "env-index shards (rank r = envs [r * B, (r + 1) * B) of the global batch, one seed), no data-path collective"},
            "env_steps_per_s": value / A,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # counters are not collected in a timing run: the figure is the stored rocprofv3 measurement
                         # of this very build of the kernels, null otherwise
                         "traffic": stored_traffic("%s_%s_B%d_%s" % (args.scenario, args.dtype, B, args.mode)),
                         # VALU use of the same kernel, from the same kind of stored counter pass (null otherwise)
                         "valu": stored_valu("%s_%s_B%d_%s" % (args.scenario, args.dtype, B, args.mode)),
                         "kernel": "%s<%s, %s, %s>" % ("consensus_kernel" if args.scenario == "consensus" else (("wave_stream_kernel" if s == 4 and args.mode == "rollout" and (lo.n_agents * (lo.n_agents + lo.n_landmarks + 1)) % 2 == 0
                                                         else "wave_kernel") if lo.state_env_major else "small_kernel"),
                                                       "float" if s == 4 else "double", args.scenario, "rollout" if args.mode == "rollout" else "step"),
                         "bytes_per_env_step": bytes_env_step, "bytes_per_launch": bytes_launch,
                         "launch_ms": kern_ms, "launch_ms_source": kern_src, "event_pairs": n_pairs, "launches": launches},
            "cpu_baseline": cpu,
            "step_mode": step_mode,
            "step_auto": step_auto,
            "sustained": sustained,
            "configs": configs,
            "ranks": ranks,
            "mean_episode_return_agent0": mean_rew0,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
