#!/usr/bin/env python3
"""Profile bench.py with rocprofv3 on the GPU box and summarise the result.

Four separate runs (counters are never combined with tracing, and FETCH_SIZE / WRITE_SIZE do not
fit one pass):
  1. rocprofv3 --kernel-trace --stats      -> per-kernel call counts and average durations
  2. rocprofv3 --pmc FETCH_SIZE            -> bytes read from beyond L2, per dispatch
  3. rocprofv3 --pmc WRITE_SIZE            -> bytes written beyond L2, per dispatch
  4. rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
                                           -> VALU instructions per wave and the share of a wave's lifetime spent issuing them

The dominant kernel is the one with the largest total duration in pass 1; the counter passes pick
its dispatches by that exact name.

Writes into <out>/ :  kernel_stats.csv (copy of pass 1's table), bench.json (the line bench.py
printed under pass 1), valu.json and traffic.json, keyed like bench.py's lookup
("<scenario>_<dtype>_B<envs>_<mode>") and stamped with bench.source_hash() of the kernel sources they
were measured on (bench.py reports a stored figure only for the same sources), holding HBM bytes per
launch of the dominant kernel:
    hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests
at 64 bytes, i.e. half the bytes of a coalesced streaming read, hence the factor 2 (checked here
against the known byte count of this kernel's reads, see "fetch_check" in the output).

Usage (on the GPU box, through gpurun):
    python3 tools/collect_profiles.py --out gpurun_out/prof_r01 [-- extra bench.py args]
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as bench_mod  # noqa: E402  (source_hash; importing it touches no GPU)


def run(cmd, log, limit):
    with open(log, "w") as f:
        try:
            return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=limit, cwd=ROOT).returncode
        except subprocess.TimeoutExpired:
            return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prof")
    ap.add_argument("--scenario", default="simple_spread")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--mode", default="rollout")
    ap.add_argument("--kernel", default=None, help="substring the dominant kernel's name must contain (default: any)")
    args, extra = ap.parse_known_args()
    out = os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    os.environ.update(env)
    bench = ["python3", "bench.py", "--no-cpu-baseline", "--no-configs", "--no-sustained", "--scenario", args.scenario, "--envs", str(args.envs),
             "--dtype", args.dtype, "--mode", args.mode] + [a for a in extra if a != "--"]
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(out, "stats"), "--"] + bench,
             os.path.join(out, "stats.log"), 400)
    print("stats pass rc", rc)
    if rc != 0:
        sys.exit(rc)
    for fn in glob.glob(os.path.join(out, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(fn, os.path.join(out, "kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))) if (args.kernel or "") in r["Name"]]
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    kern = top["Name"]
    print("dominant kernel:", kern, "calls", top["Calls"], "avg ns", top["AverageNs"])
    for line in open(os.path.join(out, "stats.log")):
        if line.startswith("{") and '"metric"' in line:
            open(os.path.join(out, "bench.json"), "w").write(line)
    short = bench + ["--no-step-mode", "--steps", "250", "--warmup", "25", "--target-steps", "250"]
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, counter.lower())
        rc = run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + short, os.path.join(out, counter.lower() + ".log"), 200)
        print(counter, "pass rc", rc)
        if rc != 0:
            sys.exit(rc)
        vals = []
        for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] == counter and r["Kernel_Name"] == kern:
                    vals.append(float(r["Counter_Value"]))
        res[counter] = (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
    fetch_kib, n_f = res["FETCH_SIZE"]
    write_kib, n_w = res["WRITE_SIZE"]
    key = "%s_%s_B%d_%s" % (args.scenario, args.dtype, args.envs, args.mode)
    traffic = {key: {
        "source_hash": bench_mod.source_hash(), "kernel": kern, "avg_ns_rocprof": float(top["AverageNs"]),
        "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib, "dispatches": [n_f, n_w],
        "hbm_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024 if fetch_kib is not None and write_kib is not None else None,
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE as reported; both KiB",
    }}
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic))

    # 4. VALU utilisation of the dominant kernel (its own pass: 5 SQ counters + 1 GRBM counter)
    names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    d = os.path.join(out, "valu")
    rc = run(["rocprofv3", "--pmc"] + names + ["--output-format", "csv", "-d", d, "--"] + short, os.path.join(out, "valu.log"), 200)
    print("VALU pass rc", rc)
    if rc != 0:
        sys.exit(rc)
    acc = {n: [] for n in names}
    for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] in acc and r["Kernel_Name"] == kern:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {n: (sum(v) / len(v) if v else None) for n, v in acc.items()}
    valu = {key: dict(avg, dispatches=len(acc["SQ_WAVES"]), source_hash=bench_mod.source_hash(), kernel=kern)}
    if avg["SQ_WAVES"] and avg["SQ_WAVE_CYCLES"]:
        # SQ_WAVE_CYCLES and SQ_ACTIVE_INST_VALU count in units of four cycles, summed over waves
        valu[key]["valu_insts_per_wave"] = avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]
        valu[key]["valu_issue_fraction_of_wave_lifetime"] = avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"]
        valu[key]["cycles_per_valu_inst_while_issuing"] = 4.0 * avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_INSTS_VALU"]
        valu[key]["mean_wave_lifetime_cycles"] = 4.0 * avg["SQ_WAVE_CYCLES"] / avg["SQ_WAVES"]
    json.dump(valu, open(os.path.join(out, "valu.json"), "w"), indent=1)
    print(json.dumps(valu))


if __name__ == "__main__":
    main()
