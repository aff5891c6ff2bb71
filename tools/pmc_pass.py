#!/usr/bin/env python3
"""One rocprofv3 --pmc pass over a short bench.py run; prints the per-dispatch averages of the counters for
the rollout kernel of the scenario.  Keep each call within the hardware's limits (8 SQ/SQC counters per pass).

    python3 tools/pmc_pass.py simple_predator_prey 131072 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY ... [-- extra bench.py arguments]
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scn, envs, names = sys.argv[1], sys.argv[2], sys.argv[3:]
extra = []
if "--" in names:       # bench.py arguments after a "--", e.g. -- --agents 64 --landmarks 64
    extra = names[names.index("--") + 1:]
    names = names[:names.index("--")]
out = os.path.join(ROOT, "gpurun_out", "pmc_%s_%d" % (scn, os.getpid()))
os.environ["TMPDIR"] = "/tmp"
cmd = ["rocprofv3", "--pmc"] + names + ["--output-format", "csv", "-d", out, "--", "python3", "bench.py", "--no-cpu-baseline", "--no-step-mode", "--no-configs", "--no-sustained",
                                        "--scenario", scn, "--envs", envs, "--steps", "250", "--warmup", "25", "--target-steps", "250"] + extra
rc = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=200).returncode
if rc != 0:
    sys.exit("rocprofv3 failed with %d" % rc)
mode_re = re.compile(r"(?:>|Cfg), (?:true|false), 1, (?:true|false), (?:true|false), \d+(?:, (?:true|false))?>\(|consensus_kernel<\w+, 1(?:, (?:true|false))?>|wave_kernel<\w+, 1, 0|wave_stream_kernel<")
acc = {n: [] for n in names}
for fn in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] in acc and mode_re.search(r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({n: (sum(v) / len(v) if v else None) for n, v in acc.items()}))
