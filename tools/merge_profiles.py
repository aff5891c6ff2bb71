#!/usr/bin/env python3
"""Copy what tools/collect_all_profiles.sh left under gpurun_out/<tag>/prof_<name>/ into profiles/:
<tag>_kernel_stats_<name>.csv, <tag>_bench_<name>.json, <tag>_valu_<name>.json, and the entries of traffic.json.

    python3 tools/merge_profiles.py r02
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d)
    if not (d.startswith("prof_") and os.path.isdir(p)):
        continue
    name = d[len("prof_"):]
    for fn, out in (("kernel_stats.csv", "%s_kernel_stats_%s.csv"), ("bench.json", "%s_bench_%s.json"), ("valu.json", "%s_valu_%s.json")):
        if os.path.exists(os.path.join(p, fn)):
            shutil.copy(os.path.join(p, fn), os.path.join(dst, out % (tag, name)))
    tp = os.path.join(p, "traffic.json")
    if os.path.exists(tp):
        traffic.update(json.load(open(tp)))
    print("merged", name)
json.dump(traffic, open(tpath, "w"), indent=1)
