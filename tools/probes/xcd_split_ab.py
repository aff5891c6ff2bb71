"""A/B of the uneven XCD deal on the box at hand: the three launches it applies to, with MPE_XCD_SPLIT unset and =0, in
fresh processes, twice each.   python3 tools/probes/xcd_split_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [("spread 64/64 x 2048", ["--agents", "64", "--landmarks", "64", "--envs", "2048", "--steps", "20", "--warmup", "5"]),
         ("spread 3/3 x 1048576", ["--envs", "1048576", "--steps", "20", "--warmup", "5"]),
         ("consensus x 65536", ["--scenario", "consensus", "--envs", "65536", "--steps", "20", "--warmup", "5"]),
         ("spread 3/3 x 65536 (no split at this size)", ["--steps", "100", "--warmup", "25"])]
import socket
print("host", socket.gethostname(), flush=True)
for name, args in CASES:
    row = []
    for rep in range(2):
        for split in ("1", "0"):
            env = dict(os.environ, MPE_XCD_SPLIT=split)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-step-mode", "--no-configs", "--no-sustained"] + args,
                                 env=env, capture_output=True, text=True, timeout=300).stdout
            d = json.loads(out.strip().splitlines()[-1])
            row.append("%s %.4f" % ("split" if split == "1" else "plain", d["roofline"]["launch_ms"]))
    print("%-45s %s" % (name, "  ".join(row)), flush=True)
