#!/usr/bin/env python3
"""Registers, spills and LDS of every small_kernel instantiation in a device-only object.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -c mpe_small_f32_core.hip -o /tmp/f32.co
    python3 tools/kernel_resources.py /tmp/f32.co [substring of the scenario's Cfg name]
"""
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin/"
co, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
elf = co + ".elf"
subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co,
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
notes = subprocess.run([LLVM + "llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
rows = []
for k in notes.split("- .agpr_count:")[1:]:
    g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, k).group(1)
    rows.append((g("name"), int(k.split()[0]), int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("sgpr_spill_count")),
                 int(g("group_segment_fixed_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.strip().split("\n")
print("scenario, CLAMP, MODE, FULL, INFO, EPW, NOISE | agpr vgpr(total) vgpr_spill sgpr_spill lds_bytes")
for r, d in sorted(zip(rows, names), key=lambda x: x[1]):
    m = re.match(r"void mpe::small_kernel<(\w+), mpe::(\w+)(?:<[^>]*>)?, (\w+), (\d), (\w+), (\w+), (\d+), (\w+)(?:, (\w+))?>", d)
    if m and want in m.group(2):
        print(", ".join(g for g in m.groups() if g is not None), "|", *r[1:])
