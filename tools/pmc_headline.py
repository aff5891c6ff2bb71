#!/usr/bin/env python3
"""rocprofv3 --pmc passes over the headline launch (simple_spread 3/3, 65 536 worlds, mpe_episode) for the write side of
the memory system: L2 (TCC) write requests to the fabric (EA) and their stalls by cause, L2 busy / tag / queue stalls,
the vector L1's (TCP) write requests, their latency and stalls, and the waves' own waiting.  One rocprofv3 run per pass
(counters are not split over passes; at most 4 TCC, 4 TCP, 2 TA, 8 SQ, 2 GRBM per run).

    python3 tools/pmc_headline.py <tag> [--lib path/to/libmpe_variant.so] [--envs N] [--scenario S] [-- extra bench.py args]

Writes gpurun_out/<tag>.json: per-dispatch averages of every counter over the rollout kernel's dispatches, the
per-XCC / per-instance values of the raw (un-summed) counters where rocprofv3's output carries them, and ratios.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum"],
    ["TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum", "TCC_EA0_WRREQ_LEVEL_sum"],
    ["TCC_BUSY_sum", "TCC_CYCLE_sum", "TCC_WRITE_sum", "TCC_REQ_sum"],
    ["TCC_TAG_STALL_sum", "TCC_SRC_FIFO_FULL_sum", "TCC_LATENCY_FIFO_FULL_sum", "TCC_IB_STALL_sum"],
    ["TCC_EA0_WRREQ_DRAM_sum", "TCC_NORMAL_WRITEBACK_sum", "TCC_STREAMING_REQ_sum", "TCC_HIT_sum"],
    ["TCP_TCC_WRITE_REQ_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum",
     "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"],
    ["TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_LFIFO_STALL_CYCLES_sum", "TCP_RFIFO_STALL_CYCLES_sum"],
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_VMEM",
     "GRBM_GUI_ACTIVE", "GRBM_EA_BUSY"],
    # raw counters (no _sum): rocprofv3 keeps their dimensions (XCC, instance) in its json output
    ["TCC_EA0_WRREQ_STALL", "TCC_EA0_WRREQ", "TCC_BUSY", "TCC_CYCLE"],
]
MODE_RE = re.compile(r"(?:>|Cfg), (?:true|false), 1, (?:true|false), (?:true|false), \d+(?:, (?:true|false))?>\(|consensus_kernel<\w+, 1(?:, (?:true|false))?>|wave_kernel<\w+, 1, 0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--envs", default="65536")
    ap.add_argument("--scenario", default="simple_spread")
    ap.add_argument("--collate", action="store_true", help="do not run anything: read the pass directories a run left in gpurun_out/")
    args, extra = ap.parse_known_args()
    extra = [a for a in extra if a != "--"]
    env = dict(os.environ, TMPDIR="/tmp")
    if args.lib:
        env["MPE_HIP_LIB"] = os.path.abspath(args.lib)
    bench = ["python3", "bench.py", "--no-cpu-baseline", "--no-step-mode", "--no-configs", "--no-sustained", "--scenario", args.scenario,
             "--envs", args.envs, "--steps", "250", "--warmup", "25", "--target-steps", "250"] + extra
    res, dims = {}, {}
    for k, names in enumerate(PASSES):
        raw = k == len(PASSES) - 1
        out = os.path.join(ROOT, "gpurun_out", "pmc_%s_pass%d" % (args.tag, k))
        cmd = ["rocprofv3", "--pmc"] + names + ["--output-format", "json" if raw else "csv", "-d", out, "--"] + bench
        if not args.collate:
            try:
                rc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120).returncode
            except subprocess.TimeoutExpired:
                sys.exit("pass %d timed out: stopping (no further GPU work)" % k)
            print("pass", k, names, "rc", rc, flush=True)
            if rc != 0:
                sys.exit("rocprofv3 failed with %d in pass %d" % (rc, k))
        if not raw:
            acc = {n: [] for n in names}
            for fn in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if r["Counter_Name"] in acc and MODE_RE.search(r["Kernel_Name"]):
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for n, v in acc.items():
                res[n] = sum(v) / len(v) if v else None
            res.setdefault("dispatches", len(next(iter(acc.values()))))
        else:
            for fn in glob.glob(os.path.join(out, "**", "*_results.json"), recursive=True):
                dims.update(per_dimension(json.load(open(fn)), names))
    r = lambda a, b: (res[a] / res[b]) if res.get(a) is not None and res.get(b) else None
    ratios = {
        "ea_wrreq_stall_cycles_per_wrreq": r("TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_sum"),
        "ea_wrreq_64B_share": r("TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_sum"),
        "tcc_busy_share_of_tcc_cycles": r("TCC_BUSY_sum", "TCC_CYCLE_sum"),
        "ea_wrreq_stall_share_of_tcc_cycles": r("TCC_EA0_WRREQ_STALL_sum", "TCC_CYCLE_sum"),
        "ea_dram_credit_stall_share_of_tcc_cycles": r("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_CYCLE_sum"),
        "ea_gmi_credit_stall_share_of_tcc_cycles": r("TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_CYCLE_sum"),
        "too_many_ea_wrreqs_stall_share_of_tcc_cycles": r("TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_CYCLE_sum"),
        "tag_stall_share_of_tcc_cycles": r("TCC_TAG_STALL_sum", "TCC_CYCLE_sum"),
        "ea_wrreq_mean_outstanding_per_channel": r("TCC_EA0_WRREQ_LEVEL_sum", "TCC_CYCLE_sum"),
        "tcp_write_latency_cycles_per_req": r("TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum"),
        "wave_wait_share_of_lifetime": r("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
        "vmem_wr_issue_share_of_lifetime": r("SQ_INST_CYCLES_VMEM_WR", "SQ_WAVE_CYCLES"),
        "ea_busy_share_of_gui_active": r("GRBM_EA_BUSY", "GRBM_GUI_ACTIVE"),
    }
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    out = {"tag": args.tag, "lib": args.lib or "in-tree", "scenario": args.scenario, "envs": int(args.envs), "source_hash": bench_mod.source_hash(),
           "per_dispatch": res, "ratios": ratios, "per_dimension": dims}
    dest = os.path.join(ROOT, "gpurun_out", "%s.json" % args.tag)
    os.makedirs(os.path.dirname(dest), exist_ok=True)
    json.dump(out, open(dest, "w"), indent=1)
    print(json.dumps({"ratios": ratios, "per_dimension_keys": list(dims)[:8]}))


def per_dimension(doc, names):
    """rocprofv3's json lists, per counter, its instances (DIMENSION_INSTANCE = the L2 channel, DIMENSION_XCC) and, per
    dispatch, one value per instance in that same order.  Returns {counter: {"per_xcc": [sum over the XCC's channels,
    averaged over the rollout kernel's dispatches], "channel_min_max": [..]}}; {} with an "error" entry when the layout
    is not the expected one."""
    try:
        top = doc["rocprofiler-sdk-tool"][0]
        kernels = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("demangled_kernel_name") or k.get("kernel_name", "")
                   for k in top["kernel_symbols"]}
        info = {}
        for c in top["counters"]:
            if c["name"] in names:
                xcc = [next(d["index"] for d in i["dimensions"] if d["dimension_name"] == "DIMENSION_XCC") for i in c["instances"]]
                info[c["id"]["handle"]] = (c["name"], xcc)
        acc, n_disp = {}, 0
        for rec in top["callback_records"]["counter_collection"]:
            kid = rec["dispatch_data"]["dispatch_info"]["kernel_id"]
            if not MODE_RE.search(kernels.get(kid, "")):
                continue
            n_disp += 1
            seen = {}
            for r in rec["records"]:
                h = r["counter_id"]["handle"]
                if h not in info:
                    continue
                k = seen.get(h, 0)
                seen[h] = k + 1
                name, xcc = info[h]
                a = acc.setdefault(name, [0.0] * len(xcc))
                a[k] += float(r["value"])
        out = {"dispatches": n_disp}
        for h, (name, xcc) in info.items():
            if name not in acc:
                continue
            vals = [v / max(1, n_disp) for v in acc[name]]
            per_xcc = [0.0] * (max(xcc) + 1)
            for v, x in zip(vals, xcc):
                per_xcc[x] += v
            out[name] = {"per_xcc": per_xcc, "channel_min": min(vals), "channel_max": max(vals), "channels": len(vals)}
        return out
    except (KeyError, IndexError, TypeError, ValueError, StopIteration) as e:
        return {"error": repr(e)}


if __name__ == "__main__":
    main()
