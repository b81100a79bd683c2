#!/usr/bin/env python3
"""Summarises the rocprofv3 outputs of tools/gpu_profile.sh: per kernel, the mean of
every collected counter per launch, the kernel-trace durations, and derived figures
(VALU lane utilisation, L2 hit rate, HBM bytes with the gfx950 FETCH_SIZE correction).

    python3 tools/summarize_pmc.py gpurun_out/prof_<tag> [--json out.json]
"""
from __future__ import annotations

import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    root = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    counters = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per dispatch
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
        for (disp, cname), val in per_dispatch.items():
            counters[names[disp]][cname].append(val)
    durations = {}
    # kernel durations: from the trace of the run with one frame in flight if there is one (every kernel
    # alone on the GPU, as under the counter passes), else from the default run's trace
    trace_dir = "trace_solo" if os.path.isdir(os.path.join(root, "trace_solo")) else "trace"
    for path in glob.glob(os.path.join(root, trace_dir, "**", "*_kernel_stats.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                durations[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                                 "percent": float(row["Percentage"])}
    summary = {}
    for kernel in sorted(set(counters) | set(durations)):
        if not kernel.startswith("k_"):
            continue
        c = {k: sum(v) / len(v) for k, v in counters.get(kernel, {}).items()}
        d = dict(durations.get(kernel, {}))
        d["counters"] = c
        derived = {}
        if c.get("SQ_ACTIVE_INST_VALU"):
            # lanes that did work per issued VALU instruction, out of 64
            derived["valu_lane_utilisation"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
            derived["valu_busy_per_sq_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
        if c.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                      "SQ_ACTIVE_INST_LDS"):
                if k in c:
                    derived[k.lower() + "_per_wave_cycle"] = c[k] / c["SQ_WAVE_CYCLES"]
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            derived["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
            fetch = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
            write = c.get("WRITE_SIZE", 0.0) * 1024.0
            derived["hbm_read_bytes"] = fetch
            derived["hbm_write_bytes"] = write
            if d.get("avg_us"):
                derived["hbm_gbs"] = (fetch + write) / (d["avg_us"] * 1e-6) / 1e9
        d["derived"] = derived
        summary[kernel] = d
    for kernel, d in summary.items():
        print(f"== {kernel}: {d.get('calls', '?')} calls, avg {d.get('avg_us', float('nan')):.1f} us, {d.get('percent', 0):.1f} %")
        for k, v in sorted(d["counters"].items()):
            print(f"     {k:28s} {v:16.1f}")
        for k, v in sorted(d["derived"].items()):
            print(f"   * {k:34s} {v:14.4f}")
    if out_json:
        with open(out_json, "w") as f:
            json.dump(summary, f, indent=1)
            f.write("\n")
    if "--traffic-json" in sys.argv:
        # HBM bytes per launch of the kernel with the most device time, for bench.py's roofline.traffic
        path = sys.argv[sys.argv.index("--traffic-json") + 1]
        workload = sys.argv[sys.argv.index("--workload") + 1]
        timed = {k: d for k, d in summary.items() if d.get("avg_us") and "hbm_read_bytes" in d["derived"]}
        if timed:
            dom = max(timed, key=lambda k: timed[k]["avg_us"] * timed[k]["calls"])
            try:
                with open(path) as f:
                    table = json.load(f)
            except Exception:
                table = {}
            dd = timed[dom]["derived"]
            cc = timed[dom]["counters"]
            # a frame = one launch of the resolve kernel (frames of up to 16 Mi samples: one batch)
            frames = max(summary.get("k_resolve", {}).get("calls", 0), 1)
            frame_bytes = 0.0
            launches_per_frame = {}
            for k, d in timed.items():
                per_frame = d["calls"] / frames
                launches_per_frame[k] = per_frame
                frame_bytes += (d["derived"]["hbm_read_bytes"] + d["derived"]["hbm_write_bytes"]) * per_frame
            table[workload] = {"kernel": dom, "hbm_read_bytes": dd["hbm_read_bytes"], "hbm_write_bytes": dd["hbm_write_bytes"],
                               "valu_wave_insts": cc.get("SQ_INSTS_VALU"),
                               "scalar_insts": (cc.get("SQ_INSTS_SALU", 0.0) + cc.get("SQ_INSTS_SMEM", 0.0) + cc.get("SQ_INSTS_BRANCH", 0.0)) or None,
                               "waves": cc.get("SQ_WAVES"),
                               # busy and wait counters as collected (bench.py's roofline_issue / roofline_l2 are computed from these)
                               "valu_active_quad_cycles": cc.get("SQ_ACTIVE_INST_VALU"), "salu_cycles": cc.get("SQ_INST_CYCLES_SALU"),
                               "smem_cycles": cc.get("SQ_INST_CYCLES_SMEM"), "branch_insts": cc.get("SQ_INSTS_BRANCH"),
                               "wait_any": cc.get("SQ_WAIT_ANY"), "wave_cycles": cc.get("SQ_WAVE_CYCLES"), "tcc_req": cc.get("TCC_REQ_sum"),
                               "scalar_cache_hit_rate": (cc["SQC_DCACHE_HITS"] / cc["SQC_DCACHE_REQ"]) if cc.get("SQC_DCACHE_REQ") else None,
                               "frame_hbm_bytes": frame_bytes, "launches_per_frame": launches_per_frame,
                               "durations_from": trace_dir,
                               "bytes": dd["hbm_read_bytes"] + dd["hbm_write_bytes"], "avg_us": timed[dom]["avg_us"],
                               "hbm_gbs": (dd["hbm_read_bytes"] + dd["hbm_write_bytes"]) / (timed[dom]["avg_us"] * 1e-6) / 1e9,
                               "l2_hit_rate": dd.get("l2_hit_rate"), "valu_lane_utilisation": dd.get("valu_lane_utilisation"),
                               "source": root}
            with open(path, "w") as f:
                json.dump(table, f, indent=1)
                f.write("\n")


if __name__ == "__main__":
    main()
