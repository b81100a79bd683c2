#!/usr/bin/env python3
"""Reads a rocprofv3 *_kernel_trace.csv and prints, for the frames of the render, how the
wall time of a frame divides into kernel execution and the gaps between kernels.

    python3 tools/trace_gaps.py <kernel_trace.csv>
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
            name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:30]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    # frames start at each camera-ray kernel
    starts = [i for i, r in enumerate(rows) if r[2].startswith(("k_trace_packet<0>", "k_trace_fast<0>", "k_trace_closest"))]
    frames = [rows[a:b] for a, b in zip(starts, starts[1:])]
    frames = frames[len(frames) // 2:]  # the later (timed) frames
    if not frames:
        print("no frames found")
        return
    busy = defaultdict(float)
    gap_after = defaultdict(float)
    total = 0.0
    for fr, nxt in zip(frames, frames[1:]):
        seq = fr + [nxt[0]]
        for (s0, e0, n0), (s1, e1, n1) in zip(seq, seq[1:]):
            busy[n0] += (e0 - s0) / 1e3
            gap_after[n0] += (s1 - e0) / 1e3
        total += (nxt[0][0] - fr[0][0]) / 1e3
    n = len(frames) - 1
    print(f"{n} frames, {total / n:.1f} us per frame")
    for k in busy:
        print(f"  {k:22s} runs {busy[k] / n:8.1f} us, gap to the next kernel {gap_after[k] / n:7.1f} us")
    print(f"  kernels {sum(busy.values()) / n:.1f} us, gaps {sum(gap_after.values()) / n:.1f} us")


if __name__ == "__main__":
    main()
