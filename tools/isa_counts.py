#!/usr/bin/env python3
"""Counts instruction kinds per kernel in a gfx950 assembly dump of rt_backend.hip
(hipcc ... -S --cuda-device-only -o rt_backend.s): scalar loads, vector loads, scratch
traffic.  A wave-uniform record fetch that shows up as a vector load is the thing to look for.

    python3 tools/isa_counts.py /tmp/rt_backend.s
"""
import re
import sys

text = open(sys.argv[1]).read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\w+):", text, re.M)]
starts.append((len(text), "end"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    body = text[a:b]
    body = body[: body.find(".section") if ".section" in body else len(body)]

    def c(p):
        return len(re.findall(p, body))

    kinds = [("s_load", r"\ts_load"), ("global_load", r"\tglobal_load"), ("global_store", r"\tglobal_store"),
             ("scratch_ld", r"\tscratch_load"), ("scratch_st", r"\tscratch_store"), ("ds", r"\tds_"),
             ("s_waitcnt", r"\ts_waitcnt")]
    print(f"{name[18:60]:44s} lines {body.count(chr(10)):6d} " + " ".join(f"{k} {c(p):4d}" for k, p in kinds))
