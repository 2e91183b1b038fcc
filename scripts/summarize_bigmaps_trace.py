#!/usr/bin/env python3
"""Per-shape kernel durations from rocprofv3's kernel trace of scripts/run_bigmaps_for_rocprof.py (the stats CSV lumps
every shape of one kernel name together): the runner launches 9 forward + 9 backward kernels per shape, in order; the
first launch of each (cold instruction cache, first touch of the inputs) is dropped.
usage: python scripts/summarize_bigmaps_trace.py gpurun_out/prof_r03_bigmaps/trace_kernel_trace.csv [B]"""
import csv, sys
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
shapes = [(16, 112), (24, 56), (40, 28), (128, 28), (64, 56)]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nfp::fwd_tile" in r["Kernel_Name"] or "nfp::bwd_tile" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == 18 * len(shapes), len(rows)
print("shape,kernel,launches,avg_us,algorithmic_MB,achieved_TBps,frac_of_8TBps")
for i, (C, S) in enumerate(shapes):
    part = rows[18 * i: 18 * (i + 1)]
    for kind, per_px in (("fwd_tile", C * 4 + 32), ("bwd_tile", 2 * C * 4 + 32)):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in part if kind in r["Kernel_Name"]][1:]
        us = sum(d) / len(d)
        mb = B * S * S * per_px / 1e6
        print(f"[{B};{C};{S};{S}],{kind},{len(d)},{us:.2f},{mb:.1f},{mb / us:.3f},{mb / us / 8.0:.3f}")
