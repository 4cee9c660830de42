#!/usr/bin/env python3
"""Timeline of the LAST bench step in a rocprofv3 kernel trace (tools/gpu_kstats.sh leaves gpurun_out/kstats_quick/
r_kernel_trace.csv): every kernel's start offset, duration and the idle gap in front of it; totals.
usage: tools/trace_gaps.py [trace.csv] [first-kernel-pattern]"""
import csv
import sys
fn = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kstats_quick/r_kernel_trace.csv"
pat = sys.argv[2] if len(sys.argv) > 2 else "sfe6_dot"
rows = list(csv.DictReader(open(fn)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, k in enumerate(ks) if pat in k[2]]
if len(starts) < 3:
    sys.exit("pattern not found often enough")
a, b = starts[-2], starts[-1]            # the last complete step: from its first kernel to the next step's first kernel
t0 = ks[a][0]
busy, prev_end = 0, None
for s, e, n in ks[a:b]:
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  {name:46s} {(e - s) / 1e3:9.1f} us   gap {gap:7.1f}")
    busy += e - s
    prev_end = max(prev_end or e, e)
span = (ks[b][0] - t0) / 1e3
print(f"step span {span:.1f} us, kernels {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us (incl. the gap to the next step)")
