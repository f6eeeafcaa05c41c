#!/usr/bin/env python3
"""Reading aid for a rocprofv3 --kernel-trace CSV: the last N kernel launches in start order with their durations and the
idle gap in front of each (end of the previous kernel -> start of this one).
    python scripts/experiments/trace_tail.py <rocprof output dir> <N>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
prev_end, busy, gaps = None, 0.0, 0.0
for r in rows[-n:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f'{(en - st) / 1e3:9.1f} us  gap {gap:6.1f} us  {r["Kernel_Name"][:70]}')
    busy += (en - st) / 1e3
    gaps += max(gap, 0.0) if prev_end is not None else 0.0
    prev_end = en
print(f"kernel time {busy:.1f} us, gaps {gaps:.1f} us")
