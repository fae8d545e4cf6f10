"""Timeline of the last iteration of a traced run: python tools/trace_timeline.py <dir with *_kernel_trace.csv> <first kernel of a step>
Prints start (relative to the step's first kernel), duration and stream/queue of every kernel of the last step --
used to see which kernels really overlap (profiles/r04/experiments.md)."""
import csv
import glob
import os
import sys

d, first = sys.argv[1], sys.argv[2]
cands = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
f = max(cands, key=os.path.getsize)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"] == first]
if len(starts) < 2:
    sys.exit("kernel %s not found twice" % first)
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f ms  +%8.3f ms  queue %-4s %s" % ((s - t0) * 1e-6, (e - s) * 1e-6, r.get("Queue_Id", "?"), r["Kernel_Name"]))
print("step: %.3f ms" % ((int(rows[b]["Start_Timestamp"]) - t0) * 1e-6))
