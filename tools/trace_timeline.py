"""Timeline of the last iteration of a traced run (rocprofv3 --kernel-trace [--memory-copy-trace] --output-format csv):
    python tools/trace_timeline.py <output dir> <first kernel of a step> [--copies]
Prints start (relative to the step's first event), duration and queue of every kernel -- and, with --copies, every memory
copy -- of the last step: which kernels really overlap, where the device waits for the host (profiles/r04/experiments.md)."""
import csv
import glob
import os
import sys

d, first = sys.argv[1], sys.argv[2]
copies = "--copies" in sys.argv
cands = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
f = max(cands, key=os.path.getsize)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "queue %-3s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"]), r["Kernel_Name"])
        for r in csv.DictReader(open(f))]
if copies:
    m = f.replace("_kernel_trace.csv", "_memory_copy_trace.csv")
    if os.path.exists(m):
        for r in csv.DictReader(open(m)):
            b = r.get("Bytes") or r.get("Size") or "?"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy      %s %s bytes" % (r.get("Direction", ""), b), "copy"))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[3] == first]
if len(starts) < 2:
    sys.exit("kernel %s not found twice" % first)
a, b = starts[-2], starts[-1]
if copies:      # the step begins with the copies before its first kernel (back to the previous step's last kernel)
    while a > 0 and rows[a - 1][3] == "copy":
        a -= 1
    while b > 0 and rows[b - 1][3] == "copy":
        b -= 1
t0 = rows[a][0]
for s, e, text, _ in rows[a:b]:
    print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) * 1e-6, (e - s) * 1e-6, text))
print("step: %.3f ms" % ((rows[b][0] - t0) * 1e-6))
