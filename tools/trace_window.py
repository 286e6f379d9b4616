"""Print the kernel sequence of a rocprofv3 --kernel-trace CSV between the last two launches of a named kernel (by start time):
   python tools/trace_window.py <kernel_trace.csv> <from-substring> <to-substring>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
a, b = sys.argv[2], sys.argv[3]
end = max(i for i, r in enumerate(rows) if b in r["Kernel_Name"])
start = max(i for i, r in enumerate(rows[:end]) if a in r["Kernel_Name"])
print("columns:", list(rows[0].keys()))
prev = None; run = 0
for r in rows[start:end + 1]:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:60]
    key = (n, r.get("Queue_Id"), r.get("Stream_Id"))
    if key == prev:
        run += 1; continue
    if prev is not None: print("  x%-4d %-60s queue %s stream %s" % (run, prev[0], prev[1], prev[2]))
    prev, run = key, 1
print("  x%-4d %-60s queue %s stream %s" % (run, prev[0], prev[1], prev[2]))
