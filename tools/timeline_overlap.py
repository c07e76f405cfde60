"""From a rocprofv3 --kernel-trace CSV: per kernel, the median duration of launches that ran (mostly) beside a k_match of the OTHER queue
and of those that did not.  usage: python tools/timeline_overlap.py <dir with *kernel_trace.csv>"""
import collections
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp::", "").replace("void ", "")[:28], r.get("Queue_Id", "?")))
rows.sort()
rows = rows[len(rows) // 3:]      # steady state
match = [r for r in rows if r[2].startswith("k_match")]
beside, alone = collections.defaultdict(list), collections.defaultdict(list)
for s, e, k, q in rows:
    ov = sum(max(0, min(e, me) - max(s, ms)) for ms, me, mk, mq in match if mq != q)
    thr = 0.3 if k.startswith("k_match") else 0.7
    (beside if ov > thr * (e - s) else alone)[k].append((e - s) / 1e3)
print("%-28s %8s %6s %10s %6s" % ("kernel", "beside", "n", "not beside", "n"))
for k in sorted(set(beside) | set(alone)):
    b, a = beside.get(k, []), alone.get(k, [])
    print("%-28s %8.1f %6d %10.1f %6d" % (k, statistics.median(b) if b else float("nan"), len(b), statistics.median(a) if a else float("nan"), len(a)))
