# GPU box: kernel timeline of a few steady-state steps of the bench (rocprofv3 --kernel-trace), to see which kernels of the two
# search contexts actually overlap.  usage: bash tools/gpu_timeline.sh [extra bench flags]   -> gpurun_out/timeline.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python3 $R/bench.py --steps 20 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e "$@" > /dev/null 2>&1
cd $R
python3 - <<'P' | tee gpurun_out/timeline.txt
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp::", "").replace("void ", "")[:34], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# the last region: take the last 60 % of the k_match launches' time span, print ~3 steps of it
ms = [r for r in rows if r[2].startswith("k_match")]
t_from = ms[-8][0]
t_to = ms[-4][1]
sel = [r for r in rows if r[0] >= t_from and r[0] <= t_to]
queues = sorted({r[3] for r in sel})
print("queues:", queues)
for s, e, k, q, st in sel:
    col = queues.index(q)
    print("%9.1f us  %8.1f us  %s%-34s q%s" % ((s - t_from) / 1e3, (e - s) / 1e3, "    " * 0 + ("" if col == 0 else " " * 40), k, q))
# overlap accounting over the whole second half of the run
half = rows[len(rows) // 2][0]
ev = []
for s, e, k, q, st in rows:
    if s < half: continue
    kind = "match" if k.startswith("k_match") else "other"
    ev.append((s, 1, kind)); ev.append((e, -1, kind))
ev.sort()
act = collections.Counter(); last = None; acc = collections.Counter()
for t, d, kind in ev:
    if last is not None:
        key = "match x%d + other x%d" % (act["match"], min(act["other"], 3))
        acc[key] += t - last
    act[kind] += d; last = t
tot = sum(acc.values())
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print("%-24s %6.2f %%" % (k, 100.0 * v / tot))
P
