# GPU box: from a kernel trace of the headline bench (four host-driven contexts): share of the steady-state wall time with 0, 1, 2, 3+ kernels running, and with / without a
# matching kernel running -- is the step bound by the GPU's throughput or are there gaps the host leaves?   -> gpurun_out/idle_share.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_idle
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_idle -o tl -- python3 $R/bench.py --steps 40 --warmup 5 --regions 3 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy "$@" > /tmp/tl_idle_bench.json 2>/dev/null
cd $R
python3 - <<'P' | tee gpurun_out/idle_share.txt
import csv, glob, json
rows = []
for f in glob.glob("/tmp/tl_idle/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ms = [r for r in rows if "k_match" in r[2]]
# inside a timed region of the four-context measurement (K steps = K matching launches): the 30 consecutive matching launches with the shortest span
best = min(range(0, len(ms) - 30), key=lambda i: ms[i + 29][1] - ms[i][0])
lo, hi = ms[best + 3][0], ms[best + 26][0]
NL = 23
ev = []
for s, e, k in rows:
    if e <= lo or s >= hi: continue
    ev.append((max(s, lo), 1, "k_match" in k)); ev.append((min(e, hi), -1, "k_match" in k))
ev.sort()
depth = 0; md = 0; last = lo; acc = {}; accm = {}
for t, d, m in ev:
    acc[min(depth, 4)] = acc.get(min(depth, 4), 0) + (t - last); accm[min(md, 3)] = accm.get(min(md, 3), 0) + (t - last)
    last = t; depth += d; md += d if m else 0
tot = hi - lo
print("window %.2f ms of the kernel trace: %d consecutive matching launches inside a timed region = %.3f ms per step" % (tot / 1e6, NL, tot / 1e6 / NL))
print("kernels running at once:  " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 4 else "", 100.0 * v / tot) for k, v in sorted(acc.items())))
print("matching kernels at once: " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 3 else "", 100.0 * v / tot) for k, v in sorted(accm.items())))
P
