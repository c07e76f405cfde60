#!/bin/bash
# GPU box: kernel timeline (every launch, both streams of a context) of the headline step in the mode BENCH_FLAGS names, middle of the run -> gpurun_out/${1:-step_tl}.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stl
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/stl -o tl -- python3 $R/bench.py --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 20 --regions 2 --warmup 3 ${BENCH_FLAGS:---pipeline 1 --in-flight 2} > /dev/null 2>&1
python3 - $R/gpurun_out/stl <<'PY' > $R/gpurun_out/${1:-step_tl}.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp::", "").replace("void ", "")[:24], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20], "copy"))
rows.sort()
m = [i for i, r in enumerate(rows) if r[2].startswith("k_match")]
i0 = m[len(m) // 2]; i1 = m[min(len(m) - 1, len(m) // 2 + 3)]
t0 = rows[i0][0]
for s, e, k, q in rows[i0 - 6:i1 + 2]:
    print("%9.1f %9.1f %8.1f  %-6s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k))
PY
rm -rf $R/gpurun_out/stl
