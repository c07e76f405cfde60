#!/bin/bash
# GPU box: full kernel timeline (every launch) of the dense step with two host-driven contexts, middle of the run -> gpurun_out/dense_tl.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dtl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dtl -o tl -- python3 $R/bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 10 --regions 2 --warmup 2 --pipeline 2 --host-driven ${BENCH_FLAGS} > /dev/null 2>&1
python3 - $R/gpurun_out/dtl <<'PY' > $R/gpurun_out/${1:-dense_tl}.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp::", "").replace("void ", "")[:22], r.get("Queue_Id", "?")))
rows.sort()
m = [i for i, r in enumerate(rows) if r[2].startswith("k_match")]
i0 = m[len(m) // 2 - 3]; i1 = m[len(m) // 2 + 3]
t0 = rows[i0][0]
for s, e, k, q in rows[i0 - 5:i1 + 30]:
    print("%9.1f %9.1f %8.1f  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k))
PY
rm -rf $R/gpurun_out/dtl
