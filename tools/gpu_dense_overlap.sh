#!/bin/bash
# GPU box: dense step, two host-driven contexts: which kernels run beside the other context's k_match, and how long they take then (timeline from rocprofv3)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mb in ${MB:-2 3}; do
  rm -rf $R/gpurun_out/dov$mb
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dov$mb -o tl -- python3 $R/bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --steps 10 --regions 2 --warmup 2 --pipeline 2 --host-driven --match-blocks $mb > /dev/null 2>&1
  echo "== dense, host-driven x2, k_match blocks per CU: $mb"; python3 $R/tools/timeline_overlap.py $R/gpurun_out/dov$mb
  python3 - $R/gpurun_out/dov$mb <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sp::", "").replace("void ", "")[:22], r.get("Queue_Id", "?")))
rows.sort()
big = [r for r in rows if r[1] - r[0] > 30000]
t0 = big[-60][0]
for s, e, k, q in big[-60:]:
    print("%9.1f %9.1f  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, k))
PY
  rm -rf $R/gpurun_out/dov$mb
done 2>&1 | tee $R/gpurun_out/dense_overlap.txt
