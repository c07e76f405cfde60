#!/bin/bash
# GPU box: the device-side result text (tools/text_time.py) with its kernels' durations, for the proxy (100 000 spacers vs 1e7) and the dense index (10 000 vs 2e8) -> gpurun_out/${1:-text_prof}.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${1:-text_prof}.txt
for wl in "100000 50000" "10000 1000000"; do
  rm -rf $R/gpurun_out/txp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/txp -o t -- python3 $R/tools/text_time.py $wl 2>&1 | grep spacers | tail -1 >> $R/gpurun_out/${1:-text_prof}.txt
  python3 - $R/gpurun_out/txp <<'PY' >> $R/gpurun_out/${1:-text_prof}.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("text", "aln_", "k_reorder", "k_pairs")):
            print("   %-44s calls %4s  avg %9.1f us" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $R/gpurun_out/txp
cat $R/gpurun_out/${1:-text_prof}.txt
