#!/bin/bash
# the probe-list cut against the plain target-sharded pass on one GPU (tools/probe_cut_time.py) + per-kernel times of the same command
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_SWEEP" ]; then for S in 2 4 8; do timeout 600 python3 $R/tools/probe_cut_time.py $S; done 2>&1 | tee $R/gpurun_out/r05/probe_cut_time.txt; fi
for S in ${PROF_S:-8}; do
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python3 $R/tools/probe_cut_time.py $S > /dev/null 2>&1
python3 - <<P | tee -a $R/gpurun_out/r05/probe_cut_time.txt
import csv
rows = list(csv.DictReader(open('/tmp/pc/t_kernel_stats.csv')))
print("per-kernel times of tools/probe_cut_time.py $S (rocprofv3 --kernel-trace --stats):")
for r in [r for i, r in enumerate(rows) if i < 10 or any(k in r['Name'] for k in ('probe', 'scan', 'order', 'prep', 'score', 'cand'))]:
    print("  %-60s calls %4s avg %10.1f us  %5s %%" % (r['Name'].replace('sp::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
done
