#!/bin/bash
# kernel times of the device-side text on the dense and the sparse workload
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for nt in ${NT:-1000000 50000}; do
rm -rf /tmp/pt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python3 $R/tools/text_time.py 100000 $nt 2>&1 | grep spacers
python3 - <<'P'
import csv
rows = list(csv.DictReader(open('/tmp/pt/t_kernel_stats.csv')))
for r in rows[:16]:
    print("  %-44s calls %4s avg %10.1f us  %5s %%" % (r['Name'].replace('sp::','')[:44], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
done
