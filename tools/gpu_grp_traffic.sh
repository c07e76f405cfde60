#!/bin/bash
# HBM-side traffic and L2 hit / miss counts of the matching kernel, grouped or not (separate --pmc passes)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in ${1:-0 1}; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    rm -rf /tmp/pmc_g
    SP_MATCH_GROUP=$g timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_g -o p -- python3 $R/tools/pmc_workload.py prefilter > /dev/null 2>&1
    python3 - "$g" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_g/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_match' in k or 'k_diag_read' in k:
            acc[k.split('(')[0][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print('group=%s %s: ' % (sys.argv[1], k) + '  '.join('%s=%.2fM' % (c, sorted(v)[len(v)//2] / 1e6) for c, v in sorted(d.items())))
PY
  done
done 2>&1 | tee $R/gpurun_out/grp_traffic.txt
