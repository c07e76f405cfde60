#!/bin/bash
# L2 misses of the matching kernel under the XCD-affinity experiment of tools/exp_query_order.py (SP_EXP_XCD=1): does placing a query on the XCD that "owns" its first
# window's rows of the occupancy table change what the L2s miss?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_x
SP_EXP_XCD=1 timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_x -o p -- python3 $R/tools/exp_query_order.py none > /tmp/xcd_out.txt 2>&1
cat /tmp/xcd_out.txt | tail -4
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/pmc_x/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_match' in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value'])))
d = collections.defaultdict(dict)
for i, c, v in rows: d[i][c] = v
ids = sorted(d)
# three phases of 12 launches each: original order, all static random classes, all static classes by residue 3
for name, chunk in (("original order", ids[0:12]), ("all static, random class", ids[12:24]), ("all static, class by residue 3", ids[24:36])):
    if chunk:
        print("%-32s launches %d  TCC_MISS %.2f M  TCC_HIT %.2f M" % (name, len(chunk), sum(d[i]['TCC_MISS_sum'] for i in chunk) / len(chunk) / 1e6, sum(d[i]['TCC_HIT_sum'] for i in chunk) / len(chunk) / 1e6))
PY
