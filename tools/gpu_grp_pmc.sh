#!/bin/bash
# instruction / wait counters of the matching kernel (grouped or not) on the bench workload: SP_MATCH_GROUP=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in ${1:-0 1}; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
    rm -rf /tmp/pmc_g
    SP_MATCH_GROUP=$g timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_g -o p -- python3 $R/tools/pmc_workload.py prefilter > /dev/null 2>&1
    python3 - "$g" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_g/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_match' in k:
            acc[k.split('(')[0][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print('group=%s %s: ' % (sys.argv[1], k) + '  '.join('%s=%.1fM' % (c, sorted(v)[len(v)//2] / 1e6) for c, v in sorted(d.items())))
PY
  done
done 2>&1 | tee $R/gpurun_out/grp_pmc.txt
