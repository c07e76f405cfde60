#!/bin/bash
# front-end counters of the dense matching kernel: instruction cache, instruction fetch latency, branches, scalar cache
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in ${PERCU:-3 1}; do
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVES" "SQ_WAIT_INST_LDS SQ_INSTS_WAVE32 SQ_ITEMS SQ_THREAD_CYCLES_VALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_FLAT" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_FLAT"; do
  rm -rf /tmp/pmc_g
  SP_MATCH_PERCU=$n timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_g -o p -- python3 $R/tools/pmc_workload.py dense > /tmp/pmc_g.log 2>&1 || { echo "pass failed: $pass"; grep -i "error\|invalid\|not" /tmp/pmc_g.log | head -3; }
  python3 - $n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_g/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_match' in k:
            acc[k.split('(')[0][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print('percu=%s %s: ' % (sys.argv[1], k) + '  '.join('%s=%.2fM' % (c, sorted(v)[len(v)//2] / 1e6) for c, v in sorted(d.items())))
PY
done
done 2>&1 | tee $R/gpurun_out/pmc_front_dense.txt
