#!/bin/bash
# memory-side counters of the dense matching kernel at 1 / 2 / 3 blocks per CU: request counts by size, average read latency (LEVEL / REQ)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in ${PERCU:-1 2 3}; do
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum"; do
  rm -rf /tmp/pmc_g
  SP_MATCH_PERCU=$n timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_g -o p -- python3 $R/tools/pmc_workload.py dense > /tmp/pmc_g.log 2>&1 || tail -3 /tmp/pmc_g.log
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
done 2>&1 | tee $R/gpurun_out/pmc_mem_dense.txt
