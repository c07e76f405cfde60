#!/bin/bash
# a few counter passes of the dense matching kernel: PASSES="A B C;D E" PERCU="3" LIB=...
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra PP <<< "${PASSES:-TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum;SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INSTS_SALU}"
for n in ${PERCU:-3}; do
for pass in "${PP[@]}"; do
  rm -rf /tmp/pmc_g
  SP_MATCH_PERCU=$n SP_LIB_NAME=${LIB:-libspacepharer_amd.so} timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_g -o p -- python3 $R/tools/pmc_workload.py ${WLARG:-dense} > /tmp/pmc_g.log 2>&1 || { echo "pass failed: $pass"; grep -i "error\|invalid" /tmp/pmc_g.log | head -3; }
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
done 2>&1 | tee -a $R/gpurun_out/pmc_quick.txt
