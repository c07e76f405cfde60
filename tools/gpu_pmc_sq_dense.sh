#!/bin/bash
# instruction / wait / LDS counters of the dense matching kernel (10 000 spacers vs 2e8 residues); summary -> gpurun_out/pmc_sq_dense.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_I8"; do
  rm -rf /tmp/pmc_g
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_g -o p -- python3 $R/tools/pmc_workload.py dense > /tmp/pmc_g.log 2>&1 || tail -3 /tmp/pmc_g.log
  python3 - <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_g/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_match' in k or 'k_select' in k:
            acc[k.split('(')[0][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print('%s: ' % k + '  '.join('%s=%.1fM' % (c, sorted(v)[len(v)//2] / 1e6) for c, v in sorted(d.items())))
PY
done 2>&1 | tee $R/gpurun_out/pmc_sq_dense.txt
