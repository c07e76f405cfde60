# GPU box: extra counter passes for k_match (issue mix, outstanding-request levels, instruction cache, texture addresser); one pass per group
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_VALU"; do      # (a TA_* / GRBM_* group hung the profiler on this pool until the gpurun limit: SQ / SQC groups only)
  i=$((i+1)); rm -rf $R/gpurun_out/pmcd$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcd$i -o p -- python3 $R/tools/pmc_workload.py prefilter > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcd*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_match" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f)
    for k, v in acc.items():
        n = 3      # launches of k_match in the workload
        print("   %-40s %.4g per launch" % (k, sum(v) / n))
PY
