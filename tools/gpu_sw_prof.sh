# kernel-time profile of the bench + the SQ / LDS counters of the SW forward kernel; summaries -> gpurun_out/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sw $R/gpurun_out/pmc_sw1 $R/gpurun_out/pmc_sw2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sw -o sw -- python3 $R/bench.py --steps 60 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e --no-dense > /dev/null 2>&1
cut -c1-160 $R/gpurun_out/prof_sw/sw_kernel_stats.csv | head -14
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_sw1 -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_sw2 -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
cd $R
python3 - <<'P'
import csv, glob, collections
for d in ("gpurun_out/pmc_sw1", "gpurun_out/pmc_sw2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "sw_fwd" in k or "sw_byte1p" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k, {c: round(v / max(1, n[(k, c)])) for c, v in acc[k].items()})
P
