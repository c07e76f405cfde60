# SQ counters of the forward Smith-Waterman kernel per wave, on the dense (WL=dense) or the proxy workload: instructions by kind, busy / wait cycles -> gpurun_out/${1:-sw_pmc}.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sw1 $R/gpurun_out/pmc_sw2
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_sw1 -o p -- python3 $R/tools/pmc_workload.py full ${WL:-dense} > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_sw2 -o p -- python3 $R/tools/pmc_workload.py full ${WL:-dense} > /dev/null 2>&1
cd $R
python3 - <<'P' | tee gpurun_out/${1:-sw_pmc}.txt
import csv, glob, collections, os
tot = {}
for d in ("gpurun_out/pmc_sw1", "gpurun_out/pmc_sw2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if os.environ.get("KERNEL", "sw_fwd") in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        tot.setdefault(k, {}).update({c: v / max(1, n[(k, c)]) for c, v in acc[k].items()})
for k, v in tot.items():
    w = v.get("SQ_WAVES", 1.0)
    print(k, "waves per launch %d" % w)
    print("  per wave:", {c: round(x / w, 1) for c, x in sorted(v.items()) if c != "SQ_WAVES"})
P
rm -rf $R/gpurun_out/pmc_sw1 $R/gpurun_out/pmc_sw2
