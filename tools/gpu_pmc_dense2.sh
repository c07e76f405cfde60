# HBM traffic + cache counters of the dense matching kernels, old (SP_MATCH_DENSE2=0: k_match<512,false,false>) and new (k_match_dense), workload c; summary -> gpurun_out/pmc_dense2.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/pmc_dense2.txt
for v in ${VARIANTS:-0 1}; do
export SP_MATCH_DENSE2=$v
rm -rf $R/gpurun_out/pmc_d1 $R/gpurun_out/pmc_d2 $R/gpurun_out/pmc_d3
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_d1 -o p -- python3 $R/tools/pmc_workload.py dense 2>&1 | tail -2
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_d2 -o p -- python3 $R/tools/pmc_workload.py dense > /dev/null 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum --output-format csv -d $R/gpurun_out/pmc_d3 -o p -- python3 $R/tools/pmc_workload.py dense > /dev/null 2>&1
cd $R
echo "== SP_MATCH_DENSE2=$v" >> gpurun_out/pmc_dense2.txt
python3 - <<'P' | tee -a gpurun_out/pmc_dense2.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("gpurun_out/pmc_d1", "gpurun_out/pmc_d2", "gpurun_out/pmc_d3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:28]
            if "k_match" in k or "k_diag_read" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / max(1, n[(k, c)])) for c, v in acc[k].items()})
P
cd /tmp
done
rm -rf $R/gpurun_out/pmc_d1 $R/gpurun_out/pmc_d2 $R/gpurun_out/pmc_d3
