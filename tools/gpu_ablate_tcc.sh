# GPU box: L2 request / miss accounting of k_match_grp by ablation (developer builds -DSP_ABLATE=1..4, make ablate): which part of the kernel asks for which lines
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libspacepharer_amd.so libspacepharer_amd_ab1.so libspacepharer_amd_ab2.so libspacepharer_amd_ab3.so libspacepharer_amd_ab4.so}; do
  rm -rf /tmp/abt_$lib
  SP_LIB_NAME=$lib timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/abt_$lib -o p -- python3 $R/tools/pmc_workload.py > /dev/null 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys, collections
lib = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/abt_%s/**/*counter_collection.csv" % lib, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {k: sum(v.values()) / max(1, len(v)) for k, v in acc.items()}
print("%-28s " % lib + "  ".join("%s %.2fM" % (k.replace("_sum", ""), v / 1e6) for k, v in sorted(out.items())))
PY
done
