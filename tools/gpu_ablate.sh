# GPU box: instruction accounting of k_match by ablation (developer builds -DSP_ABLATE=1..3: no tail / no slot batch / no probe either).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libspacepharer_amd.so libspacepharer_amd_ab1.so libspacepharer_amd_ab2.so libspacepharer_amd_ab3.so libspacepharer_amd_ab4.so}; do
  rm -rf /tmp/ab_$lib
  SP_LIB_NAME=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/ab_$lib -o p -- python3 $R/tools/pmc_workload.py > /dev/null 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys, collections
lib = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("/tmp/ab_%s/**/*counter_collection.csv" % lib, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
dur = []
for f in glob.glob("/tmp/ab_%s/**/*kernel_trace.csv" % lib, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(v.values()) / max(1, len(v)) for k, v in acc.items()}
print("%-28s us %.0f  " % (lib, sorted(dur)[len(dur) // 2] if dur else -1) + "  ".join("%s %.1fM" % (k.replace("SQ_", ""), v / 1e6) for k, v in sorted(out.items())))
PY
done
