cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sw
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sw -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_sw2 -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in ("pmc_sw","pmc_sw2"):
    f=glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True)[0]
    rows=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_sw_byte1p" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); rows[int(r["Dispatch_Id"])]["us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k in sorted(rows)[-2:]:
        print(k, {a:round(b) for a,b in rows[k].items()})
PY
