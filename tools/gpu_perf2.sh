cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_avr" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum" "FETCH_SIZE" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pm$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pm$i -o p -- python3 $R/tools/pmc_workload.py prefilter > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/pm[0-9]*")):
    fs=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not fs: print(d,"no output"); continue
    rows=collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        if "k_match" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); rows[int(r["Dispatch_Id"])]["us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k in sorted(rows)[-1:]:
        print({a:round(b,1) for a,b in rows[k].items()})
PY
