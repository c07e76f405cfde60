# one counter pass focused on the SW kernels
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sw
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sw -o p -- python3 $R/tools/pmc_workload.py full > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_sw.json $(find gpurun_out/pmc_sw -name '*counter_collection.csv') > /dev/null
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_sw.json'))
k=d.get('kernels', d)
for n,v in k.items():
    if 'sw_' in n or 'match' in n: print(n, {a:(round(b) if isinstance(b,(int,float)) else b) for a,b in v.items()})
PY
