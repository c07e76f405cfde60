R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/exp_sw
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/exp_sw -o t -- python3 $R/tools/exp_sw_scaling.py > $R/gpurun_out/exp_sw.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/exp_sw/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sw_byte1p' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
grid = [int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)) for r in rows]
# launches alternate forward / reverse; 4 iterations per n
for i in range(0, len(d), 8):
    fw = d[i:i + 8:2]; rv = d[i + 1:i + 8:2]
    print('grid %8d  forward %s us   reverse %s us' % (grid[i], ' '.join('%.0f' % x for x in fw), ' '.join('%.0f' % x for x in rv)))
PY
