R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "--pipeline 3 --in-flight 1" "--pipeline 3 --host-driven"; do
  rm -rf $R/gpurun_out/ovx
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ovx -o tl -- python3 $R/bench.py --steps 30 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e $cfg > /tmp/line.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/line.json')); print('== $cfg: %.2f M/s %.4f ms/step' % (d['value']/1e6, d['ms_per_step']))"
  python3 $R/tools/timeline_overlap.py $R/gpurun_out/ovx
  python3 - <<P
import csv, glob
rows=[]
for f in glob.glob("$R/gpurun_out/ovx/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:20], r["Queue_Id"]))
rows.sort(); rows=rows[len(rows)//2:]
span=rows[-1][1]-rows[0][0]
busy={}
for s,e,k,q in rows: busy[q]=busy.get(q,0)+(e-s)
print("span %.1f ms; per-queue busy fraction:" % (span/1e6), {q: round(v/span,3) for q,v in busy.items()})
nm=sum(1 for r in rows if "k_match" in r[2]); print("k_match launches in span:", nm, " -> %.4f ms per step" % (span/1e6/nm))
P
done
