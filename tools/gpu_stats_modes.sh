# GPU box: rocprofv3 kernel stats of the bench step in each mode on its own (device-driven x4, host-driven x4) -> gpurun_out/stats_<mode>.csv + the bench line
# of the same run gpurun_out/stats_<mode>.json (k_match per launch by HIP events inside the timed regions: must agree with the profiler's average)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in "device_driven_x2:--pipeline 2" "host_driven_x4:--pipeline 4 --host-driven"; do
  name=${m%%:*}; flags=${m#*:}
  rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o s -- python3 $R/bench.py --steps 100 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e --no-dense $flags 2>/dev/null | grep "^{" | tail -1 > $R/gpurun_out/stats_$name.json
  cp /tmp/st_$name/s_kernel_stats.csv $R/gpurun_out/stats_$name.csv
  python3 - $R/gpurun_out/stats_$name.json $R/gpurun_out/stats_$name.csv $name <<'PY'
import sys, json, csv
d = json.load(open(sys.argv[1]))
ev = d["roofline"]["algorithmic_bytes_per_launch"] / d["roofline"]["achieved"] / 1e6
for r in csv.DictReader(open(sys.argv[2])):
    if "k_match" in r["Name"]:
        print("%s: %.2f M spacer-queries/s, %.4f ms per step; k_match per launch: HIP events in the timed regions %.4f ms, rocprofv3 average over all %s launches %.4f ms (min %.4f)"
              % (sys.argv[3], d["value"] / 1e6, d["ms_per_step"], ev, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
PY
done
