# GPU box: rocprofv3 kernel stats of the bench step in two modes (device-driven x2, host-driven x3) -> gpurun_out/stats_<mode>.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in "dev2:--pipeline 2" "host3:--pipeline 4 --host-driven"; do
  name=${m%%:*}; flags=${m#*:}
  rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o s -- python3 $R/bench.py --steps 100 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e $flags 2>/dev/null | grep "^{" | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value']/1e6, d['ms_per_step'])"
  cp /tmp/st_$name/s_kernel_stats.csv $R/gpurun_out/stats_$name.csv
done
