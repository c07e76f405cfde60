cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o full -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e --no-dense 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'])"
python3 - <<'P'
import csv
for r in csv.DictReader(open('/tmp/pf/full_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('k_scan_counts','k_match_grp','k_select(','lookback','scan_impl')): print(r['Name'][:40], r['Calls'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs'])
P
