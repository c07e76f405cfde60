#!/bin/bash
# round-4 profile set, part 2: shared tables, configs[4] with the reference beside it and on four ranks, dense counters, SW kernels, served laps
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04
O=gpurun_out/r04
{ echo "bench.py --cpu-sample 0 --no-module-e2e --no-dense: one staged target set shared by the four search contexts (default, sp_ctx_share_targets) / a copy per context (--unshared)";
  for rep in 1 2; do for u in "" "--unshared"; do echo "== bench.py $u (run $rep)"; timeout 600 python bench.py --cpu-sample 0 --no-module-e2e --no-dense $u 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); sm = d['step_modes']
        print('value %.2f M  ms_per_step %.4f  regions %s  k_match per launch (host-driven x4) %.4f ms, (device-driven x2) %.4f ms' % (d['value'] / 1e6, d['ms_per_step'], [round(x / 1e6, 2) for x in d['regions']['spacer_queries_per_s']], sm['host-driven x4']['k_match_ms_per_launch'], sm['device-driven x2']['k_match_ms_per_launch']))
"; done; done;
  echo "device memory in use (sp_device_memory) with 1 and 4 contexts holding the bench workload:"; python - <<'PY'
import ctypes as C, bench, spacepharer_amd as sa
L = sa.load_library()
def free():
    f, t = C.c_uint64(), C.c_uint64(); L.sp_device_memory(C.byref(f), C.byref(t)); return f.value
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
w = sa.SearchContext(); w.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1); w.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1); w.prefilter(); w.align_hits_raw(); w.close()
for shared in (True, False):
    f0 = free(); cs = []
    for i in range(4):
        c = sa.SearchContext()
        if i and shared: c.share_targets(cs[0])
        else: c.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
        c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1); c.prefilter(); c.align_hits_raw(); cs.append(c)
        if i in (0, 3): print("  %s: %d context(s): %.2f GB in use" % ("shared" if shared else "a copy each", i + 1, (f0 - free()) / 1e9))
    for c in cs: c.close()
PY
} > $O/shared_tables.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_s4; timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_s4 -o p -- python3 $R/bench.py --steps 40 --warmup 3 --regions 1 --cpu-sample 0 --no-module-e2e --no-dense --pipeline 4 --host-driven > /dev/null 2>&1
rm -rf /tmp/pmc_u4; timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_u4 -o p -- python3 $R/bench.py --steps 40 --warmup 3 --regions 1 --cpu-sample 0 --no-module-e2e --no-dense --pipeline 4 --host-driven --unshared > /dev/null 2>&1
cd $R
python3 - >> $O/shared_tables.txt <<'PY'
import csv, glob, collections
print("L2 counters of k_match_grp in the four-context host-driven run (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum; counter collection serialises the kernels, so these are per-launch cache counts, not the overlapped run's):")
for name, d in (("shared", "/tmp/pmc_s4"), ("a copy per context", "/tmp/pmc_u4")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_match' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print("  %-20s launches %d  TCC_MISS %.2f M  TCC_HIT %.2f M per launch" % (name, len(acc['TCC_MISS_sum']), sum(acc['TCC_MISS_sum']) / max(1, len(acc['TCC_MISS_sum'])) / 1e6, sum(acc['TCC_HIT_sum']) / max(1, len(acc['TCC_HIT_sum'])) / 1e6))
PY
timeout 1500 python tools/predictmatch_full.py --reference > $O/predictmatch_full.json 2> $O/predictmatch_full.err
timeout 1500 python tools/predictmatch_ranks.py --ranks 4 --same-gpu > $O/predictmatch_ranks4.json 2> $O/predictmatch_ranks4.err
timeout 1500 python tools/predictmatch_ranks.py --ranks 1 > $O/predictmatch_ranks1.json 2>> $O/predictmatch_ranks4.err
bash tools/gpu_pmc_dense.sh > $O/pmc_dense.txt 2>&1
bash tools/gpu_sw_prof.sh > $O/sw_prof.txt 2>&1
bash tools/e2e_served_laps.sh > $O/served_laps.txt 2>&1
ls -la $O
