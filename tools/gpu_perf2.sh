cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['index_build_s'])"; done
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import bench, spacepharer_amd as sa
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
ctx = sa.SearchContext()
for i in range(3):
    t0=time.perf_counter(); ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff)-1); t1=time.perf_counter()
    print("set_targets %.3f s" % (t1-t0), {k: round(ctx.stat(k),2) for k in ("ms_ext_table","ms_index_build","ms_mask_gpu","ms_index_gpu","index_entries","masked_residues")})
PY
SP_DEBUG=1 timeout 300 python tools/host_prof.py 2>&1 | grep "^prefilter" | tail -3
