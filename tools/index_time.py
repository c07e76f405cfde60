"""index build time on the GPU box (HIP events around the whole device-side set-up of a target set): python tools/index_time.py [n_targets ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spacepharer_amd as sa
for nt in [int(a) for a in sys.argv[1:]] or [50000, 1000000]:
    tres, toff, qres, qoff = bench.make_workload(100, nt, 200, 12345)
    ctx = sa.SearchContext()
    ms, wall = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), nt)
        ctx.sync()
        wall.append((time.perf_counter() - t0) * 1e3); ms.append(ctx.stat("ms_index_build"))
    print("%s targets %d (%.0e residues): index build on the device %.2f ms (min of %s), set_targets wall %.1f ms; masked residues %d, index entries %d" %
          (os.environ.get("SP_LIB_NAME", "libspacepharer_amd.so"), nt, nt * 200, min(ms[1:]), ["%.2f" % x for x in ms], min(wall[1:]), ctx.stat("masked_residues"), ctx.stat("index_entries")), flush=True)
    ctx.close()
