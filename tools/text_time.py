"""wall time of the device-side result text (sp_ctx_hits_text / sp_ctx_align_hits_text) on a bench workload; run under `rocprofv3 --kernel-trace --stats` for the kernels.
usage: python tools/text_time.py [n_spacers] [n_targets]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
tres, toff, qres, qoff = bench.make_workload(ns, nt, 200, 12345)
L = sa.load_library()
L.sp_host_alloc.restype = C.c_void_p
L.sp_host_alloc.argtypes = [C.c_uint64]
ctx = sa.SearchContext(evalue=200.0)
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), nt)
nq = len(qoff) - 1
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
cap = 400 << 20
text = L.sp_host_alloc(cap)
lens = C.cast(L.sp_host_alloc(4 * (nq + 1)), C.POINTER(C.c_uint32))
used = C.c_uint64()
for rep in range(3):
    t0 = time.perf_counter()
    ctx.prefilter()
    t1 = time.perf_counter()
    rc = L.sp_ctx_hits_text(ctx.ctx, text, cap, C.byref(used), lens)
    t2 = time.perf_counter()
    n1 = used.value
    rc2 = L.sp_ctx_align_hits_text(ctx.ctx, 0.0, 0, text, cap, C.byref(used), lens)
    t3 = time.perf_counter()
    print("%d spacers vs %.0e residues: prefilter %.2f ms, prefilter text %.2f ms (%.1f MB, rc %d), align + alignment text %.2f ms (%.1f MB, rc %d), hits %d"
          % (ns, nt * 200, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n1 / 1e6, rc, (t3 - t2) * 1e3, used.value / 1e6, rc2, ctx.num_hits()), flush=True)
ctx.close()
