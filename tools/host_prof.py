import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, spacepharer_amd as sa
tres, toff, qres, qoff = bench.make_workload(10000, 50000, 200, 12345)
ctx = sa.SearchContext()
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
nq = len(qoff) - 1
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
for it in range(6):
    t0 = time.perf_counter(); ctx.prefilter(); t1 = time.perf_counter(); ctx.hits_compact(); t2 = time.perf_counter(); ctx.align_hits_raw(); t3 = time.perf_counter()
    print("prefilter %.3f  hits_compact %.3f  align_hits %.3f ms | kernels: %s" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3,
          {k: round(ctx.stat(k), 3) for k in ("ms_prep","ms_match","ms_score","ms_select","ms_sw_forward","ms_sw_reverse","ms_traceback")}))
