"""Host-side trace of the dense step with two host-driven contexts: when does each thread enter / leave prefilter() and align_hits_begin()?
usage: python tools/exp_cosched_trace.py [steps]  (SP_CO_SCHEDULE=0/1)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spacepharer_amd as sa

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nt = 1000000
tres, toff, qres, qoff = bench.make_workload(10000, nt, 200, 12345)
ctxs = []
for i in range(2):
    c = sa.SearchContext()
    if i == 0:
        c.set_targets_raw(tres.tobytes(), bench.u64p(toff), nt)
    else:
        c.share_targets(ctxs[0])
    c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
    ctxs.append(c)
for c in ctxs:
    for _ in range(2):
        c.prefilter(); c.align_hits_begin(); c.align_hits_end()
log = []
T0 = time.perf_counter()
def run(i):
    c = ctxs[i]
    for j in range(steps):
        a = time.perf_counter(); c.prefilter(); b = time.perf_counter(); c.align_hits_begin(); d = time.perf_counter()
        log.append((i, j, (a - T0) * 1e3, (b - T0) * 1e3, (d - T0) * 1e3))
th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
for c in ctxs: c.align_hits_end(); c.sync()
dt = time.perf_counter() - t0
for r in sorted(log, key=lambda r: r[2]):
    print("ctx %d step %d: prefilter %8.3f -> %8.3f (%.3f ms)  align -> %8.3f (%.3f ms)" % (r[0], r[1], r[2], r[3], r[3] - r[2], r[4], r[4] - r[3]))
print("SP_CO_SCHEDULE=%s: %.3f ms per step" % (os.environ.get("SP_CO_SCHEDULE", "default"), dt * 1e3 / (2 * steps)))
