"""End-to-end timing of the module drop-ins on MMseqs2 DB triples: `prefilter` then `align` with the workflow's command
lines on the bench.py workload.  Wall clock per module = process start of the module call to
result DB written (reading the DBs, target set preparation on the device, search, formatting, writing)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402
from test_gpu_modules import ALN_FLAGS, PREF_FLAGS, write_seq_db  # noqa: E402

n_spacers = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tres, toff, qres, qoff = bench.make_workload(n_spacers, 50000, 200, 12345)
T = [tres[int(toff[i]):int(toff[i + 1])].tobytes().decode() for i in range(len(toff) - 1)]
Q = [qres[int(qoff[i]):int(qoff[i + 1])].tobytes().decode() for i in range(len(qoff) - 1)]
with tempfile.TemporaryDirectory() as d:
    q, t, pref, aln = (os.path.join(d, x) for x in ("qdb", "tdb", "pref", "aln"))
    write_seq_db(q, Q)
    write_seq_db(t, T)
    for rep in range(2):
        for f in os.listdir(d):
            if f.startswith("pref") or f.startswith("aln"):
                os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        assert sa.module_prefilter([q, t, pref] + (PREF_FLAGS % 300).split()) == 0
        t1 = time.perf_counter()
        assert sa.module_align([q, t, pref, aln] + ALN_FLAGS.split()) == 0
        t2 = time.perf_counter()
        print("run %d: prefilter module %.3f s, align module %.3f s -> %.0f spacer-queries/s end to end (files in tmpfs/disk: %s)" %
              (rep, t1 - t0, t2 - t1, n_spacers / (t2 - t0), d))
    print("pref DB %d bytes, aln DB %d bytes" % (os.path.getsize(pref), os.path.getsize(aln)))
