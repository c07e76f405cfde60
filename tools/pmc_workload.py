"""Workload for the rocprofv3 counter passes (run as `rocprofv3 ... -- python3 tools/pmc_workload.py`).

Three calibration launches of a streaming read with a known byte count per lane width (4 / 16 B per lane, 1 GiB each,
larger than the 256 MiB Infinity Cache), then 3 prefilter passes of the bench.py workload.  tools/pmc_summary.py turns
the counter CSV into per-launch numbers (profiles/rNN_pmc_traffic.json)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402

GIB = 1 << 30
DENSE = "dense" in sys.argv[1:]          # 10 000 spacers vs 2e8 residues (the survey's estimate of GenBank_phage_2018_09) instead of the bench proxy
tres, toff, qres, qoff = bench.make_workload(10000, 1000000 if DENSE else 50000, 200, 12345)
ctx = sa.SearchContext()
for width in (4, 16):
    for _ in range(3):
        ms = ctx.diag_stream_read(GIB, width)
    print("calibration read %2d B/lane: %.3f ms -> %.0f GB/s" % (width, ms, GIB / ms / 1e6))
ctx.set_targets_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1)
ctx.set_queries_raw(qres.tobytes(), bench.u64p(qoff), len(qoff) - 1)
stage = "full" if "full" in sys.argv[1:] else "prefilter"
for _ in range(3):
    ctx.prefilter()
    if stage == "full":
        ctx.hits_compact()
        ctx.align_hits_raw()
print("K=%d M=%d D=%d" % (ctx.stat("kmers"), ctx.stat("db_matches"), ctx.stat("diagonals")))
