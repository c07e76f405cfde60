"""One GPU as shard 0 of 8 at the REAL density (2e8 residues in all, 2.5e7 per shard; SPACERS spacers, default 20 000): the plain target-sharded pass (k_match_grp<true> on the
shard) against the probe-list cut (enumerate 1/8 of the queries against the union map, fetch what the eight enumerating ranks address to this shard) -- bench.probe_cut_proxy
on the dense workload, plus the occupancies.  usage: SP_PROBE_PER_QUERY=512 python tools/probe_cut_dense.py [shards]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spacepharer_amd as sa  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spacers = int(os.environ.get("SPACERS", "20000"))
args = argparse.Namespace(targets=int(os.environ.get("TARGETS", "1000000")), target_len=200)
out = bench.probe_cut_proxy(sa, args, shards=S, spacers=spacers)
print(json.dumps(out))
