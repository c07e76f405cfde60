"""BASELINE.json configs[4] on N ranks (one process per GPU): the predictmatch chain of tools/predictmatch_full.py -- both searches, the set-level steps, FDR filtering
(data/predictmatch.sh:27-125) -- with the QUERIES partitioned by CRISPR array: rank r takes a contiguous range of arrays (DBReader::decomposeDomainByAminoAcid over the
arrays' peptide residues, sp_dist_decompose), searches their peptides against the whole target set and the control in its own HBM and runs the set-level chain for them
(sp_ctx_set_scores) -- a query set never leaves its rank, so nothing is exchanged until the per-array scores meet on rank 0 for filtermatchbyfdr (one ragged all-gather of
text entries through the library's communicator).  The match list equals the one-rank run's bytes (tests/test_gpu_predictmatch_ranks.py).

usage: python tools/predictmatch_ranks.py --ranks N [--same-gpu] [unique_genomes genome_nt replicas arrays spacers_per_array]   -> rank 0 prints one JSON line
       (without a launcher the script starts its N ranks itself; under torchrun it uses RANK / WORLD_SIZE / LOCAL_RANK)"""
import binascii
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


class FlatSlice:
    """entries [i0, i1) of a predictmatch_full.Flat"""

    def __init__(self, flat, i0, i1):
        lo, hi = int(flat.off[i0]), int(flat.off[i1])
        self.n = i1 - i0
        self.res = np.ascontiguousarray(flat.res[lo:hi])
        self.off = np.ascontiguousarray(flat.off[i0:i1 + 1] - flat.off[i0])

    def ptr(self):
        return C.cast(self.res.ctypes.data, C.c_char_p)

    def offs(self):
        return self.off.ctypes.data_as(C.POINTER(C.c_uint64))


def main():
    args = sys.argv[1:]
    ranks, same_gpu, keep = 1, False, None
    if "--ranks" in args:
        i = args.index("--ranks"); ranks = int(args[i + 1]); del args[i:i + 2]
    if "--same-gpu" in args:
        same_gpu = True; args.remove("--same-gpu")
    if "--dir" in args:
        i = args.index("--dir"); keep = args[i + 1]; del args[i:i + 2]
    sizes = [int(x) for x in args] + [900, 20000, 10, 5000, 20][len(args):]
    if "WORLD_SIZE" not in os.environ and ranks > 1:
        from spacepharer_amd import dist as spd
        uid = binascii.hexlify(spd.unique_id()).decode()
        shared = keep or tempfile.mkdtemp(prefix="pm_ranks_")
        os.makedirs(shared, exist_ok=True)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ranks", str(ranks), "--dir", shared] + (["--same-gpu"] if same_gpu else []) + [str(x) for x in sizes],
                                  env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ranks), SP_DIST_ID=uid)) for r in range(ranks)]
        sys.exit(max(abs(p.wait()) for p in procs))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    import predictmatch_e2e as pm
    import predictmatch_full as pf
    import spacepharer_amd as sa
    from spacepharer_amd import dist as spd
    L = sa.load_library()
    L.sp_init(1 << local)
    comm = spd.init_from_env(spd.TRANSPORT_HOST if same_gpu else spd.TRANSPORT_RCCL, rank=rank, world=world) if world > 1 else None
    U, genome_nt, replicas, n_arrays, per_array = sizes
    d = keep or tempfile.mkdtemp(prefix="pm_ranks_")
    os.makedirs(d, exist_ok=True)
    P = lambda x: os.path.join(d, x)      # noqa: E731
    times = {}
    t_all = time.perf_counter()
    genomes, arrays, truth = pm.make_data(U, genome_nt, n_arrays, per_array, seed=2024)
    if rank == 0:
        dbs = pf.prepare(d, genomes, arrays, replicas, 16, times)      # the target-side preparation runs once; every rank reads its DBs
        if comm: comm.barrier()
    else:
        comm.barrier()
        dbs = {}
        for name in ("q", "t", "c"):
            aa, nuc = pf.Flat(P(name)), pf.Flat(P(name + "_nucl_orf"))
            (sets,) = pf.read_table(P(name + "_nucl_orf.lookup"), [2])
            dbs[name] = {"aa": aa, "nucl": nuc, "set": sets.astype(np.uint32)}
    # ---- this rank's arrays: a contiguous range, balanced by peptide residues (+2 per peptide, the DB entry length)
    q = dbs["q"]
    qset = q["set"].astype(np.int64)
    assert (np.diff(qset) >= 0).all(), "the query ORFs are not in set order"
    per_set = np.bincount(qset, weights=np.diff(q["aa"].off).astype(np.float64) + 2, minlength=n_arrays).astype(np.uint64)
    a_from, a_cnt = spd.decompose_by_residues(np.maximum(per_set, 1), rank, world)
    i0, i1 = int(np.searchsorted(qset, a_from, "left")), int(np.searchsorted(qset, a_from + a_cnt, "left"))
    mine = {"aa": FlatSlice(q["aa"], i0, i1), "nucl": FlatSlice(q["nucl"], i0, i1), "set": np.ascontiguousarray(q["set"][i0:i1])}
    local_dbs = {"q": mine, "t": dbs["t"], "c": dbs["c"]}
    ctx = sa.SearchContext()
    if comm: comm.barrier()      # (every rank has its DBs in memory: the clock starts together)
    t0 = time.perf_counter()
    score, n_t = pf.search_and_score(ctx, local_dbs, "t", times)
    score_rev, n_c = pf.search_and_score(ctx, local_dbs, "c", times)
    t1 = time.perf_counter()
    assert all(a_from <= k < a_from + a_cnt for k in list(score) + list(score_rev))
    # ---- the per-array score entries meet on every rank (text; a few MB): ragged all-gather through the communicator
    def gather(entries):
        if comm is None:
            return dict(entries)
        blob = b"".join(b"%d\x01%s\x02" % (k, v if isinstance(v, bytes) else v.encode()) for k, v in sorted(entries.items()))
        parts = comm.allgatherv_host(np.frombuffer(blob, dtype=np.uint8))
        out = {}
        for p in parts:
            for item in p.tobytes().split(b"\x02")[:-1]:
                k, v = item.split(b"\x01", 1)
                out[int(k)] = v.decode()
        return out
    all_score, all_rev = gather(score), gather(score_rev)
    t2 = time.perf_counter()
    stats = comm.allreduce([float(n_t), float(n_c), float(i1 - i0)], "sum") if comm else [float(n_t), float(n_c), float(i1 - i0)]
    span = comm.allreduce([t1 - t0, t2 - t1], "max") if comm else [t1 - t0, t2 - t1]
    if rank == 0:
        pm.write_db(P("cScore"), [(k, v) for k, v in sorted(all_score.items())], 5)
        pm.write_db(P("cScore_rev"), [(k, v) for k, v in sorted(all_rev.items())], 5)
        sa.module_host("filtermatchbyfdr", [P("cScore"), P("cScore_rev"), P("match"), "--fdr", "0.05", "--report-fdr", "0", "--threads", "16", "--compressed", "0", "-v", "3"])
        match = pm.read_db(P("match"))
        t3 = time.perf_counter()
        found = other = 0
        for a, entry in match.items():
            for line in entry.splitlines():
                g = int(line.split("\t")[0])
                if truth[a] >= 0 and g % U == truth[a]: found += 1
                else: other += 1
        import zlib
        crc = 0
        for ext in ("", ".index"):
            crc = zlib.crc32(open(P("match") + ext, "rb").read(), crc)
        print(json.dumps({"config": "BASELINE configs[4], queries partitioned by array over %d rank(s)%s" % (world, " sharing one GPU" if same_gpu and world > 1 else ""),
                          "ranks": world, "spacers": n_arrays * per_array, "query_peptides": int(stats[2]), "target_orfs": dbs["t"]["aa"].n, "target_residues": int(dbs["t"]["aa"].off[-1]),
                          "alignments_target": int(stats[0]), "alignments_control": int(stats[1]), "two_searches_and_chain_s_max_over_ranks": round(span[0], 4), "score_exchange_s": round(span[1], 4),
                          "fdr_s": round(t3 - t2, 4), "whole_run_s": round(t3 - t_all, 4), "matches_reported": found + other, "true_matches_found": found, "other_matches": other,
                          "match_db_crc": "%08x" % crc, "exchange": "none until the per-array scores meet for filtermatchbyfdr; transport %s" % ("host sockets" if (same_gpu or world == 1) else "sp_dist (RCCL communicator up)")}))
    ctx.close()
    if comm:
        comm.barrier(); comm.close()


if __name__ == "__main__":
    main()
