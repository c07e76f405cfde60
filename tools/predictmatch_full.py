"""BASELINE.json configs[4] at its stated size on one MI355X: 100 000 spacers (5 000 CRISPR arrays x 20) against a 10x replicated phage
set of ~10^8 translated ORF residues AND against its codon-reversed control, the set-level chain, FDR filtering -- the path
data/predictmatch.sh:27-125 walks.  (The 8-GPU form of the config shards the same target set; tests/test_gpu_predictmatch_full.py
checks that 8 target shards reproduce the unsplit bytes.)

Nucleotide route, as the workflow has it: synthetic coding-dense genomes (tools/predictmatch_e2e.py: make_data) -> `replicas` copies
with distinct keys (each copy its own target set, like the config's replicated GenBank) -> extractorfs / translatenucs /
reverseseqbycodon through the host module drop-ins on DB triples -> the DBs loaded as flat buffers (numpy; no Python string per ORF)
-> per target set (targets, control): index on the device, ONE batch of all query peptides through prefilter + align, the chain
proteinaln2nucl ... combinescore in memory (sp_ctx_set_scores) -> filtermatchbyfdr on the two score DBs.

usage: python tools/predictmatch_full.py [unique_genomes] [genome_nt] [replicas] [arrays] [spacers_per_array]   -> one JSON line
       defaults 900 x 20 000 nt x 10 replicas (~10^8 ORF residues), 5 000 x 20 spacers"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import predictmatch_e2e as pm      # noqa: E402


def read_table(path, cols):
    """integer columns of a tab separated text file (index: key, offset, length; lookup: key, name, set)"""
    try:
        import pandas as pd
        df = pd.read_csv(path, sep="\t", header=None, usecols=cols, dtype=np.int64, engine="c")
        return [df[c].to_numpy() for c in cols]
    except ImportError:
        a = np.loadtxt(path, dtype=np.int64, usecols=cols, delimiter="\t", ndmin=2)
        return [a[:, i] for i in range(len(cols))]


class Flat:
    """a sequence DB as the C ABI takes it: residues back to back (entry terminators "\\n\\0" removed), offsets[n + 1]"""

    def __init__(self, path):
        key, off, length = read_table(path + ".index", [0, 1, 2])
        assert (key == np.arange(len(key))).all(), path + ": keys are not 0..n-1"
        assert (off[1:] == off[:-1] + length[:-1]).all() and off[0] == 0, path + ": entries are not contiguous in key order"
        data = np.fromfile(path, dtype=np.uint8)
        assert len(data) == int(off[-1] + length[-1])
        self.n = len(key)
        self.res = np.ascontiguousarray(data[(data != 10) & (data != 0)])
        self.off = np.zeros(self.n + 1, dtype=np.uint64)
        self.off[1:] = np.cumsum(length - 2).astype(np.uint64)
        assert int(self.off[-1]) == len(self.res), path + ": a sequence holds a newline or NUL"

    def ptr(self):
        return C.cast(self.res.ctypes.data, C.c_char_p)

    def offs(self):
        return self.off.ctypes.data_as(C.POINTER(C.c_uint64))

    def seq(self, i):
        return self.res[int(self.off[i]):int(self.off[i + 1])].tobytes().decode()


class LibFlat:
    """the same through the C ABI (sp_flat_db_open): the arrays stay in the library's memory"""

    def __init__(self, path):
        import spacepharer_amd as sa
        self.f = sa.FlatDb(path)
        self.n = self.f.n
        assert all(self.f.keys[i] == i for i in (0, self.n // 2, self.n - 1)), path + ": keys are not 0..n-1"
        self.off = np.ctypeslib.as_array(self.f.offsets, shape=(self.n + 1,))      # views of the library's arrays (no copy)
        self.res = np.ctypeslib.as_array(C.cast(self.f.residues, C.POINTER(C.c_uint8)), shape=(int(self.off[-1]),))

    def ptr(self):
        return self.f.residues

    def offs(self):
        return self.f.offsets

    def sets(self):
        assert self.f.sets, "no .lookup next to the DB"
        return np.ctypeslib.as_array(self.f.sets, shape=(self.n,)).astype(np.uint32)

    def seq(self, i):
        o0, o1 = int(self.f.offsets[i]), int(self.f.offsets[i + 1])
        return C.string_at(C.cast(self.f.residues, C.c_void_p).value + o0, o1 - o0).decode()


def u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


def prepare(d, genomes, arrays, replicas, threads, times):
    import spacepharer_amd as sa
    P = lambda x: os.path.join(d, x)      # noqa: E731
    th = ["--threads", str(threads), "--compressed", "0", "-v", "3"]
    t0 = time.perf_counter()
    pm.nucl_db(P("q_nucl"), arrays)
    pm.nucl_db(P("t_nucl"), [[g] for g in genomes] * replicas)      # copy r of genome g: contig key = set = r * len(genomes) + g
    t1 = time.perf_counter()
    sa.module_host("extractorfs", [P("q_nucl"), P("q_nucl_orf"), "--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"] + th)
    sa.module_host("translatenucs", [P("q_nucl_orf"), P("q")] + th)
    sa.module_host("extractorfs", [P("t_nucl"), P("t_nucl_orf"), "--min-length", "30", "--orf-start-mode", "0", "--create-lookup", "1"] + th)
    sa.module_host("translatenucs", [P("t_nucl_orf"), P("t")] + th)
    sa.module_host("reverseseqbycodon", [P("t_nucl_orf"), P("c_nucl_orf")] + th)
    sa.module_host("translatenucs", [P("c_nucl_orf"), P("c")] + th)
    t2 = time.perf_counter()
    dbs = {}
    for name in ("q", "t", "c"):
        # the library's own DB reader (sp_flat_db_open: memory-mapped, threaded) instead of pandas + numpy: 1.14 -> ~0.2 s for the six DBs of this configuration
        aa, nuc = LibFlat(P(name)), LibFlat(P(name + "_nucl_orf"))
        assert aa.n == nuc.n
        sets = nuc.sets()      # reverseseqbycodon / translatenucs link the lookup of the ORF DB
        assert len(sets) == aa.n
        dbs[name] = {"aa": aa, "nucl": nuc, "set": sets}
    t3 = time.perf_counter()
    times.update(write_input_dbs_python_s=t1 - t0, six_host_modules_s=t2 - t1, load_dbs_numpy_s=t3 - t2)
    return dbs


def search_and_score(ctx, dbs, tname, times, keep=False):
    """one search of ALL query peptides (one batch) against target set `tname` + the in-memory chain -> {query set: cScore entry}"""
    q, t = dbs["q"], dbs[tname]
    t0 = time.perf_counter()
    ctx.set_targets_raw(t["aa"].ptr(), t["aa"].offs(), t["aa"].n)
    t1 = time.perf_counter()
    ctx.set_queries_raw(q["aa"].ptr(), q["aa"].offs(), q["aa"].n)
    ctx.prefilter()
    t2 = time.perf_counter()
    ctx.reserve_results(ctx.num_hits())      # (page-locked result buffers: grown when this search has more hits than the one before)
    t2b = time.perf_counter()
    aln, cig, n = ctx.align_hits_raw()
    t3 = time.perf_counter()
    qs, qsp = u32(q["set"])
    ts, tsp = u32(t["set"])
    score = ctx.set_scores_raw((q["nucl"].ptr(), q["nucl"].offs()), (t["nucl"].ptr(), t["nucl"].offs()), qsp, tsp, aln, cig, stage=0)
    t4 = time.perf_counter()
    times[tname] = {"stage_targets_s": t1 - t0, "prefilter_s": t2 - t1, "align_s": t3 - t2, "of_which_pinning_result_buffers_s": t2b - t2, "set_scores_s": t4 - t3, "alignments": n,
                    "kernel_ms": {k: ctx.stat("ms_" + k) for k in ("match", "sw_forward")},
                    "counters": {k: ctx.stat(k) for k in ("kmers", "db_matches", "diagonals", "hits")}}
    return score, n


def reference_timing(d, dbs, threads=16, sample=24000):
    """The reference on the same inputs, on this box's host cores (oracle/_ref: the reference's own translation units compiled by oracle/Makefile): its six target- /
    query-side modules on the same DB triples (ref_sets), and its search loops (ref_harness --bench: Prefiltering::runSplit + Alignment::run as the reference's classes
    run them, index build and k-mer tables reported separately) on a bounded SAMPLE of the query peptides against the whole target set -- scaled to all peptides and
    doubled for the control search, whose set is as large.  The set-level chain behind the searches is not timed for the reference (it needs the alignment DBs as files)."""
    import subprocess
    sets_bin, harness = os.path.join(ROOT, "oracle", "_ref", "ref_sets"), os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not (os.path.exists(sets_bin) and os.path.exists(harness)):
        return {"error": "oracle/_ref is not built"}
    P = lambda x: os.path.join(d, x)      # noqa: E731
    out = {"threads": threads, "modules_s": {}}
    th = ["--threads", str(threads)]
    for name in ("q_nucl", "t_nucl"):      # the reference's extractorfs reads the header DB beside the sequence DB (createdb writes both)
        if not os.path.exists(P(name + "_h")):
            keys = read_table(P(name + ".index"), [0])[0]
            pm.write_db(P(name + "_h"), [(int(k), "c%d\n" % int(k)) for k in keys], 12)
    for name, args in (("extractorfs q", ["extractorfs", P("q_nucl"), P("ref_q_orf"), "--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"]),
                       ("translatenucs q", ["translatenucs", P("ref_q_orf"), P("ref_q")]),
                       ("extractorfs t", ["extractorfs", P("t_nucl"), P("ref_t_orf"), "--min-length", "30", "--orf-start-mode", "0", "--create-lookup", "1"]),
                       ("translatenucs t", ["translatenucs", P("ref_t_orf"), P("ref_t")]),
                       ("reverseseqbycodon", ["reverseseqbycodon", P("ref_t_orf"), P("ref_c_orf")]),
                       ("translatenucs c", ["translatenucs", P("ref_c_orf"), P("ref_c")])):
        t0 = time.perf_counter()
        r = subprocess.run([sets_bin] + args + th, capture_output=True, text=True)
        out["modules_s"][name] = round(time.perf_counter() - t0, 4)
        if r.returncode != 0:
            out["modules_s"][name] = "failed: " + (r.stderr or r.stdout)[-200:]
    out["six_modules_s"] = round(sum(v for v in out["modules_s"].values() if isinstance(v, float)), 4)
    # the search loops on a sample of the peptides
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import matrix_text
    q, t = dbs["q"]["aa"], dbs["t"]["aa"]
    n_s = min(sample, q.n)
    pick = np.linspace(0, q.n - 1, n_s).astype(np.int64)
    tf, qf, of = P("ref_t.txt"), P("ref_q.txt"), P("ref_o.json")
    open(P("VTML80.out"), "w").write(matrix_text("VTML80")); open(P("VTML40.out"), "w").write(matrix_text("VTML40"))
    with open(tf, "wb") as f:
        res = t.res.tobytes()
        f.write(b"\n".join(res[int(t.off[i]):int(t.off[i + 1])] for i in range(t.n)) + b"\n")
    with open(qf, "wb") as f:
        res = q.res.tobytes()
        f.write(b"\n".join(res[int(q.off[i]):int(q.off[i + 1])] for i in pick) + b"\n")
    t0 = time.perf_counter()
    r = subprocess.run([harness, P("VTML80.out"), P("VTML40.out"), tf, qf, of, "--bench-reps", "1", "--bench-threads", str(threads)], capture_output=True, text=True)
    out["harness_wall_s"] = round(time.perf_counter() - t0, 3)
    if r.returncode != 0:
        out["search"] = {"error": (r.stderr or r.stdout)[-300:]}
        return out
    b = json.load(open(of))["bench"]
    scale = q.n / float(n_s)
    out["search"] = {"sample_peptides": n_s, "of": q.n, "index_build_s": b["index_build_s"], "kmer_tables_s": b["table_s"], "prefilter_s_sample": b["prefilter_s"], "align_s_sample": b["align_s"],
                     "pairs_sample": b["pairs"], "search_loops_s_all_peptides_one_set": round((b["prefilter_s"] + b["align_s"]) * scale, 3),
                     "one_search_module_pair_s_estimate": round(b["index_build_s"] + b["table_s"] + (b["prefilter_s"] + b["align_s"]) * scale, 3)}
    out["two_searches_s_estimate"] = round(2 * out["search"]["one_search_module_pair_s_estimate"], 3)
    out["prep_plus_two_searches_s_estimate"] = round(out["six_modules_s"] + out["two_searches_s_estimate"], 3)
    return out


def run_pipelined(d, genomes, arrays, replicas, threads, n_arrays, hits_hint=0, overlap_scores=True):
    """The same chain with its independent steps overlapped (host threads; every ctypes call releases the interpreter lock): the query, target and control preparations run
    beside each other, a DB is read as soon as its module has written it, the target search starts when query and target peptides are in memory, its set-level scores are
    computed on host threads while the GPU searches the control set (two search contexts).  Same modules, same calls, same files: only their order in time differs.
    -> (score, score_rev, n_t, n_c, match, wall seconds from the first module call to the match DB read back, marks)"""
    import spacepharer_amd as sa
    from concurrent.futures import ThreadPoolExecutor
    P = lambda x: os.path.join(d, x)      # noqa: E731
    th = ["--threads", str(threads), "--compressed", "0", "-v", "3"]
    pm.nucl_db(P("q_nucl"), arrays)
    pm.nucl_db(P("t_nucl"), [[g] for g in genomes] * replicas)
    ex = ThreadPoolExecutor(12)
    marks = {}
    t0 = time.perf_counter()

    def mark(k):
        marks[k] = round(time.perf_counter() - t0, 4)

    def make_ctx():
        c = sa.SearchContext(threads=max(4, threads - 4))      # (its set-level scores run beside the other context's search: leave the search's host thread a share of the CPU quota)
        if hits_hint:
            c.reserve_results(hits_hint)      # (page-locked result buffers of the scale the job is known to have: pinned beside the first modules, not inside the first align)
        return c
    ctx_f = [ex.submit(make_ctx), ex.submit(make_ctx)]      # tables on the device: beside the first modules

    def load(name):
        fa, fn = ex.submit(LibFlat, P(name)), ex.submit(LibFlat, P(name + "_nucl_orf"))
        aa, nuc = fa.result(), fn.result()
        assert aa.n == nuc.n
        sets = nuc.sets()
        mark("loaded_" + name)
        return {"aa": aa, "nucl": nuc, "set": sets}

    def chain_q():
        sa.module_host("extractorfs", [P("q_nucl"), P("q_nucl_orf"), "--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"] + th)
        sa.module_host("translatenucs", [P("q_nucl_orf"), P("q")] + th)
        return load("q")

    def chain_c():
        sa.module_host("reverseseqbycodon", [P("t_nucl_orf"), P("c_nucl_orf")] + th)
        sa.module_host("translatenucs", [P("c_nucl_orf"), P("c")] + th)
        return load("c")

    fq = ex.submit(chain_q)
    sa.module_host("extractorfs", [P("t_nucl"), P("t_nucl_orf"), "--min-length", "30", "--orf-start-mode", "0", "--create-lookup", "1"] + th)
    fc = ex.submit(chain_c)
    sa.module_host("translatenucs", [P("t_nucl_orf"), P("t")] + th)
    dbs = {"t": load("t")}
    dbs["q"] = fq.result()

    def search(ctx, tname):
        q, t = dbs["q"], dbs[tname]
        ctx.set_targets_raw(t["aa"].ptr(), t["aa"].offs(), t["aa"].n)
        ctx.set_queries_raw(q["aa"].ptr(), q["aa"].offs(), q["aa"].n)
        ctx.prefilter()
        aln, cig, n = ctx.align_hits_raw()
        mark("searched_" + tname)
        return aln, cig, n

    def scores(ctx, tname, res):
        q, t = dbs["q"], dbs[tname]
        aln, cig, n = res
        qs, qsp = u32(q["set"])
        ts, tsp = u32(t["set"])
        out = ctx.set_scores_raw((q["nucl"].ptr(), q["nucl"].offs()), (t["nucl"].ptr(), t["nucl"].offs()), qsp, tsp, aln, cig, stage=0)
        mark("scored_" + tname)
        return out

    ctx_t, ctx_c = ctx_f[0].result(), ctx_f[1].result()
    res_t = search(ctx_t, "t")
    if overlap_scores:      # the target search's set-level scores on host threads while the second context searches the control set
        fs = ex.submit(scores, ctx_t, "t", res_t)
        dbs["c"] = fc.result()
        res_c = search(ctx_c, "c")
        score_rev = scores(ctx_c, "c", res_c)
        score = fs.result()
    else:                   # one after the other (the preparation above is what overlaps)
        score = scores(ctx_t, "t", res_t)
        dbs["c"] = fc.result()
        res_c = search(ctx_c, "c")
        score_rev = scores(ctx_c, "c", res_c)
    w1 = ex.submit(pm.write_db, P("cScore"), [(k, v) for k, v in sorted(score.items())], 5)
    pm.write_db(P("cScore_rev"), [(k, v) for k, v in sorted(score_rev.items())], 5)
    w1.result()
    sa.module_host("filtermatchbyfdr", [P("cScore"), P("cScore_rev"), P("match"), "--fdr", "0.05", "--report-fdr", "0", "--threads", str(threads), "--compressed", "0", "-v", "3"])
    match = pm.read_db(P("match"))
    wall = time.perf_counter() - t0
    mark("done")
    ex.shutdown()
    ctx_t.close()
    ctx_c.close()
    return score, score_rev, res_t[2], res_c[2], match, wall, marks


def run(unique_genomes=900, genome_nt=20000, replicas=10, n_arrays=5000, per_array=20, threads=16, keep_dir=None, seed=2024):
    import spacepharer_amd as sa
    t_all = time.perf_counter()
    genomes, arrays, truth = pm.make_data(unique_genomes, genome_nt, n_arrays, per_array, seed=seed)
    times = {"make_data_python_s": time.perf_counter() - t_all}
    tmp = None
    if keep_dir is None:
        tmp = tempfile.TemporaryDirectory()
        keep_dir = tmp.name
    P = lambda x: os.path.join(keep_dir, x)      # noqa: E731
    dbs = prepare(keep_dir, genomes, arrays, replicas, threads, times)
    ctx = sa.SearchContext()
    t0 = time.perf_counter()
    score, n_t = search_and_score(ctx, dbs, "t", times)
    t1 = time.perf_counter()
    score_rev, n_c = search_and_score(ctx, dbs, "c", times)
    t2 = time.perf_counter()
    pm.write_db(P("cScore"), [(k, v) for k, v in sorted(score.items())], 5)
    pm.write_db(P("cScore_rev"), [(k, v) for k, v in sorted(score_rev.items())], 5)
    sa.module_host("filtermatchbyfdr", [P("cScore"), P("cScore_rev"), P("match"), "--fdr", "0.05", "--report-fdr", "0", "--threads", str(threads), "--compressed", "0", "-v", "3"])
    match = pm.read_db(P("match"))
    t3 = time.perf_counter()
    times.update(search_target_s=t1 - t0, search_control_s=t2 - t1, fdr_s=t3 - t2)
    U = unique_genomes
    found = other = 0
    for a, entry in match.items():
        for line in entry.splitlines():
            g = int(line.split("\t")[0])
            if truth[a] >= 0 and g % U == truth[a]:
                found += 1
            else:
                other += 1
    chain_s = times["six_host_modules_s"] + times["load_dbs_numpy_s"] + (t3 - t0)
    n_sp = n_arrays * per_array
    info = {"config": "BASELINE configs[4] on one MI355X: %d spacers vs %d genomes (%d unique x %d replicas) + codon-reversed control + FDR" % (n_sp, U * replicas, U, replicas),
            "spacers": n_sp, "arrays": n_arrays, "query_peptides": dbs["q"]["aa"].n, "target_orfs": dbs["t"]["aa"].n, "target_residues": int(dbs["t"]["aa"].off[-1]),
            "target_nt": genome_nt * U * replicas, "alignments_target": n_t, "alignments_control": n_c,
            "seconds": {k: (round(v, 4) if not isinstance(v, dict) else v) for k, v in times.items()},
            "chain_s_without_python_input_writing": round(chain_s, 4), "spacers_per_s_whole_chain": n_sp / chain_s,
            "spacer_queries_per_s_two_searches_gpu_side": 2 * n_sp / sum(times[x][k] for x in ("t", "c") for k in ("prefilter_s", "align_s")),
            "matches_reported": found + other, "true_matches_found": found, "real_arrays": sum(1 for g in truth if g >= 0), "planted_pairs": sum(1 for g in truth if g >= 0) * replicas,
            "other_matches": other}
    if os.environ.get("SP_CHAIN_PIPELINED", "1") != "0":
        # the same chain, its independent steps overlapped: a second directory, compared entry by entry with the sequential run above
        with tempfile.TemporaryDirectory() as d2:
            laps = {}
            for lap, overlap in enumerate((True, False, True, False)):      # (two laps each: page cache and allocator warm from the second on, as for the sequential chain, whose modules ran after make_data warmed them)
                for f in os.listdir(d2):
                    os.remove(os.path.join(d2, f))
                p_score, p_rev, p_nt, p_nc, p_match, wall, marks = run_pipelined(d2, genomes, arrays, replicas, threads, n_arrays, hits_hint=max(n_t, n_c), overlap_scores=overlap)
                assert p_score == score and p_rev == score_rev and (p_nt, p_nc) == (n_t, n_c) and p_match == match, "the pipelined chain delivered other results"
                k = "preparation_and_scores_overlapped" if overlap else "preparation_overlapped"
                if k not in laps or wall < laps[k][0]:
                    laps[k] = (wall, marks)
            best = min(laps.values(), key=lambda x: x[0])
            info["pipelined"] = {"what": "the same module calls and files, independent steps overlapped on host threads: query / target / control preparation side by side and DBs read as they appear "
                                         "(preparation_overlapped); additionally the target search's set-level scores computed while a second search context searches the control set "
                                         "(preparation_and_scores_overlapped: 0.93 s on some boxes, slower than the sequential chain on others -- two host-side phases beside each other under the "
                                         "box's 16-core quota); results equal the sequential chain's in every lap",
                                 "chain_s": round(best[0], 4), "spacers_per_s_whole_chain": n_sp / best[0], "marks_s": best[1], "results_equal_sequential": True,
                                 "variants": {k: {"chain_s": round(v[0], 4), "spacers_per_s_whole_chain": n_sp / v[0], "marks_s": v[1]} for k, v in laps.items()}}
    state = {"dbs": dbs, "score": score, "score_rev": score_rev, "match": match, "truth": truth, "ctx": ctx, "dir": keep_dir, "tmp": tmp, "unique": U, "replicas": replicas}
    return info, state


if __name__ == "__main__":
    want_ref = "--reference" in sys.argv
    a = [int(x) for x in sys.argv[1:] if x != "--reference"]
    info, st = run(*a)
    st["ctx"].close()
    if want_ref:
        info["reference_on_this_box"] = reference_timing(st["dir"], st["dbs"])
        ref = info["reference_on_this_box"]
        if "prep_plus_two_searches_s_estimate" in ref:
            mine = info["seconds"]["six_host_modules_s"] + sum(info["seconds"][x][k] for x in ("t", "c") for k in ("stage_targets_s", "prefilter_s", "align_s"))
            info["x_reference_prep_plus_two_searches"] = round(ref["prep_plus_two_searches_s_estimate"] / mine, 1)
    print(json.dumps(info))
