"""predictmatch end to end on synthetic data (BASELINE.json configs[4] in miniature, one GPU): CRISPR spacer arrays against phage genomes
AND against the codon-reversed control set, the set-level chain, FDR filtering -- the path data/predictmatch.sh:27-125 walks with
~25 module calls and ~20 DBs.  Here: the preparation steps (extractorfs / translatenucs / reverseseqbycodon) run as the host module
drop-ins on DB triples, the two searches on the GPU, proteinaln2nucl ... combinescore in memory on the delivered alignment records
(sp_ctx_set_scores), filtermatchbyfdr as a host module on the two tiny score DBs.

Synthetic genomes are coding-dense (genes = ATG + sense codons + stop on either strand, short random spacers between them); half
of the arrays are "real": their spacers are 30-nt windows of ONE genome with 0-2 substitutions, the other half is random.
usage: python tools/predictmatch_e2e.py [genomes] [genome_nt] [arrays] [spacers_per_array]   -> one JSON line"""
import json
import os
import random
import struct
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STOPS = ("TAA", "TAG", "TGA")
SENSE = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT" if a + b + c not in STOPS]
COMP = str.maketrans("ACGT", "TGCA")


def make_genome(rng, n_nt):
    out, n = [], 0
    while n < n_nt:
        gene = "ATG" + "".join(rng.choice(SENSE) for _ in range(rng.randint(60, 350))) + rng.choice(STOPS)
        if rng.random() < 0.5:
            gene = gene.translate(COMP)[::-1]
        gap = "".join(rng.choice("ACGT") for _ in range(rng.randint(5, 60)))
        out += [gene, gap]
        n += len(gene) + len(gap)
    return "".join(out)[:n_nt]


def make_data(n_genomes, genome_nt, n_arrays, per_array, seed=2024):
    rng = random.Random(seed)
    genomes = [make_genome(rng, genome_nt) for _ in range(n_genomes)]
    arrays, truth = [], []
    for a in range(n_arrays):
        real = a % 2 == 0
        g = rng.randrange(n_genomes) if real else -1
        sp = []
        for _ in range(per_array):
            if real and rng.random() < 0.8:
                p = rng.randrange(0, genome_nt - 30)
                s = list(genomes[g][p:p + 30])
                for _ in range(rng.choice((0, 0, 1, 1, 2))):
                    s[rng.randrange(30)] = rng.choice("ACGT")
                s = "".join(s)
                if rng.random() < 0.5:
                    s = s.translate(COMP)[::-1]
            else:
                s = "".join(rng.choice("ACGT") for _ in range(30))
            sp.append(s)
        arrays.append(sp)
        truth.append(g)
    return genomes, arrays, truth


def write_db(path, entries, dbtype):
    off, idx = 0, []
    with open(path, "wb") as f:
        for k, payload in entries:
            b = payload.encode() + b"\0"
            f.write(b)
            idx.append((k, off, len(b)))
            off += len(b)
    with open(path + ".index", "w") as f:
        for k, o, l in sorted(idx):
            f.write("%d\t%d\t%d\n" % (k, o, l))
    open(path + ".dbtype", "wb").write(struct.pack("<i", dbtype))


def read_db(path):
    data = open(path, "rb").read()
    out = {}
    for line in open(path + ".index"):
        k, o, l = (int(x) for x in line.split("\t"))
        out[k] = data[o:o + l - 1].decode()
    return out


def nucl_db(path, sets):
    """sets: list of lists of nucleotide sequences; contig keys 0.., set = position in `sets`"""
    seqs, lookup, k = [], [], 0
    for s, members in enumerate(sets):
        for m in members:
            seqs.append((k, m + "\n"))
            lookup.append("%d\tc%d\t%d\n" % (k, k, s))
            k += 1
    write_db(path, seqs, 1)
    open(path + ".lookup", "w").write("".join(lookup))


def prepare(d, genomes, arrays, times=None):
    """createsetdb's preparation through the host module drop-ins; returns peptides, nucleotide ORFs and set keys of queries, targets, control"""
    import spacepharer_amd as sa
    P = lambda x: os.path.join(d, x)      # noqa: E731
    th = ["--threads", "16", "--compressed", "0", "-v", "3"]
    t0 = time.perf_counter()
    nucl_db(P("q_nucl"), arrays)
    nucl_db(P("t_nucl"), [[g] for g in genomes])
    t1 = time.perf_counter()
    sa.module_host("extractorfs", [P("q_nucl"), P("q_nucl_orf"), "--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"] + th)
    sa.module_host("translatenucs", [P("q_nucl_orf"), P("q")] + th)
    sa.module_host("extractorfs", [P("t_nucl"), P("t_nucl_orf"), "--min-length", "30", "--orf-start-mode", "0", "--create-lookup", "1"] + th)
    sa.module_host("translatenucs", [P("t_nucl_orf"), P("t")] + th)
    sa.module_host("reverseseqbycodon", [P("t_nucl_orf"), P("c_nucl_orf")] + th)
    sa.module_host("translatenucs", [P("c_nucl_orf"), P("c")] + th)
    t2 = time.perf_counter()
    if times is not None:
        times["write_input_dbs_python_s"], times["six_host_modules_s"] = t1 - t0, t2 - t1
    out = {}
    for name in ("q", "t", "c"):
        aa, nuc = read_db(P(name)), read_db(P(name + "_nucl_orf"))
        assert sorted(aa) == list(range(len(aa))) == sorted(nuc)
        look = P(("t" if name == "c" else name) + "_nucl_orf.lookup")
        sets = [int(l.rstrip("\n").split("\t")[2]) for l in open(look)]
        out[name] = {"aa": [aa[k].rstrip("\n") for k in range(len(aa))], "nucl": [nuc[k].rstrip("\n") for k in range(len(nuc))], "set": sets}
    return out


def buffers(dbs):
    """the DBs as the flat buffers the C ABI takes (what a host program that read the DB files holds); not part of the timed chain"""
    import ctypes as C
    from spacepharer_amd._lib import _concat
    return {k: {"aa": _concat(v["aa"]), "nucl": _concat(v["nucl"]), "set": (C.c_uint32 * len(v["set"]))(*v["set"]), "n": len(v["aa"])} for k, v in dbs.items()}


def search_and_score(ctx, bufs, tname, stage=0, times=None):
    """one search of the queries against target set `tname` + the in-memory set-level chain -> {query set: cScore entry}"""
    q, t = bufs["q"], bufs[tname]
    t0 = time.perf_counter()
    ctx.set_targets_raw(t["aa"][0], t["aa"][1], t["n"])
    t1 = time.perf_counter()
    ctx.set_queries_raw(q["aa"][0], q["aa"][1], q["n"])
    ctx.prefilter()
    aln, cig, n = ctx.align_hits_raw()
    t2 = time.perf_counter()
    out = ctx.set_scores_raw(q["nucl"], t["nucl"], q["set"], t["set"], aln, cig, stage=stage)
    t3 = time.perf_counter()
    if times is not None:
        times[tname + "_stage_targets_s"], times[tname + "_search_s"], times[tname + "_set_scores_s"] = t1 - t0, t2 - t1, t3 - t2
    return out, n


def run(n_genomes=1000, genome_nt=10000, n_arrays=500, per_array=20):
    import spacepharer_amd as sa
    genomes, arrays, truth = make_data(n_genomes, genome_nt, n_arrays, per_array)
    t, detail = {}, {}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        dbs = prepare(d, genomes, arrays, detail)
        t["host_modules_s"] = detail.pop("six_host_modules_s")      # without the Python that writes the inputs / reads the outputs back
        ctx = sa.SearchContext()
        bufs = buffers(dbs)
        t0 = time.perf_counter()
        score, n_t = search_and_score(ctx, bufs, "t", times=detail)
        t["search_target_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        score_rev, n_c = search_and_score(ctx, bufs, "c", times=detail)
        t["search_control_s"] = time.perf_counter() - t0
        ctx.close()
        t0 = time.perf_counter()
        P = lambda x: os.path.join(d, x)      # noqa: E731
        write_db(P("cScore"), [(k, v) for k, v in sorted(score.items())], 5)
        write_db(P("cScore_rev"), [(k, v) for k, v in sorted(score_rev.items())], 5)
        sa.module_host("filtermatchbyfdr", [P("cScore"), P("cScore_rev"), P("match"), "--fdr", "0.05", "--report-fdr", "0", "--threads", "16", "--compressed", "0", "-v", "3"])
        match = read_db(P("match"))
        t["fdr_s"] = time.perf_counter() - t0
    found = wrong = 0
    for a, entry in match.items():
        for line in entry.splitlines():
            g = int(line.split("\t")[0])
            if g == truth[a]:
                found += 1
            else:
                wrong += 1
    real = sum(1 for g in truth if g >= 0)
    total = sum(t.values())
    return {"genomes": n_genomes, "genome_nt": genome_nt, "arrays": n_arrays, "spacers": n_arrays * per_array, "query_peptides": len(dbs["q"]["aa"]), "target_orfs": len(dbs["t"]["aa"]),
            "target_residues": sum(len(x) for x in dbs["t"]["aa"]), "alignments_target": n_t, "alignments_control": n_c, "seconds": {k: round(v, 4) for k, v in t.items()}, "detail": {k: round(v, 4) for k, v in detail.items()},
            "spacers_per_s_whole_chain": n_arrays * per_array / total, "matches_reported": found + wrong, "true_matches_found": found, "real_arrays": real, "other_matches": wrong}, dbs, score, score_rev, match, truth


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)[0]))
