#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference classes (oracle/_ref/ref_harness).

Runs only in the build container (needs /root/reference and `make -C oracle ref`).  The committed
outputs (tests/golden/*.json.gz) are pure data: seeded synthetic inputs + what the reference computed.

The harness mirrors the control loops around the reference classes (Prefiltering::runSplit, IndexBuilder::fillDatabase,
Alignment::run).  When a binary of the reference is at hand (SP_REF_BINARY, default /tmp/ref_build/src/spacepharer -- the
survey's cmake build; this script never builds one) every case is ALSO searched by the real `prefilter` and `align` modules
with the workflow's flag lists on DB triples, and their result DBs must equal the harness' text entry
for entry; `ref.cross_checked_with_reference_binary` records that it happened.

usage: python tests/golden/make_golden.py
"""
import gzip, json, os, random, struct, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SP_REFERENCE", "/root/reference")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
M80 = os.path.join(REF, "lib/mmseqs/data/VTML80.out")
M40 = os.path.join(REF, "lib/mmseqs/data/VTML40.out")

AA = "ACDEFGHIKLMNPQRSTVWY"
# VTML40 background (first 20 entries of the header line of the matrix file)
BG = [0.0819, 0.0169, 0.0453, 0.0616, 0.0465, 0.0745, 0.0217, 0.0549, 0.0582, 0.0905, 0.0274, 0.0435,
      0.0506, 0.0375, 0.0447, 0.0533, 0.0569, 0.081, 0.0132, 0.0329]


def rand_seq(rng, n):
    return "".join(rng.choices(AA, weights=BG, k=n))


def mutate(rng, s, nsub):
    s = list(s)
    for _ in range(nsub):
        p = rng.randrange(len(s))
        s[p] = rng.choice(AA)
    return "".join(s)


def make_case(seed, n_targets, n_queries, tmin, tmax, qmin, qmax, families=0):
    rng = random.Random(seed)
    T = []
    for i in range(n_targets):
        L = rng.randint(tmin, tmax)
        s = rand_seq(rng, L)
        r = rng.random()
        if r < 0.08 and L > 40:      # low-complexity / tandem repeat stretch -> tantan masks it
            unit = rand_seq(rng, rng.randint(1, 6))
            p = rng.randrange(0, L - 30)
            rep = (unit * 40)[: rng.randint(15, 30)]
            s = s[:p] + rep + s[p + len(rep):]
        elif r < 0.12:               # a few X / odd letters
            s = list(s)
            for _ in range(rng.randint(1, 4)):
                s[rng.randrange(L)] = rng.choice("XBZJUO*x")
            s = "".join(s)
        T.append(s)
    # families: near-identical targets so that one query collects many hits (exercises the top-N cut)
    for f in range(families):
        base = rand_seq(rng, rng.randint(60, 120))
        for _ in range(rng.randint(20, 60)):
            T[rng.randrange(n_targets)] = mutate(rng, base, rng.randint(0, 3)) + rand_seq(rng, rng.randint(0, 20))
    Q = []
    for i in range(n_queries):
        L = rng.randint(qmin, qmax)
        r = rng.random()
        if r < 0.6:                  # planted: window of a target with 0-2 substitutions
            for _ in range(100):
                t = T[rng.randrange(n_targets)]
                if len(t) >= L:
                    break
            p = rng.randrange(0, max(1, len(t) - L + 1))
            q = mutate(rng, t[p:p + L], rng.choice([0, 0, 1, 1, 2]))
            if rng.random() < 0.15 and L > 12:   # planted with a gap (exercises gapped SW / cigar)
                c = rng.randrange(3, L - 3)
                q = q[:c] + (rand_seq(rng, rng.randint(1, 2)) if rng.random() < 0.5 else "") + q[c + rng.randint(0, 2):]
        elif r < 0.9:
            q = rand_seq(rng, L)
        elif r < 0.95:
            q = list(rand_seq(rng, L)); q[rng.randrange(L)] = "X"; q = "".join(q)
        else:
            q = (rand_seq(rng, 2) * 20)[:L]      # biased composition
        Q.append(q)
    return T, Q


CASES = {
    # name: (seed, n_targets, n_queries, tmin, tmax, qmin, qmax, families, extra harness flags)
    "small":   (11, 300, 96, 20, 400, 6, 30, 0, ["--dump-kmers", "8", "--dump-index"]),
    "spacer":  (12345, 2000, 400, 50, 300, 9, 12, 0, ["--dump-kmers", "4"]),
    "family":  (77, 1500, 120, 40, 250, 10, 40, 6, ["--dump-kmers", "0", "--max-seqs", "20"]),
    "longq":   (5, 600, 40, 100, 1200, 40, 300, 2, ["--dump-kmers", "0"]),
    # the family case with CacheFriendlyOperations<16> forced (this host's L2 would pick <2> for any DB below 4 M sequences): the bin a
    # target falls into decides which of several equal-score hits survive the --max-seqs cut (QueryMatcher.cpp:383-411)
    "family16": (77, 1500, 120, 40, 250, 10, 40, 6, ["--dump-kmers", "0", "--max-seqs", "20", "--bin-size", "16"]),
    # targets beyond 32 768 residues: 16-bit diagonals wrap and UngappedAlignment scores such diagonals in pieces
    # (UngappedAlignment.cpp:216-233 computeLongScore); see make_long_target_case
    "longt":   (31, 40, 60, 50, 300, 10, 40, 0, ["--dump-kmers", "0"]),
    # other sensitivities than the workflow's 5.7 (PredictMatch.cpp:11): -s 7.5 = k-mer threshold 96, long similar-k-mer lists, and -s 1 = threshold 154, nearly
    # exact k-mers only; --max-seqs 50 cuts inside the lists of the family members (round 6)
    "sens75":  (4242, 1200, 300, 40, 300, 9, 14, 2, ["--dump-kmers", "2", "-s", "7.5", "--max-seqs", "50"]),
    "sens1":   (4243, 1200, 300, 40, 300, 9, 30, 2, ["--dump-kmers", "2", "-s", "1", "--max-seqs", "300"]),
}


def make_long_target_case(seed, n_targets, n_queries, tmin, tmax, qmin, qmax):
    """a few ordinary targets plus one of 40 000 and one of 33 500 residues; most queries are windows of the long ones, spread over
    their whole length (so that target positions below and above 32 768 / 65 536 - query length occur), with 0-2 substitutions"""
    rng = random.Random(seed)
    T = [rand_seq(rng, rng.randint(tmin, tmax)) for _ in range(n_targets)]
    T[3] = rand_seq(rng, 40000)
    T[17] = rand_seq(rng, 33500)
    Q = []
    for i in range(n_queries):
        L = rng.randint(qmin, qmax)
        if i % 6 == 5:
            Q.append(rand_seq(rng, L))
            continue
        t = T[3] if i % 2 == 0 else T[17]
        edge = [0, 5, 32760, 32768 - L // 2, 32768, 32770, len(t) - L]
        p = edge[(i // 2) % len(edge)] if i < 28 else rng.randrange(0, len(t) - L + 1)
        p = max(0, min(len(t) - L, p))
        Q.append(mutate(rng, t[p:p + L], rng.choice([0, 1, 1, 2])))
    return T, Q


REF_BINARY = os.environ.get("SP_REF_BINARY", "/tmp/ref_build/src/spacepharer")
PREF_FLAGS = ("--sub-mat nucl:nucleotide.out,aa:VTML40.out --seed-sub-mat nucl:nucleotide.out,aa:VTML80.out -k 6 --k-score 2147483647 --alph-size nucl:5,aa:21 "
              "--max-seq-len 65535 --split 0 --split-mode 2 --split-memory-limit 0 -c 0 --cov-mode 0 --comp-bias-corr 1 --diag-score 1 --exact-kmer-matching 0 "
              "--mask 1 --mask-lower-case 0 --min-ungapped-score 15 --add-self-matches 0 --spaced-kmer-mode 1 --db-load-mode 0 --pca 1 --pcb 1.5 "
              "--spaced-kmer-pattern 11011101 --threads 1 --compressed 0 -v 1 -s 5.7").split()
ALN_FLAGS = ("--sub-mat nucl:nucleotide.out,aa:VTML40.out -a 1 --alignment-mode 2 --alignment-output-mode 0 --wrapped-scoring 0 -e 200 --min-seq-id 0 --min-aln-len 0 "
             "--seq-id-mode 0 --alt-ali 0 -c 0 --cov-mode 0 --max-seq-len 65535 --comp-bias-corr 1 --max-rejected 2147483647 --max-accept 2147483647 "
             "--add-self-matches 0 --db-load-mode 0 --pca 1 --pcb 1.5 --score-bias 0 --realign 0 --realign-score-bias -0.2 --realign-max-seqs 2147483647 "
             "--gap-open nucl:10,aa:16 --gap-extend 2 --zdrop 40 --threads 1 --compressed 0 -v 1").split()


def write_db(path, seqs):
    off, idx = 0, []
    with open(path, "wb") as f:
        for k, s in enumerate(seqs):
            b = s.encode() + b"\n\0"
            f.write(b)
            idx.append("%d\t%d\t%d\n" % (k, off, len(b)))
            off += len(b)
    open(path + ".index", "w").write("".join(idx))
    open(path + ".dbtype", "wb").write(struct.pack("<i", 0))


def read_db(path):
    files = [path] if os.path.exists(path) else []
    while not os.path.exists(path) and os.path.exists("%s.%d" % (path, len(files))):      # split data files <path>.0, <path>.1, ...
        files.append("%s.%d" % (path, len(files)))
    data = b"".join(open(p, "rb").read() for p in files)
    out = {}
    for line in open(path + ".index"):
        k, o, l = (int(x) for x in line.split("\t"))
        out[k] = data[o:o + l - 1].decode()
    return out


def cross_check(d, T, Q, ref, flags):
    """the real modules on DB triples against the harness' text; False when there is no binary or the case forces what no flag can (--bin-size)"""
    if not os.path.exists(REF_BINARY) or "--bin-size" in flags:
        return False
    max_seqs = flags[flags.index("--max-seqs") + 1] if "--max-seqs" in flags else "300"
    pref_flags = list(PREF_FLAGS)
    if "-s" in flags:
        pref_flags[pref_flags.index("-s") + 1] = flags[flags.index("-s") + 1]
    q, t, pref, aln = (os.path.join(d, x) for x in ("qdb", "tdb", "pref", "aln"))
    write_db(q, Q)
    write_db(t, T)
    subprocess.check_call([REF_BINARY, "prefilter", q, t, pref, "--max-seqs", max_seqs] + pref_flags, stdout=subprocess.DEVNULL)
    subprocess.check_call([REF_BINARY, "align", q, t, pref, aln] + ALN_FLAGS, stdout=subprocess.DEVNULL)
    got_pref, got_aln = read_db(pref), read_db(aln)
    exp_pref = {x["id"]: x["pref"] for x in ref["queries"]}
    exp_aln = {x["id"]: x["text"] for x in ref["aln"]}
    for k in range(len(Q)):
        assert got_pref.get(k, "") == exp_pref.get(k, ""), ("prefilter", k, got_pref.get(k, "")[:200], exp_pref.get(k, "")[:200])
        assert got_aln.get(k, "") == exp_aln.get(k, ""), ("align", k, got_aln.get(k, "")[:200], exp_aln.get(k, "")[:200])
    return True


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build the harness first: make -C oracle ref")
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, (seed, nt, nq, tmin, tmax, qmin, qmax, fam, flags) in CASES.items():
        if only and name not in only:
            continue
        T, Q = make_long_target_case(seed, nt, nq, tmin, tmax, qmin, qmax) if name == "longt" else make_case(seed, nt, nq, tmin, tmax, qmin, qmax, fam)
        with tempfile.TemporaryDirectory() as d:
            tf, qf, of = (os.path.join(d, x) for x in ("t.txt", "q.txt", "o.json"))
            open(tf, "w").write("\n".join(T) + "\n")
            open(qf, "w").write("\n".join(Q) + "\n")
            subprocess.check_call([HARNESS, M80, M40, tf, qf, of] + flags)
            ref = json.load(open(of))
            ref["cross_checked_with_reference_binary"] = cross_check(d, T, Q, ref, flags)
            print(name, "real prefilter + align modules equal the harness:", ref["cross_checked_with_reference_binary"])
        doc = {"case": name, "seed": seed, "targets": T, "queries": Q, "flags": flags, "ref": ref}
        out = os.path.join(HERE, name + ".json.gz")
        # a fixture that WAS checked against the real modules is not silently replaced by one that was not (a container without the survey's build of the
        # reference binary): tests/test_oracle_golden.py::test_committed_fixtures_were_cross_checked would fail on the result anyway
        if os.path.exists(out) and not ref["cross_checked_with_reference_binary"] and "--bin-size" not in flags and "--allow-unchecked" not in sys.argv:
            was = json.load(gzip.open(out))["ref"].get("cross_checked_with_reference_binary")
            if was:
                sys.exit("%s: the committed fixture was cross-checked against the reference binary, this run cannot (no %s); not overwriting (--allow-unchecked to force)" % (name, REF_BINARY))
        with gzip.GzipFile(out, "wb", mtime=0) as f:
            f.write(json.dumps(doc, separators=(",", ":")).encode())
        print(name, "->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
