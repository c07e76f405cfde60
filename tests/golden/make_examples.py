#!/usr/bin/env python3
"""Generate tests/golden/examples.json.gz: BASELINE.json configs[0] -- the reference's bundled examples
(/root/reference/examples: two spacer files vs three phage genomes, the workload of util/run_regression.sh) taken through
the set-level steps around the search hot path by the REAL reference code: oracle/_ref/ref_sets runs the reference's own
module translation units (extractorfs.cpp ... SummarizeResults.cpp, compiled in place and unmodified, with the reference's own
Parameters.cpp / LocalParameters.cpp parsing the flags below -- round 6), oracle/_ref/ref_harness the search classes.

Runs only in the build container.  The committed output is pure data: the example sequences (data files of the reference's
own regression test) and the content of every DB a step produced, keyed by DB key.

Pipeline (reference workflow lines in parentheses):
  nucleotide DBs (createdb is out of scope: sequences in file order, keys 0..n-1, set = source file)
  queries : extractorfs --min-length 9 --orf-start-mode 1  -> translatenucs          (data/createsetdb.sh:87-91,105-109)
  targets : extractorfs --min-length 30 --orf-start-mode 0 -> translatenucs          (data/createsetdb.sh:92-95; CreateSetDb.cpp:9-11)
  control : the targets' ORFs through reverseseqbycodon -> translatenucs             (data/createsetdb.sh:97-104)
  search  : prefilter + align with the workflow's flags      (ref_harness: QueryMatcher, Matcher)
  proteinaln2nucl -> combineprotnuclaln -> besthitperset -> mergeresultsbyset -> combinescore   (data/predictmatch.sh:27-75)
  the same against the control set, then filtermatchbyfdr --fdr 0.05                (data/predictmatch.sh:77-125)
  summarizeresults --fmt 1 on an alignment DB in convertalis' 13-column layout      (data/predictmatch.sh:166)

When a binary of the reference is at hand (SP_REF_BINARY, default /tmp/ref_build/src/spacepharer -- the survey's cmake build;
this script never builds one) every module call below is repeated through it with the workflow's flag syntax and the two
outputs must be equal; `info.cross_checked_with_reference_binary` says whether that happened.

usage: python tests/golden/make_examples.py
"""
import glob
import gzip
import json
import os
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SP_REFERENCE", "/root/reference")
SETS = os.path.join(ROOT, "oracle", "_ref", "ref_sets")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
DATA = os.path.join(REF, "lib/mmseqs/data")


def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    out, name, seq = [], None, []
    with op(path, "rt") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(seq)))
                name, seq = line[1:], []
            elif line:
                seq.append(line.strip())
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def write_db(path, entries, dbtype):
    """entries: list of (key, bytes payload without the terminating NUL) in data order"""
    off, idx = 0, []
    with open(path, "wb") as f:
        for k, payload in entries:
            f.write(payload + b"\0")
            idx.append((k, off, len(payload) + 1))
            off += len(payload) + 1
    with open(path + ".index", "w") as f:
        for k, o, l in sorted(idx):
            f.write("%d\t%d\t%d\n" % (k, o, l))
    open(path + ".dbtype", "wb").write(struct.pack("<i", dbtype))


def read_db(path):
    files = [path] if os.path.exists(path) else sorted(glob.glob(path + ".[0-9]*"), key=lambda p: int(p.rsplit(".", 1)[1]))
    data = b"".join(open(p, "rb").read() for p in files)
    out = {}
    for line in open(path + ".index"):
        k, o, l = (int(x) for x in line.split("\t"))
        out[k] = data[o:o + l - 1].decode()
    return out, struct.unpack("<i", open(path + ".dbtype", "rb").read(4))[0]


def nucl_db(path, files):
    """what createdb leaves for a list of FASTA files: sequence DB (dbtype 1), header DB, lookup (key, accession, file number)"""
    seqs, heads, lookup, source = [], [], [], []
    k = 0
    for fi, fn in enumerate(files):
        source.append("%d\t%s" % (fi, os.path.basename(fn)))
        for name, s in read_fasta(fn):
            seqs.append((k, s.encode() + b"\n"))
            heads.append((k, name.encode() + b"\n"))
            lookup.append("%d\t%s\t%d" % (k, name.split()[0], fi))
            k += 1
    write_db(path, seqs, 1)
    write_db(path + "_h", heads, 12)
    open(path + ".lookup", "w").write("\n".join(lookup) + "\n")
    open(path + ".source", "w").write("\n".join(source) + "\n")
    return [s.decode().strip() for _, s in seqs], [int(l.split("\t")[2]) for l in lookup]


def set_dbs(prefix, contig_set):
    """<prefix>_member_to_set, _set_to_member, _set_size from <prefix>_nucl_orf.lookup (orf key, contig key, set) -- what
    orftocontig / filterdb / swapdb / result2stats leave (data/createsetdb.sh:111-139); inputs of the aggregation steps"""
    member_set = {}
    for line in open(prefix + "_nucl_orf.lookup"):
        orf, contig, _ = line.rstrip("\n").split("\t")
        member_set[int(orf)] = contig_set[int(contig)]
    write_db(prefix + "_member_to_set", [(m, b"%d\n" % s) for m, s in sorted(member_set.items())], 5)
    sets = sorted(set(member_set.values()))
    write_db(prefix + "_set_to_member", [(s, b"".join(b"%d\n" % m for m in sorted(member_set) if member_set[m] == s)) for s in sets], 5)
    write_db(prefix + "_set_size", [(s, b"%d\n" % sum(1 for m in member_set if member_set[m] == s)) for s in sets], 12)
    return member_set


def run(*args, **kw):
    return subprocess.run([str(a) for a in args], check=True, **kw)


def search(d, q, t, out, max_seqs=300):
    """prefilter + align of peptide DB q against peptide DB t through the reference classes; writes the pref / aln text DBs"""
    qdb, _ = read_db(q)
    tdb, _ = read_db(t)
    assert sorted(qdb) == list(range(len(qdb))) and sorted(tdb) == list(range(len(tdb)))
    tf, qf, of = (os.path.join(d, x) for x in ("t.txt", "q.txt", "o.json"))
    open(tf, "w").write("".join(tdb[k] for k in range(len(tdb))))
    open(qf, "w").write("".join(qdb[k] for k in range(len(qdb))))
    run(HARNESS, os.path.join(DATA, "VTML80.out"), os.path.join(DATA, "VTML40.out"), tf, qf, of, "--dump-kmers", "0", "--max-seqs", max_seqs)
    ref = json.load(open(of))
    pref = {x["id"]: x["pref"] for x in ref["queries"]}
    aln = {x["id"]: x["text"] for x in ref["aln"]}
    write_db(out + "_pref", [(k, pref.get(k, "").encode()) for k in range(len(qdb))], 7)
    write_db(out, [(k, aln.get(k, "").encode()) for k in range(len(qdb))], 5)
    return {"aaDbSize": ref["aaDbSize"], "alpGapped": ref["alpGapped"], "binSize": ref["binSize"]}


REF_BINARY = os.environ.get("SP_REF_BINARY", "/tmp/ref_build/src/spacepharer")
XCHECK = {"n": 0}


def module(name, dbs, flags=(), outputs=1, binary_flags=None, binary_extra=(), **kw):
    """one module call through oracle/_ref/ref_sets (the reference's translation unit); when the reference binary exists the same call
    goes through it as well, into <out>.bin, and the DB contents must be equal.  The last `outputs` DBs are outputs."""
    # (round 6: ref_sets parses its command line with the reference's own Parameters / LocalParameters -- the flag spellings are the binary's)
    p = run(SETS, name, *dbs, *(flags if binary_flags is None else binary_flags), "--threads", "1", "-v", "1", **kw)
    if os.path.exists(REF_BINARY) and outputs:
        alt = list(dbs[:-outputs]) + [x + ".bin" for x in dbs[-outputs:]]
        run(REF_BINARY, name, *alt, *binary_extra, *(flags if binary_flags is None else binary_flags), "--threads", "1", "-v", "1", stdout=subprocess.DEVNULL)
        for a, b in zip(dbs[-outputs:], alt[-outputs:]):
            if os.path.exists(a + ".index") or os.path.exists(b + ".index"):
                assert read_db(a) == read_db(b), (name, a)
            else:
                assert open(a).read() == open(b).read(), (name, a)
        XCHECK["n"] += 1
    return p


def extractorfs(src, out, min_len, start_mode):
    # createsetdb.sh:87-95: the workflow's flag list (the other ORF options stay at their defaults)
    module("extractorfs", [src, out], ["--min-length", min_len, "--orf-start-mode", start_mode, "--create-lookup", 1])


def ancillary(prefix):
    """the files a module left next to <prefix> (suffix -> 'link' / 'file'): translatenucs and reverseseqbycodon end with
    DBReader::softlinkDb(in, out, SEQUENCE_ANCILLARY)"""
    out = {}
    for f in sorted(glob.glob(prefix + "*")):
        suf = f[len(prefix):]
        if suf and suf[0] in "._" and not suf.endswith(".bin") and ".bin" not in suf:
            out[suf] = "link" if os.path.islink(f) else "file"
    return out


def min_score(db):
    """filtermatchbyfdr keeps the scores >= threshold and the threshold is one of the target scores: the smallest score written"""
    v = [float(ln.split("\t")[1]) for e in db.values() for ln in e.splitlines()]
    return min(v).hex() if v else None


def main():
    if not (os.path.exists(SETS) and os.path.exists(HARNESS)):
        sys.exit("build the harnesses first: make -C oracle ref")
    ex = os.path.join(REF, "examples")
    qfiles = sorted(glob.glob(os.path.join(ex, "*.fas")))
    tfiles = sorted(glob.glob(os.path.join(ex, "GCA*.fna.gz")))
    doc = {"case": "examples", "query_files": [os.path.basename(f) for f in qfiles], "target_files": [os.path.basename(f) for f in tfiles], "db": {}, "info": {}}
    with tempfile.TemporaryDirectory() as d:
        P = lambda x: os.path.join(d, x)      # noqa: E731
        _, qset = nucl_db(P("q_nucl"), qfiles)
        _, tset = nucl_db(P("t_nucl"), tfiles)
        extractorfs(P("q_nucl"), P("q_nucl_orf"), 9, 1)
        module("translatenucs", [P("q_nucl_orf"), P("q")])
        extractorfs(P("t_nucl"), P("t_nucl_orf"), 30, 0)
        module("translatenucs", [P("t_nucl_orf"), P("t")])
        module("reverseseqbycodon", [P("t_nucl_orf"), P("c_nucl_orf")])
        module("translatenucs", [P("c_nucl_orf"), P("c")])
        doc["ancillary"] = {n: ancillary(P(n)) for n in ("q_nucl_orf", "q", "t", "c_nucl_orf", "c")}
        set_dbs(P("q"), qset)
        set_dbs(P("t"), tset)
        set_dbs(P("c"), tset)
        nuc = os.path.join(DATA, "nucleotide.out")
        for tag, T in (("", "t"), ("_rev", "c")):
            doc["info"]["search" + tag] = search(d, P("q"), P(T), P("prot_result" + tag))
            # predictmatch.sh:40-44 passes the search's parameter list: nucleotide matrix of the data directory, gap open 10, extend 2
            module("proteinaln2nucl", [P("q_nucl_orf"), P(T + "_nucl_orf"), P("q"), P(T), P("prot_result" + tag), P("nucl_result" + tag)],
                   ["--sub-mat", nuc, "--gap-open", 10, "--gap-extend", 2], binary_flags=["--sub-mat", "aa:VTML40.out,nucl:" + nuc, "--gap-open", "aa:16,nucl:10", "--gap-extend", "aa:2,nucl:2"])
            p = run(SETS, "nuclalp", P(T + "_nucl_orf"), nuc, 10, 2, capture_output=True, text=True)
            doc["info"]["nucl_evalue" + tag] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
            module("combineprotnuclaln", [P("prot_result" + tag), P("nucl_result" + tag), P("result" + tag)])
            module("besthitperset", [P("q"), P(T), P("result" + tag), P("aggregate" + tag)], ["--simple-best-hit", 1])
            module("mergeresultsbyset", [P("q_set_to_member"), P("aggregate" + tag), P("aggregate_merged" + tag)])
            module("combinescore", [P("q"), P(T), P("aggregate_merged" + tag), P("cScore" + tag)], binary_extra=[d])      # the command table also lists a tmpDir
        # the module's own gap costs (nucleotide gap open 5, extend 2: Parameters.cpp defaults), i.e. a bare proteinaln2nucl call
        module("proteinaln2nucl", [P("q_nucl_orf"), P("t_nucl_orf"), P("q"), P("t"), P("prot_result"), P("nucl_result_default")], ["--sub-mat", nuc],
               binary_flags=["--sub-mat", "aa:VTML40.out,nucl:" + nuc])
        for fdr, rep, name in ((0.05, 0, "match"), (0.05, 1, "match_fdr"), (1.0, 0, "match_all")):
            module("filtermatchbyfdr", [P("cScore"), P("cScore_rev"), P(name)], ["--fdr", fdr, "--report-fdr", rep])
            doc["info"][name] = {"threshold": min_score(read_db(P(name))[0])}
        # the non-simple branch of besthitperset (second-best correction), not used by predictmatch but part of the module
        module("besthitperset", [P("q"), P("t"), P("result"), P("aggregate_full")], ["--simple-best-hit", 0])
        # summarizeresults: an alignment DB in the layout convertalis writes for
        # --format-output tsetid,query,qset,target,evalue,qstart,qend,qlen,tstart,tend,qaln,taln,empty|taxid (data/predictmatch.sh:140-150),
        # one line per best hit of a spacer, merged per query set; the numbers come from the aggregate DB
        agg, _ = read_db(P("aggregate"))
        qlook = [l.split("\t") for l in open(P("q_nucl_orf.lookup")).read().splitlines()]
        qnames = [l.split("\t")[1] for l in open(P("q_nucl.lookup")).read().splitlines()]
        tnames = [l.split("\t")[1] for l in open(P("t_nucl.lookup")).read().splitlines()]
        tmember = {int(l.split("\t")[0]): int(l.split("\t")[1]) for l in open(P("t_nucl_orf.lookup")).read().splitlines()}
        # a hand-made taxonomy next to a copy of the target set (the reference's check is "<targetDB>_nodes.dmp exists",
        # SummarizeResults.cpp:17-21): a few nodes, one merged id, one target set whose taxon is unknown, one without a taxon
        nodes = [(1, 1, "no rank", "root"), (10239, 1, "superkingdom", "Viruses"), (28883, 10239, "order", "Caudovirales"), (10699, 28883, "family", "Siphoviridae"),
                 (40521, 10699, "species", "Listeria phage A118"), (10744, 28883, "family", "Podoviridae"), (244310, 10744, "no rank", "unclassified Podoviridae"),
                 (186765, 10699, "genus", "Lambdalikevirus"), (10710, 186765, "species", "Enterobacteria phage lambda")]
        open(P("tx_nodes.dmp"), "w").write("".join("%d\t|\t%d\t|\t%s\t|\t\t|\n" % (t, par, rank) for t, par, rank, _ in nodes))
        open(P("tx_names.dmp"), "w").write("".join("%d\t|\t%s\t|\t\t|\tscientific name\t|\n%d\t|\talias of %s\t|\t\t|\tsynonym\t|\n" % (t, nm, t, nm) for t, _, _, nm in nodes))
        open(P("tx_merged.dmp"), "w").write("99999\t|\t10710\t|\n")
        for ext in ("", ".index", ".dbtype"):
            os.symlink(P("t_nucl") + ext, P("tx") + ext)
        taxon_of_set = {0: 40521, 1: 99999, 2: 244310}
        lines, lines_tax = {}, {}
        for orf, contig, _ in qlook:
            for ln in agg.get(int(orf), "").splitlines():
                f = ln.split("\t")
                if len(f) < 10:
                    continue
                tcontig = tmember[int(f[0])]
                s = qset[int(contig)]
                cols = [str(tset[tcontig]), qnames[int(contig)], doc["query_files"][s], tnames[tcontig], f[3], f[4], f[5], f[6], f[7], f[8], "ACGT" * 2, "ACGA" * 2]
                lines.setdefault(s, []).append("\t".join(cols + ["-"]))
                lines_tax.setdefault(s, []).append("\t".join(cols + [str(taxon_of_set[tset[tcontig]] if s == 0 else (5 if tset[tcontig] == 0 else 0))]))
        write_db(P("aln"), [(s, ("\n".join(v) + "\n").encode()) for s, v in sorted(lines.items())], 5)
        write_db(P("aln_tax"), [(s, ("\n".join(v) + "\n").encode()) for s, v in sorted(lines_tax.items())], 5)
        # --db-output is not in the module's own flag list (LocalParameters.h:101-107: the binary always writes the flat report), but
        # par.dbOut is what the body tests: both forms are recorded, the flat one is what the reference binary is compared on
        doc["report"] = {}

        def summarize(tdb, match, aln, name, flags):
            module("summarizeresults", [tdb, match, aln, P(name)], flags + ["--db-output", 1], outputs=0)
            module("summarizeresults", [tdb, match, aln, P(name + ".tsv")], flags)
            doc["report"][name] = open(P(name + ".tsv")).read()

        for fmt in (0, 1, 2):
            summarize(P("t_nucl"), P("match"), P("aln"), "summary%d" % fmt, ["--fmt", fmt])
        summarize(P("t_nucl"), P("match_fdr"), P("aln"), "summary_fdr", ["--fmt", 1, "--report-fdr", 1])
        for name, flags in (("summary_tax", []), ("summary_tax_ranks", ["--lca-ranks", "species,genus,family,superkingdom"]), ("summary_tax_lineage1", ["--tax-lineage", 1]),
                            ("summary_tax_lineage2", ["--tax-lineage", 2, "--lca-ranks", "order"])):
            summarize(P("tx"), P("match"), P("aln_tax"), name, ["--fmt", 1] + flags)
        doc["taxonomy"] = {n: open(P("tx_" + n + ".dmp")).read() for n in ("nodes", "names", "merged")}
        # corner cases of the ORF finder and the translation that real genomes rarely show: IUPAC codes, N runs, lower case, U / u,
        # lengths that are not multiples of three, sequences shorter than a codon
        rngc = __import__("random").Random(99)
        corner = ["ATGAAATAG", "atgAAATAG", "ATGRAYTARCCC", "ATGNNNAAATGA", "ATGAAATA", "ATGAAAT", "AT", "ATG", "TTATTTCATGGG", "AUGAAAUAG", "augaaauag",
                  "ATGAAANNNNNNNNNNNNAAATAGATGCCCTAA", "CCCATGAAATTTGGGTAGCCCATGCCCAAATGACC", "ATGXXXAAA---CCCTAA", "TAATAGTGA", "ATGATGATGTAA"]
        for _ in range(12):
            corner.append("".join(rngc.choice("ACGTACGTACGTNRYacgtU") for _ in range(rngc.randint(3, 90))))
        write_db(P("x_nucl"), [(k, sq.encode() + b"\n") for k, sq in enumerate(corner)], 1)
        write_db(P("x_nucl_h"), [(k, b"c%d corner case\n" % k) for k in range(len(corner))], 12)
        open(P("x_nucl.lookup"), "w").write("".join("%d\tc%d\t%d\n" % (k, k, k % 3) for k in range(len(corner))))
        module("translatenucs", [P("x_nucl"), P("x_aa")])
        extractorfs(P("x_nucl"), P("x_orf0"), 1, 0)
        extractorfs(P("x_nucl"), P("x_orf1"), 2, 1)
        extractorfs(P("x_nucl"), P("x_orf2"), 1, 2)
        module("translatenucs", [P("x_orf1"), P("x_orf1_aa")])
        module("reverseseqbycodon", [P("x_nucl"), P("x_rev")])
        doc["lookup_x"] = {n: open(P(n + ".lookup")).read() for n in ("x_nucl", "x_orf0", "x_orf1", "x_orf2")}
        # filtermatchbyfdr beyond the examples (whose score lists are too short to leave its trivial branch): seeded synthetic
        # score DBs -- many null-like scores in both lists, a tail of true matches in the target list only; then lists of a few
        # distinct values (ties everywhere), tiny lists, a control list that dominates, cutoffs at both ends.
        # (Not pinned, because the reference reads outside its vectors there: ONE distinct target score with a non-empty control list
        # -- `slopeList.end()[-2]` of a one-segment hull, FilterMatchbyFdr.cpp:144 -- and --report-fdr 1 with an empty control list and
        # target scores <= DBL_MIN, :203-211 with :249.)
        import random
        doc["fdr_cases"] = []
        cases = [(1, 400, 400, 60, 0.05, 0, 0), (2, 400, 400, 60, 0.05, 1, 0), (3, 1500, 900, 150, 0.2, 1, 0), (4, 80, 300, 10, 0.05, 0, 0),
                 (5, 600, 40, 30, 0.1, 1, 0), (6, 300, 300, 0, 0.05, 0, 0), (7, 200, 0, 20, 0.05, 1, 1), (8, 500, 500, 100, 0.01, 0, 0),
                 (9, 400, 400, 40, 0.05, 1, 2), (10, 3, 2, 1, 0.05, 1, 0), (11, 6, 500, 2, 0.3, 1, 0), (12, 300, 300, 30, 1.0, 1, 0), (13, 300, 300, 30, 0.0, 1, 0),
                 (14, 2000, 2000, 5, 0.5, 1, 2), (15, 50, 1, 10, 0.05, 1, 0), (16, 700, 650, 80, 0.05, 1, 3), (17, 120, 0, 0, 0.05, 0, 1)]
        for ci, (seed, n_pos, n_neg, n_true, fdr, rep, kind) in enumerate(cases):
            rng = random.Random(seed)
            def draw():
                if kind == 1:      # strictly positive (an empty control list with --report-fdr 1 looks every score up in a list of the positive ones)
                    return 0.5 + rng.expovariate(0.35)
                if kind == 2:      # few distinct values: ties within and between the lists
                    return float(rng.randint(0, 9)) * 0.75
                if kind == 3:      # three-digit mantissas that collide after %.3E
                    return round(rng.expovariate(0.35), 1)
                return rng.expovariate(0.35) if rng.random() < 0.7 else 0.0
            def score_db(n, n_true):
                vals = [draw() for _ in range(n)] + [8.0 + rng.expovariate(0.08) for _ in range(n_true)]
                rng.shuffle(vals)
                ent, k = {}, 0
                while vals:
                    m = rng.randint(1, 6)
                    ent[k] = "".join("%d\t%.3E\t%.3E\n" % (j, v, rng.random()) for j, v in enumerate(vals[:m]))
                    vals = vals[m:]
                    k += 1
                return ent
            pos, neg = score_db(n_pos, n_true), (score_db(n_neg, 0) if n_neg else {})
            write_db(P("fdr_pos"), [(k, v.encode()) for k, v in pos.items()], 5)
            write_db(P("fdr_neg"), [(k, v.encode()) for k, v in neg.items()], 5)
            module("filtermatchbyfdr", [P("fdr_pos"), P("fdr_neg"), P("fdr_out")], ["--fdr", fdr, "--report-fdr", rep])
            out, _ = read_db(P("fdr_out"))
            doc["fdr_cases"].append({"fdr": fdr, "report": rep, "pos": {str(k): v for k, v in pos.items()}, "neg": {str(k): v for k, v in neg.items()},
                                     "out": {str(k): v for k, v in out.items()}, "threshold": min_score(out)})
        names = ["q_nucl", "t_nucl", "q_nucl_orf", "q_nucl_orf_h", "t_nucl_orf", "t_nucl_orf_h", "c_nucl_orf", "q", "t", "c", "aln", "aln_tax", "summary0", "summary1", "summary2",
                 "summary_fdr", "summary_tax", "summary_tax_ranks", "summary_tax_lineage1", "summary_tax_lineage2",
                 "match", "match_fdr", "match_all", "aggregate_full", "nucl_result_default", "x_nucl", "x_aa", "x_orf0", "x_orf0_h", "x_orf1", "x_orf1_h", "x_orf2", "x_orf2_h", "x_orf1_aa", "x_rev"]
        for base in ("q", "t", "c"):
            names += [base + "_member_to_set", base + "_set_to_member", base + "_set_size"]
        for tag in ("", "_rev"):
            names += [x + tag for x in ("prot_result", "nucl_result", "result", "aggregate", "aggregate_merged", "cScore")] + ["prot_result" + tag + "_pref"]
        for n in names:
            content, dbtype = read_db(P(n))
            doc["db"][n] = {"dbtype": dbtype, "entries": {str(k): v for k, v in sorted(content.items())}}
        doc["lookup"] = {n: open(P(n + ".lookup")).read() for n in ("q_nucl", "t_nucl", "q_nucl_orf", "t_nucl_orf")}
    doc["info"]["cross_checked_with_reference_binary"] = XCHECK["n"]
    out = os.path.join(HERE, "examples.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(doc, separators=(",", ":")).encode())
    print("examples ->", out, os.path.getsize(out), "bytes;", {n: len(v["entries"]) for n, v in doc["db"].items()})
    print("module calls repeated through the reference binary and equal:", XCHECK["n"])


if __name__ == "__main__":
    main()
