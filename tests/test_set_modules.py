"""The set-level module drop-ins (host-only, so they run without a GPU) on BASELINE.json
configs[0] -- the reference's bundled examples -- against the DBs the reference code produced for the same inputs
(tests/golden/examples.json.gz, made by tests/golden/make_examples.py through oracle/_ref/ref_sets: the reference's own module
translation units, and repeated through a binary of the reference when one is at hand).  Every output entry must be byte-identical.

Host-only code, but part of the path the GPU box has to deliver: every test here is collected twice, once for the CPU suite and once with the
`gpu` marker so that the driver's `-m gpu` run on the MI355X box executes it as well."""
import os
import struct

import pytest

from conftest import load_golden


@pytest.fixture(autouse=True, params=[pytest.param("cpu-suite"), pytest.param("gpu-suite", marks=pytest.mark.gpu)])
def suite(request):
    return request.param


ANCILLARY = ("_h", "_h.index", "_h.dbtype", ".lookup", ".source", "_mapping", "_names.dmp", "_nodes.dmp", "_merged.dmp")


def links(prefix):
    """suffix -> 'link' / 'file' of the sequence-DB side files next to <prefix>"""
    return {s: ("link" if os.path.islink(prefix + s) else "file") for s in ANCILLARY if os.path.lexists(prefix + s)}


def write_db(path, db):
    off, idx = 0, []
    with open(path, "wb") as f:
        for k, v in db["entries"].items():
            b = v.encode() + b"\0"
            f.write(b)
            idx.append((int(k), off, len(b)))
            off += len(b)
    with open(path + ".index", "w") as f:
        for k, o, l in sorted(idx):
            f.write("%d\t%d\t%d\n" % (k, o, l))
    open(path + ".dbtype", "wb").write(struct.pack("<i", db["dbtype"]))


def read_db(path):
    data = open(path, "rb").read()
    out = {}
    for line in open(path + ".index"):
        k, o, l = (int(x) for x in line.split("\t"))
        assert data[o + l - 1] == 0
        out[str(k)] = data[o:o + l - 1].decode()
    return out, struct.unpack("<i", open(path + ".dbtype", "rb").read(4))[0]


@pytest.fixture(scope="module")
def ex(tmp_path_factory):
    g = load_golden("examples")
    d = tmp_path_factory.mktemp("examples")
    for name, db in g["db"].items():
        write_db(str(d / name), db)
    for name, text in list(g["lookup"].items()) + list(g["lookup_x"].items()):
        open(str(d / (name + ".lookup")), "w").write(text)
    for name in ("q_nucl", "t_nucl"):      # createdb's .source: file number, file name
        files = g["query_files"] if name == "q_nucl" else g["target_files"]
        open(str(d / (name + ".source")), "w").write("".join("%d\t%s\n" % (i, f) for i, f in enumerate(files)))
    for n, text in g["taxonomy"].items():      # a target set WITH a taxonomy: tx = t_nucl + the three NCBI dump files
        open(str(d / ("tx_%s.dmp" % n)), "w").write(text)
    write_db(str(d / "tx"), g["db"]["t_nucl"])
    return g, d


def check(g, d, produced, golden_name):
    got, dbtype = read_db(str(d / produced))
    exp = g["db"][golden_name]
    assert dbtype == exp["dbtype"], (produced, dbtype)
    assert sorted(got, key=int) == sorted(exp["entries"], key=int), produced
    for k, v in exp["entries"].items():
        assert got[k] == v, (produced, k, got[k][:300], v[:300])


THREADS = ["--threads", "4", "--compressed", "0", "-v", "3"]


def test_extractorfs_translatenucs_reverseseqbycodon(ex):
    """createsetdb's target- and query-side preparation (data/createsetdb.sh:87-109)"""
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    sa.module_host("extractorfs", [P("q_nucl"), P("o_q_nucl_orf"), "--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"] + THREADS)
    check(g, d, "o_q_nucl_orf", "q_nucl_orf")
    check(g, d, "o_q_nucl_orf_h", "q_nucl_orf_h")
    assert open(P("o_q_nucl_orf.lookup")).read() == g["lookup"]["q_nucl_orf"]
    want = lambda name: {k: v for k, v in g["ancillary"][name].items() if k in ANCILLARY}      # noqa: E731
    assert links(P("o_q_nucl_orf")) == want("q_nucl_orf")      # .source linked (extractorfs.cpp:156), lookup and headers written
    sa.module_host("extractorfs", [P("t_nucl"), P("o_t_nucl_orf"), "--min-length", "30", "--orf-start-mode", "0", "--create-lookup", "1"] + THREADS)
    check(g, d, "o_t_nucl_orf", "t_nucl_orf")
    check(g, d, "o_t_nucl_orf_h", "t_nucl_orf_h")
    assert open(P("o_t_nucl_orf.lookup")).read() == g["lookup"]["t_nucl_orf"]
    sa.module_host("translatenucs", [P("o_q_nucl_orf"), P("o_q"), "--translation-table", "1", "--add-orf-stop", "0"] + THREADS)
    check(g, d, "o_q", "q")
    # DBReader::softlinkDb(in, out, SEQUENCE_ANCILLARY): headers, lookup and source of the ORF DB are linked under the peptide DB's name
    assert links(P("o_q")) == want("q") and open(P("o_q_h.index")).read() == open(P("o_q_nucl_orf_h.index")).read()
    assert os.path.realpath(P("o_q.lookup")) == os.path.realpath(P("o_q_nucl_orf.lookup"))
    sa.module_host("translatenucs", [P("o_t_nucl_orf"), P("o_t"), "--translation-table", "1", "--add-orf-stop", "0"] + THREADS)
    check(g, d, "o_t", "t")
    sa.module_host("reverseseqbycodon", [P("o_t_nucl_orf"), P("o_c_nucl_orf")] + THREADS)
    check(g, d, "o_c_nucl_orf", "c_nucl_orf")
    assert links(P("o_c_nucl_orf")) == want("c_nucl_orf")
    sa.module_host("translatenucs", [P("o_c_nucl_orf"), P("o_c"), "--translation-table", "1", "--add-orf-stop", "0"] + THREADS)
    check(g, d, "o_c", "c")
    assert links(P("o_c")) == want("c") and links(P("o_t")) == want("t")
    # a second run replaces the links instead of failing on them
    sa.module_host("reverseseqbycodon", [P("o_t_nucl_orf"), P("o_c_nucl_orf")] + THREADS)
    assert links(P("o_c_nucl_orf")) == want("c_nucl_orf")


def test_orf_and_translation_corners(ex):
    """crafted entries: IUPAC codes, N runs, lower case, U / u, lengths not divisible by three, sequences shorter than a codon; all three
    --orf-start-mode values"""
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    sa.module_host("translatenucs", [P("x_nucl"), P("o_x_aa")] + THREADS)
    check(g, d, "o_x_aa", "x_aa")
    for name, min_len, mode in (("x_orf0", 1, 0), ("x_orf1", 2, 1), ("x_orf2", 1, 2)):
        sa.module_host("extractorfs", [P("x_nucl"), P("o_" + name), "--min-length", str(min_len), "--orf-start-mode", str(mode), "--create-lookup", "1"] + THREADS)
        check(g, d, "o_" + name, name)
        check(g, d, "o_" + name + "_h", name + "_h")
        assert open(P("o_" + name + ".lookup")).read() == g["lookup_x"][name]
    sa.module_host("translatenucs", [P("o_x_orf1"), P("o_x_orf1_aa")] + THREADS)
    check(g, d, "o_x_orf1_aa", "x_orf1_aa")
    sa.module_host("reverseseqbycodon", [P("x_nucl"), P("o_x_rev")] + THREADS)
    check(g, d, "o_x_rev", "x_rev")


@pytest.mark.parametrize("tag,T", [("", "t"), ("_rev", "c")])
def test_proteinaln2nucl_to_combinescore(ex, tag, T):
    """predictmatch's chain after the search (data/predictmatch.sh:34-75), target set and reversed control set"""
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    sa.module_host("proteinaln2nucl", [P("q_nucl_orf"), P(T + "_nucl_orf"), P("q"), P(T), P("prot_result" + tag), P("o_nucl_result" + tag),
                                       "--sub-mat", "nucl:nucleotide.out,aa:VTML40.out", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"] + THREADS)
    check(g, d, "o_nucl_result" + tag, "nucl_result" + tag)
    sa.module_host("combineprotnuclaln", [P("prot_result" + tag), P("o_nucl_result" + tag), P("o_result" + tag)] + THREADS)
    check(g, d, "o_result" + tag, "result" + tag)
    sa.module_host("besthitperset", [P("q"), P(T), P("o_result" + tag), P("o_aggregate" + tag), "--simple-best-hit", "1"] + THREADS)
    check(g, d, "o_aggregate" + tag, "aggregate" + tag)
    sa.module_host("mergeresultsbyset", [P("q_set_to_member"), P("o_aggregate" + tag), P("o_aggregate_merged" + tag)] + THREADS)
    check(g, d, "o_aggregate_merged" + tag, "aggregate_merged" + tag)
    sa.module_host("combinescore", [P("q"), P(T), P("o_aggregate_merged" + tag), P("o_cScore" + tag), str(d)] + THREADS)
    check(g, d, "o_cScore" + tag, "cScore" + tag)


def test_proteinaln2nucl_with_its_own_default_gap_costs(ex):
    """a bare call: nucleotide gap open 5 / extend 2 (Parameters.cpp defaults), another row of the tabulated Gumbel parameters"""
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    sa.module_host("proteinaln2nucl", [P("q_nucl_orf"), P("t_nucl_orf"), P("q"), P("t"), P("prot_result"), P("o_nucl_result_default")] + THREADS)
    check(g, d, "o_nucl_result_default", "nucl_result_default")
    assert g["db"]["nucl_result_default"]["entries"] != g["db"]["nucl_result"]["entries"]


def test_besthitperset_second_best_correction(ex):
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    sa.module_host("besthitperset", [P("q"), P("t"), P("result"), P("o_aggregate_full"), "--simple-best-hit", "0"] + THREADS)
    check(g, d, "o_aggregate_full", "aggregate_full")


def test_filtermatchbyfdr_and_summarizeresults(ex):
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    for name, fdr, rep in (("match", "0.05", "0"), ("match_fdr", "0.05", "1"), ("match_all", "1", "0")):
        sa.module_host("filtermatchbyfdr", [P("cScore"), P("cScore_rev"), P("o_" + name), "--fdr", fdr, "--report-fdr", rep] + THREADS)
        check(g, d, "o_" + name, name)
    variants = [("summary%d" % fmt, "t_nucl", "match", "aln", ["--fmt", str(fmt)]) for fmt in (0, 1, 2)]
    variants.append(("summary_fdr", "t_nucl", "match_fdr", "aln", ["--fmt", "1", "--report-fdr", "1"]))
    # a target set with a taxonomy (<targetDB>_nodes.dmp exists, SummarizeResults.cpp:17-21): taxon id, rank and name columns, a merged taxon id,
    # a taxon that is not in the table, taxon 0; --lca-ranks and both --tax-lineage forms
    variants += [("summary_tax", "tx", "match", "aln_tax", ["--fmt", "1"]), ("summary_tax_ranks", "tx", "match", "aln_tax", ["--fmt", "1", "--lca-ranks", "species,genus,family,superkingdom"]),
                 ("summary_tax_lineage1", "tx", "match", "aln_tax", ["--fmt", "1", "--tax-lineage", "1"]),
                 ("summary_tax_lineage2", "tx", "match", "aln_tax", ["--fmt", "1", "--tax-lineage", "2", "--lca-ranks", "order"])]
    for name, tdb, match, aln, flags in variants:
        sa.module_host("summarizeresults", [P(tdb), P(match), P(aln), P("o_" + name)] + flags + ["--db-output", "1"] + THREADS)
        check(g, d, "o_" + name, name)
        # the workflow's form (and the only one the reference binary offers): one flat report file, entries in match-DB order, no index
        sa.module_host("summarizeresults", [P(tdb), P(match), P(aln), P("o_" + name + ".tsv")] + flags + THREADS)
        assert open(P("o_" + name + ".tsv")).read() == g["report"][name], name
        assert not os.path.exists(P("o_" + name + ".tsv.index"))
    assert "\t40521\tspecies\tListeria phage A118\n" in g["report"]["summary_tax"] and "\t10710\tspecies\t" in g["report"]["summary_tax"]      # direct and merged id
    with pytest.raises(sa.SpError):      # a match DB without the FDR column cannot be reported with it
        sa.module_host("summarizeresults", [P("t_nucl"), P("match"), P("aln"), P("o_bad"), "--report-fdr", "1"] + THREADS)
    with pytest.raises(sa.SpError):
        sa.module_host("summarizeresults", [P("tx"), P("match"), P("aln_tax"), P("o_bad"), "--lca-ranks", "species,nonsense"] + THREADS)


def test_unknown_flags_and_unsupported_values_fail_loudly(ex):
    import spacepharer_amd as sa
    g, d = ex
    P = lambda x: str(d / x)      # noqa: E731
    with pytest.raises(sa.SpError):
        sa.module_host("translatenucs", [P("q_nucl_orf"), P("o_x"), "--no-such-flag", "1"])
    with pytest.raises(sa.SpError):
        sa.module_host("translatenucs", [P("q_nucl_orf"), P("o_x"), "--translation-table", "11"])
    with pytest.raises(sa.SpError):
        sa.module_host("filtermatchbyfdr", [P("cScore"), P("missing"), P("o_x")])
    assert not os.path.exists(P("o_x.dbtype"))


def test_filtermatchbyfdr_on_long_score_lists(tmp_path):
    """seeded synthetic score DBs that reach the hull / interpolation branches of filtermatchbyfdr (FilterMatchbyFdr.cpp:109-200),
    with and without --report-fdr, with an empty control DB, with no true matches"""
    import spacepharer_amd as sa
    g = load_golden("examples")
    nontrivial = 0
    assert len(g["fdr_cases"]) >= 17
    for i, c in enumerate(g["fdr_cases"]):
        P = lambda x: str(tmp_path / ("%d_%s" % (i, x)))      # noqa: E731
        write_db(P("pos"), {"dbtype": 5, "entries": c["pos"]})
        write_db(P("neg"), {"dbtype": 5, "entries": c["neg"]})
        sa.module_host("filtermatchbyfdr", [P("pos"), P("neg"), P("out"), "--fdr", repr(c["fdr"]), "--report-fdr", str(c["report"])] + THREADS)
        got, dbtype = read_db(P("out"))
        assert dbtype == 12 and got == c["out"], i
        nontrivial += float.fromhex(c["threshold"]) > 0
    assert nontrivial >= 8


def test_threaded_drivers_write_the_bytes_a_single_thread_writes(tmp_path):
    """the host modules cut their input into chunks of about equal bytes and format them on --threads workers: data file, index, lookup and
    headers must not depend on the thread count (a few large genomes + thousands of small entries, so that chunks split both ways)"""
    import filecmp
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import predictmatch_e2e as pm
    import spacepharer_amd as sa
    genomes, arrays, _ = pm.make_data(40, 30000, 300, 12, seed=5)
    outs = {}
    for th in ("1", "7"):
        d = tmp_path / ("t" + th)
        d.mkdir()
        P = lambda x: str(d / x)      # noqa: E731
        common = ["--threads", th, "--compressed", "0", "-v", "3"]
        pm.nucl_db(P("q_nucl"), arrays)
        pm.nucl_db(P("t_nucl"), [[g] for g in genomes])
        sa.module_host("extractorfs", [P("q_nucl"), P("q_orf"), "--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"] + common)
        sa.module_host("extractorfs", [P("t_nucl"), P("t_orf"), "--min-length", "30", "--orf-start-mode", "0", "--create-lookup", "1"] + common)
        sa.module_host("translatenucs", [P("t_orf"), P("t_aa")] + common)
        sa.module_host("reverseseqbycodon", [P("t_orf"), P("c_orf")] + common)
        sa.module_host("translatenucs", [P("q_orf"), P("q_aa")] + common)
        outs[th] = d
    names = ["q_orf", "q_orf.index", "q_orf.lookup", "q_orf_h", "q_orf_h.index", "t_orf", "t_orf.index", "t_orf.lookup", "t_orf_h", "t_aa", "t_aa.index", "c_orf", "c_orf.index", "q_aa", "q_aa.index"]
    for n in names:
        assert filecmp.cmp(str(outs["1"] / n), str(outs["7"] / n), shallow=False), n
    assert os.path.getsize(str(outs["1"] / "t_orf")) > 500000 and sum(1 for _ in open(str(outs["1"] / "q_orf.index"))) > 10000


def test_a_failing_chunk_fails_the_module_and_leaves_no_dbtype(tmp_path):
    import spacepharer_amd as sa
    P = lambda x: str(tmp_path / x)      # noqa: E731
    seq = {"entries": {str(k): "ATGAAATTTGGGCCC\n" for k in range(3000)}, "dbtype": 1}
    aa = {"entries": {str(k): "MKFGP\n" for k in range(3000)}, "dbtype": 0}
    write_db(P("qn"), seq); write_db(P("tn"), seq); write_db(P("qa"), aa); write_db(P("ta"), aa)
    aln = {"entries": {str(k): "%d\t20\t1.00\t1.0E-03\t0\t4\t5\t0\t4\t5\t5M\n" % (k if k != 2500 else 999999) for k in range(3000)}, "dbtype": 5}
    write_db(P("aln"), aln)
    with pytest.raises(sa.SpError) as e:
        sa.module_host("proteinaln2nucl", [P("qn"), P("tn"), P("qa"), P("ta"), P("aln"), P("out"), "--gap-open", "nucl:10,aa:16", "--gap-extend", "2", "--threads", "4", "--compressed", "0", "-v", "3"])
    assert "999999" in str(e.value) and not os.path.exists(P("out.dbtype"))
