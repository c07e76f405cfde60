"""The CPU oracle of the set-level steps (oracle/sp_sets_oracle.py) against the DBs the reference's
own code produced for the bundled examples and for the synthetic score lists (tests/golden/examples.json.gz): byte-identical
entries.  This is what pins the oracle; the product is tested against the same fixtures in tests/test_set_modules.py."""
import os
import sys

import pytest

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import sp_sets_oracle as so      # noqa: E402


# (host-only checks of the checker itself: CPU suite only -- the -m gpu run counts tests that touch the product)


@pytest.fixture(scope="module")
def g():
    return load_golden("examples")


def db(g, name):
    return {int(k): v for k, v in g["db"][name]["entries"].items()}


def first_col_map(g, name):
    return {k: int(v.split()[0]) for k, v in db(g, name).items()}


def test_orfs_translation_reversal(g):
    for nucl, lookup, orf, min_len, mode in (("q_nucl", "q_nucl", "q_nucl_orf", 9, 1), ("t_nucl", "t_nucl", "t_nucl_orf", 30, 0)):
        seqs, heads, lk = so.extract_orfs(db(g, nucl), g["lookup"][lookup], min_len, mode)
        assert seqs == db(g, orf) and heads == db(g, orf + "_h") and lk == g["lookup"][orf]
    for orf, aa in (("q_nucl_orf", "q"), ("t_nucl_orf", "t"), ("c_nucl_orf", "c")):
        assert {k: so.translate_entry(v) for k, v in db(g, orf).items()} == db(g, aa)
    assert {k: so.reverse_by_codon(v) for k, v in db(g, "t_nucl_orf").items()} == db(g, "c_nucl_orf")


def test_orf_and_translation_corners(g):
    """IUPAC codes, N runs, lower case, U / u, lengths that are not multiples of three (the partial codon + newline IS translated:
    '\\n' is the ncbi4na value of Y), sequences shorter than a codon -- crafted entries taken through the reference's Orf and
    TranslateNucl classes"""
    x = db(g, "x_nucl")
    assert {k: so.translate_entry(v) for k, v in x.items() if so.translate_entry(v) is not None} == db(g, "x_aa")
    assert db(g, "x_aa")[4] == "MKY\n" and db(g, "x_aa")[6] == "I\n" and db(g, "x_aa")[2] == "MB*P\n"
    for name, min_len, mode in (("x_orf0", 1, 0), ("x_orf1", 2, 1), ("x_orf2", 1, 2)):
        seqs, heads, lk = so.extract_orfs(x, g["lookup_x"]["x_nucl"], min_len, mode)
        assert seqs == db(g, name) and heads == db(g, name + "_h") and lk == g["lookup_x"][name], name
    assert {k: so.translate_entry(v) for k, v in db(g, "x_orf1").items()} == db(g, "x_orf1_aa")
    assert {k: so.reverse_by_codon(v) for k, v in x.items()} == db(g, "x_rev")


def test_nucleotide_statistics(g):
    info = g["info"]["nucl_evalue"]
    ev = so.Evaluer([float.fromhex(x) for x in info["alp"]], info["dbResidues"])
    for s, L, e, b in info["probes"]:
        assert ev.evalue(s, L) == float.fromhex(e) and int(ev.bit_score(s)) == b, (s, L)


@pytest.mark.parametrize("tag,T", [("", "t"), ("_rev", "c")])
def test_proteinaln2nucl_to_combinescore(g, tag, T):
    info = g["info"]["nucl_evalue" + tag]
    ev = so.Evaluer([float.fromhex(x) for x in info["alp"]], info["dbResidues"])
    q_nuc, t_nuc, q_aa, t_aa = db(g, "q_nucl_orf"), db(g, T + "_nucl_orf"), db(g, "q"), db(g, T)
    t_star = {k: v.startswith("*") for k, v in t_aa.items()}
    prot = db(g, "prot_result" + tag)
    nucl = {k: so.protein_aln_to_nucl(v, q_nuc[k], t_nuc, q_aa[k].startswith("*"), t_star, ev) for k, v in prot.items()}
    assert nucl == db(g, "nucl_result" + tag)
    res = {k: so.combine_prot_nucl(prot[k], nucl[k]) for k in prot}
    assert res == db(g, "result" + tag)
    m2s, size = first_col_map(g, T + "_member_to_set"), first_col_map(g, T + "_set_size")
    agg = {k: so.best_hit_per_set(v, m2s, size, True) for k, v in res.items()}
    assert agg == db(g, "aggregate" + tag)
    merged = {k: so.merge_by_set(v, agg) for k, v in db(g, "q_set_to_member").items()}
    assert merged == db(g, "aggregate_merged" + tag)
    qsize = first_col_map(g, "q_set_size")
    assert {k: so.combine_score(v, qsize[k], m2s) for k, v in merged.items()} == db(g, "cScore" + tag)
    if tag == "":
        assert {k: so.best_hit_per_set(v, m2s, size, False) for k, v in res.items()} == db(g, "aggregate_full")


def test_fdr_and_summary(g):
    for name, fdr, rep in (("match", 0.05, 0), ("match_fdr", 0.05, 1), ("match_all", 1.0, 0)):
        out, thr = so.fdr_filter(db(g, "cScore"), db(g, "cScore_rev"), fdr, rep)
        assert out == db(g, name) and thr == float.fromhex(g["info"][name]["threshold"])
    nontrivial = 0
    for c in g["fdr_cases"]:
        out, thr = so.fdr_filter({int(k): v for k, v in c["pos"].items()}, {int(k): v for k, v in c["neg"].items()}, c["fdr"], c["report"])
        assert thr == float.fromhex(c["threshold"]) and out == {int(k): v for k, v in c["out"].items()}
        nontrivial += thr > 0
    assert nontrivial >= 8 and len(g["fdr_cases"]) >= 17
    match, aln = db(g, "match"), db(g, "aln")
    for fmt in (0, 1, 2):
        assert {k: so.summarize(v, aln[k], fmt) for k, v in match.items()} == db(g, "summary%d" % fmt)
    assert {k: so.summarize(v, aln[k], 1, report_fdr=True) for k, v in db(g, "match_fdr").items()} == db(g, "summary_fdr")


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_reference_harnesses_are_built_from_the_real_parameter_classes():
    """Round 6: oracle/Makefile compiles the reference's Parameters.cpp / LocalParameters.cpp (headers generated by the reference's own cmake/xxdi.pl from its data files) and
    links both drivers with every symbol resolved -- no `#define Parameters ...` stand-in, no --unresolved-symbols=ignore-all."""
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    rules = "\n".join(ln for ln in mk.splitlines() if not ln.lstrip().startswith("#"))
    assert "unresolved-symbols" not in rules and "ref_par" not in rules
    assert "Parameters.cpp" in rules and "LocalParameters.cpp" in rules and "xxdi.pl" in rules
    assert not os.path.exists(os.path.join(ROOT, "oracle", "ref_par.h"))
    src = open(os.path.join(ROOT, "oracle", "ref_sets.cpp")).read()
    assert "#define Parameters" not in src and "LocalParameters::getLocalInstance()" in src
