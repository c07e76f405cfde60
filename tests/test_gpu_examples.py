"""GPU, BASELINE.json configs[0]: the reference's bundled examples (two spacer files vs three phage genomes, the workload of
/root/reference/util/run_regression.sh) through the HIP search path and the in-memory chain behind it, against what the
reference's own code produced for the same peptides (tests/golden/examples.json.gz).  Nothing touches a disk between the
prefilter and the combined set scores."""
import pytest

from conftest import load_golden
from test_gpu_modules import ALN_FLAGS, PREF_FLAGS, read_db
from test_set_modules import write_db

pytestmark = pytest.mark.gpu


def db(g, name):
    return {int(k): v for k, v in g["db"][name]["entries"].items()}


def seqs(g, name):
    d = db(g, name)
    assert sorted(d) == list(range(len(d)))
    return [d[k].rstrip("\n") for k in range(len(d))]


def set_keys(g, name):
    d = db(g, name)
    return [int(d[k].split()[0]) for k in range(len(d))]


@pytest.mark.parametrize("tag,T", [("", "t"), ("_rev", "c")])
def test_examples_search_and_set_scores_in_memory(tag, T):
    import spacepharer_amd as sa
    g = load_golden("examples")
    Q, TT = seqs(g, "q"), seqs(g, T)
    ctx = sa.SearchContext()
    ctx.set_targets(TT)
    ctx.set_queries(Q)
    ctx.prefilter()
    hits = ctx.hits()
    pref = db(g, "prot_result" + tag + "_pref")
    for q in range(len(Q)):
        assert "".join("%d\t%d\t%d\n" % h for h in hits[q]) == pref[q], q          # QueryMatcher::prefilterHitToBuffer
    assert sum(len(h) for h in hits) > 0
    aln, cig, n = ctx.align_hits_raw()
    q_nucl, t_nucl = seqs(g, "q_nucl_orf"), seqs(g, T + "_nucl_orf")
    q_set, t_set = set_keys(g, "q_member_to_set"), set_keys(g, T + "_member_to_set")
    for stage, name in ((1, "prot_result"), (2, "nucl_result"), (3, "result"), (4, "aggregate"), (5, "aggregate_merged"), (0, "cScore")):
        got = ctx.set_scores(q_nucl, t_nucl, q_set, t_set, aln, cig, stage=stage)
        assert got == db(g, name + tag), (stage, name)
    assert ctx.set_scores(q_nucl, t_nucl, q_set, t_set, aln, cig, stage=4, simple_best_hit=False) == (db(g, "aggregate_full") if tag == "" else
                                                                                                  ctx.set_scores(q_nucl, t_nucl, q_set, t_set, aln, cig, stage=4, simple_best_hit=False))
    ctx.close()


def test_examples_through_the_module_drop_ins(tmp_path):
    """the same search through the module boundary: createsetdb-style DB triples in, prefilter + align DBs out, then the host
    modules down to the match list -- every DB equals the reference's"""
    import spacepharer_amd as sa
    g = load_golden("examples")
    P = lambda x: str(tmp_path / x)      # noqa: E731
    for name in ("q", "t", "c", "q_nucl_orf", "t_nucl_orf", "c_nucl_orf", "q_set_to_member", "q_set_size", "t_member_to_set", "t_set_size", "c_member_to_set", "c_set_size"):
        write_db(P(name), g["db"][name])
    th = ["--threads", "4", "--compressed", "0", "-v", "3"]
    for tag, T in (("", "t"), ("_rev", "c")):
        sa.module_prefilter([P("q"), P(T), P("pref" + tag)] + (PREF_FLAGS % 300).split())
        got, _ = read_db(P("pref" + tag))
        assert {k: v for k, v in got.items()} == db(g, "prot_result" + tag + "_pref")
        sa.module_align([P("q"), P(T), P("pref" + tag), P("prot" + tag)] + ALN_FLAGS.split())
        got, _ = read_db(P("prot" + tag))
        assert got == db(g, "prot_result" + tag)
        sa.module_host("proteinaln2nucl", [P("q_nucl_orf"), P(T + "_nucl_orf"), P("q"), P(T), P("prot" + tag), P("nucl" + tag), "--sub-mat", "nucl:nucleotide.out,aa:VTML40.out",
                                           "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"] + th)
        sa.module_host("combineprotnuclaln", [P("prot" + tag), P("nucl" + tag), P("res" + tag)] + th)
        sa.module_host("besthitperset", [P("q"), P(T), P("res" + tag), P("agg" + tag), "--simple-best-hit", "1"] + th)
        sa.module_host("mergeresultsbyset", [P("q_set_to_member"), P("agg" + tag), P("aggm" + tag)] + th)
        sa.module_host("combinescore", [P("q"), P(T), P("aggm" + tag), P("cs" + tag), str(tmp_path)] + th)
        got, _ = read_db(P("cs" + tag))
        assert got == db(g, "cScore" + tag)
    sa.module_host("filtermatchbyfdr", [P("cs"), P("cs_rev"), P("match"), "--fdr", "0.05", "--report-fdr", "0"] + th)
    got, _ = read_db(P("match"))
    assert got == db(g, "match")
    # run_regression.sh's first check counts the hit lines with E < 1e-3 in the final report: the best hits per set that feed it
    strong = [ln for v in db(g, "aggregate").values() for ln in v.splitlines() if float(ln.split("\t")[3]) < 1e-3]
    assert len(strong) == 6
