"""predictmatch end to end (BASELINE.json configs[4] in miniature; data/predictmatch.sh:27-125): synthetic genomes and spacer arrays
taken through tools/predictmatch_e2e.py -- host preparation modules, two GPU searches (targets, codon-reversed control), the
set-level chain in memory, FDR filter -- and checked against the oracles: the preparation outputs entry for entry, the combined
score entries of a sample of arrays recomputed from nothing but the peptides (C oracle: prefilter + align; Python oracle:
proteinaln2nucl ... combinescore), the match list from the score DBs, and what was planted is what is found."""
import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chain():
    import predictmatch_e2e as pm
    info, dbs, score, score_rev, match, truth = pm.run(300, 6000, 120, 12)
    return {"info": info, "dbs": dbs, "score": score, "score_rev": score_rev, "match": match, "truth": truth, "pm": pm}


def test_preparation_equals_the_oracle(chain, tmp_path):
    import sp_sets_oracle as so
    pm = chain["pm"]
    genomes, arrays, _ = pm.make_data(300, 6000, 120, 12)
    pm.nucl_db(str(tmp_path / "q_nucl"), arrays)
    orfs, _, look = so.extract_orfs(pm.read_db(str(tmp_path / "q_nucl")), open(str(tmp_path / "q_nucl.lookup")).read(), 9, 1)
    q = chain["dbs"]["q"]
    assert [orfs[k].rstrip("\n") for k in sorted(orfs)] == q["nucl"]
    assert [int(l.split("\t")[2]) for l in look.splitlines()] == q["set"]
    assert [so.translate_entry(x + "\n").rstrip("\n") for x in q["nucl"]] == q["aa"]
    t, c = chain["dbs"]["t"], chain["dbs"]["c"]
    pm.nucl_db(str(tmp_path / "t_nucl"), [[g] for g in genomes[:40]])
    orfs, _, look = so.extract_orfs(pm.read_db(str(tmp_path / "t_nucl")), open(str(tmp_path / "t_nucl.lookup")).read(), 30, 0)
    assert [orfs[k].rstrip("\n") for k in sorted(orfs)] == t["nucl"][:len(orfs)] and len(orfs) > 1000      # ORF keys are handed out genome by genome
    assert [int(l.split("\t")[2]) for l in look.splitlines()] == t["set"][:len(orfs)]
    for k in range(0, len(t["nucl"]), 7):
        assert so.translate_entry(t["nucl"][k] + "\n").rstrip("\n") == t["aa"][k]
        assert so.reverse_by_codon(t["nucl"][k] + "\n").rstrip("\n") == c["nucl"][k]
        assert so.translate_entry(c["nucl"][k] + "\n").rstrip("\n") == c["aa"][k]


@pytest.mark.parametrize("T,name", [("t", "score"), ("c", "score_rev")])
def test_score_entries_of_sampled_arrays_from_scratch(chain, T, name):
    import sp_sets_oracle as so
    from oracle_lib import Alp, Oracle
    dbs, got = chain["dbs"], chain[name]
    q, t = dbs["q"], dbs[T]
    g = json.load(gzip.open(os.path.join(ROOT, "tests/golden/spacer.json.gz")))["ref"]
    ex = json.load(gzip.open(os.path.join(ROOT, "tests/golden/examples.json.gz")))["info"]["nucl_evalue"]
    o = Oracle()
    o.set_targets(t["aa"])
    o.set_alp(Alp.from_ref(g["alpGapped"]), Alp.from_ref(g["alpGapless"]))
    ev = so.Evaluer([float.fromhex(x) for x in ex["alp"]], sum(len(x) for x in t["nucl"]))       # nucleotide.out 10/2: the constants do not depend on the data
    t_nuc = dict(enumerate(t["nucl"]))
    t_star = {k: v.startswith("*") for k, v in enumerate(t["aa"])}
    m2s = dict(enumerate(t["set"]))
    size = {}
    for s in t["set"]:
        size[s] = size.get(s, 0) + 1
    members = {}
    for k, s in enumerate(q["set"]):
        members.setdefault(s, []).append(k)
    nonempty = alignments = 0
    for a in list(range(0, 120, 11)) + [1, 2]:
        merged = ""
        for k in members[a]:
            alns = [o.align(q["aa"][k], h[0]) for h in o.prefilter_list(q["aa"][k])]
            alns = [x for x in alns if x.passed]
            alignments += len(alns)
            prot = o.alignment_text(alns)
            nucl = so.protein_aln_to_nucl(prot, q["nucl"][k], t_nuc, q["aa"][k].startswith("*"), t_star, ev)
            merged += so.best_hit_per_set(so.combine_prot_nucl(prot, nucl), m2s, size, True)
        exp = so.combine_score(merged, len(members[a]), m2s)
        assert got.get(a, "") == exp, a
        nonempty += exp != ""
    assert nonempty >= 6 and alignments > 30
    o.close()


def test_match_list_and_what_was_planted(chain):
    import sp_sets_oracle as so
    exp, thr = so.fdr_filter(chain["score"], chain["score_rev"], 0.05, 0)
    assert {k: v for k, v in exp.items() if v} == {k: v for k, v in chain["match"].items() if v}
    info = chain["info"]
    assert info["real_arrays"] == 60
    assert info["true_matches_found"] >= 54 and info["other_matches"] <= 3, info
