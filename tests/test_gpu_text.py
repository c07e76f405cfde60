"""GPU: the result-DB text formatted on the device (csrc/sp_gpu_text.inc) against the text the REFERENCE's own formatters produced for the golden cases
(QueryMatcher::prefilterHitToBuffer and Matcher::resultToBuffer through oracle/ref_harness.cpp: fields "pref" and "text" of the fixtures), byte for byte --
number formats, the sequence-identity and E-value spellings, run-length backtraces, acceptance and the order of a query's alignments."""
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = ["small", "spacer", "family", "longq", "family16", "longt", "sens75", "sens1"]


@pytest.mark.parametrize("name", CASES)
def test_prefilter_and_alignment_entries_equal_the_reference_text(name):
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    ctx = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"], bin_size=r["binSize"], evalue=200.0)
    ctx.set_targets(g["targets"])
    ctx.set_queries(g["queries"])
    ctx.prefilter()
    pref = ctx.hits_text()
    exp_pref = {x["id"]: x["pref"] for x in r["queries"]}
    assert len(pref) == len(g["queries"])
    lines = 0
    for q, got in enumerate(pref):
        assert got.decode() == exp_pref.get(q, ""), (q, got[:120])
        lines += got.count(b"\n")
    assert lines > 0
    exp_aln = {x["id"]: x["text"] for x in r["aln"]}
    if name == "longq":      # (a long gapped alignment takes the host-driven traceback there: the device text is refused, loudly, and the host formats)
        with pytest.raises(RuntimeError, match="host"):
            ctx.align_hits_text()
    else:
        aln = ctx.align_hits_text()
        import os
        for force in ("0", "1"):      # the lane-per-query and the wave-per-query measuring kernels on the same batch
            os.environ["SP_TEXT_WAVE"] = force
            try:
                assert ctx.align_hits_text() == aln, force
            finally:
                del os.environ["SP_TEXT_WAVE"]
        gapped = 0
        for q, got in enumerate(aln):
            assert got.decode() == exp_aln.get(q, ""), (q, got[:160], exp_aln.get(q, "")[:160])
            gapped += (b"I" in got) or (b"D" in got)
        if name == "family":
            assert gapped > 0
        # acceptance thresholds: a stricter --min-seq-id / --min-aln-len drops lines, never reorders the rest
        strict = ctx.align_hits_text(min_seq_id=0.8, min_aln_len=9)
        for q, got in enumerate(strict):
            kept = [ln for ln in exp_aln.get(q, "").split("\n") if ln and float(ln.split("\t")[2]) >= 0.8]
            assert [ln.split("\t")[0] for ln in got.decode().split("\n") if ln] == [ln.split("\t")[0] for ln in kept if _aln_len(ln) >= 9], q
    ctx.close()


def _aln_len(line):
    import re
    return sum(int(n) for n in re.findall(r"(\d+)[MID]", line.split("\t")[10]))


def test_text_that_would_not_fit_32_bit_offsets_is_declined_before_anything_is_written():
    """Entry lengths / offsets on the device are 32-bit: a batch whose text reaches the limit (4 GiB; SP_TEXT_LIMIT lowers it for this test, read once per process) must be refused with
    SP_ERR_UNSUPPORTED -- the module drivers then format on the host -- and never wrap."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import spacepharer_amd as sa
from conftest import load_golden
g = load_golden("small"); r = g["ref"]
ctx = sa.SearchContext(sensitivity=r["sens"], max_seqs=r["maxSeqs"], bin_size=r["binSize"], evalue=200.0)
ctx.set_targets(g["targets"]); ctx.set_queries(g["queries"]); ctx.prefilter()
assert sum(map(len, ctx.hits())) > 10
for call in (ctx.hits_text, ctx.align_hits_text):
    try:
        call()
    except sa.SpError as e:
        assert "4 GiB" in str(e) and "code -6" in str(e), str(e)
    else:
        raise SystemExit("the device text was not declined")
print("declined twice")
''' % (root, os.path.join(root, "tests"))
    env = dict(os.environ, SP_TEXT_LIMIT="64")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "declined twice" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]
