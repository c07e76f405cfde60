"""The set-level pin (rows f1-f4) compiles the reference's own module files against a plain parameter record, oracle/ref_par.h: RefPar, because the reference's
Parameters.cpp needs cmake-generated headers.  The record's defaults are typed, so this test reads the reference's defaults out of its sources
(lib/mmseqs/src/commons/Parameters.cpp:2077-2277, src/commons/LocalParameters.h:165-176) and compares them with the record's -- and the rescorediagonal defaults with what
this implementation's module assumes.  Runs where /root/reference is mounted (the build container); skipped on the GPU box."""
import os
import re

import pytest

REF = os.environ.get("SP_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "mmseqs", "src")), reason="reference sources not mounted")


def _assignments(text):
    """`name = value;` statements of a constructor body -> {name: value text}"""
    out = {}
    for m in re.finditer(r"^\s*([A-Za-z_][A-Za-z0-9_]*)\s*=\s*(.+?);", text, re.M):
        out.setdefault(m.group(1), m.group(2).strip())
    return out


def _norm(v):
    v = v.strip()
    v = re.sub(r"MultiParam<\s*(char\s*\*|int)\s*>\s*", "MultiParam", v)
    v = v.replace("0.05f", "0.05")
    return re.sub(r"\s+", "", v)


def test_the_parameter_record_of_the_set_level_pin_has_the_reference_defaults():
    par = open(os.path.join(REF, "lib/mmseqs/src/commons/Parameters.cpp")).read()
    body = par[par.index("void Parameters::setDefaults()"):]
    ref = _assignments(body)
    loc = _assignments(open(os.path.join(REF, "src/commons/LocalParameters.h")).read())
    rec_src = open(os.path.join(ROOT, "oracle", "ref_par.h")).read()
    rec_src = rec_src[rec_src.index("struct RefPar"):rec_src.index("template <typename... A> void parseParameters")]
    rec = {}
    # member initialisers `name = value` of the record: split at top-level commas / semicolons (values hold commas inside quotes and parentheses)
    rec_src = re.sub(r"//[^\n]*", "", rec_src)
    piece, depth, quote = "", 0, False
    for ch in rec_src:
        if ch == '"':
            quote = not quote
        if not quote:
            depth += ch in "(<" and 1 or 0
            depth -= ch in ")>" and 1 or 0
        if ch in ",;" and not quote and depth <= 0:
            m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*=\s*(.+)$", piece.strip(), re.S)
            if m:
                rec.setdefault(m.group(1), m.group(2).strip())
            piece, depth = "", 0
        else:
            piece += ch
    expect = {k: ref[k] for k in ("orfMinLength", "orfMaxLength", "orfMaxGaps", "contigStartMode", "contigEndMode", "orfStartMode", "forwardFrames", "reverseFrames",
                                  "useAllTableStarts", "translate", "translationTable", "createLookup", "addOrfStop", "simpleBestHit", "gapOpen", "gapExtend", "scoringMatrixFile")}
    expect["maxSeqLen"] = "65535"      # MAX_SEQ_LEN
    assert re.search(r"static const int MAX_SEQ_LEN\s*=\s*65535;", open(os.path.join(REF, "lib/mmseqs/src/commons/Parameters.h")).read())
    assert ref["maxSeqLen"].startswith("MAX_SEQ_LEN") and ref["compressed"] == "WRITER_ASCII_MODE" and ref["threads"] == "1"
    expect["compressed"], expect["threads"] = "0", "1"
    for k in ("fdrCutoff", "reportFdr", "formatType"):
        expect[k] = loc[k]
    bad = {k: (rec.get(k), v) for k, v in expect.items() if k not in rec or _norm(rec[k]) != _norm(v)}
    assert not bad, bad
    assert len(expect) >= 23


def test_the_rescorediagonal_defaults_are_the_reference_s():
    par = open(os.path.join(REF, "lib/mmseqs/src/commons/Parameters.cpp")).read()
    ref = _assignments(par[par.index("void Parameters::setDefaults()"):])
    assert ref["rescoreMode"] == "Parameters::RESCORE_MODE_HAMMING" and ref["filterHits"] == "false" and ref["wrappedScoring"] == "false" and ref["sortResults"] == "false"
    assert ref["evalThr"] == "0.001" and ref["covThr"] == "0.0" and ref["seqIdThr"] == "0.0" and ref["alnLenThr"] == "0" and ref["addBacktrace"] == "false"
    mod = open(os.path.join(ROOT, "spacepharer_amd", "csrc", "sp_modules.cpp")).read()
    body = mod[mod.index("int runRescoreDiagonalModule"):]
    for frag in ('f.num("--rescore-mode", 0)', 'f.real("-e", 0.001)', 'f.real("--min-seq-id", 0.0)', 'f.num("--min-aln-len", 0)', 'f.num("-a", 0) == 1', 'f.num("--sort-results", 0) > 0'):
        assert frag in body, frag
    h = open(os.path.join(REF, "lib/mmseqs/src/commons/Parameters.h")).read()
    modes = dict(re.findall(r"static const int (RESCORE_MODE_[A-Z_]+)\s*=\s*(\d)", h))
    assert modes == {"RESCORE_MODE_HAMMING": "0", "RESCORE_MODE_SUBSTITUTION": "1", "RESCORE_MODE_ALIGNMENT": "2", "RESCORE_MODE_GLOBAL_ALIGNMENT": "3", "RESCORE_MODE_WINDOW_QUALITY_ALIGNMENT": "4"}
