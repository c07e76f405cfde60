"""GPU: the module drop-ins (`prefilter`, `align`, `rescorediagonal`) run with the exact command lines SpacePHARER's
workflow issues on MMseqs2 DB triples and must write byte-identical entries to what the
reference modules produce (golden `pref` / `text` strings come from QueryMatcher::prefilterHitToBuffer and
Matcher::resultToBuffer of the real reference)."""
import os
import struct

import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

PREF_FLAGS = ("--sub-mat nucl:nucleotide.out,aa:VTML40.out --seed-sub-mat nucl:nucleotide.out,aa:VTML80.out -k 6 --k-score 2147483647 "
              "--alph-size nucl:5,aa:21 --max-seq-len 65535 --max-seqs %d --split 0 --split-mode 2 --split-memory-limit 0 -c 0 --cov-mode 0 "
              "--comp-bias-corr 1 --diag-score 1 --exact-kmer-matching 0 --mask 1 --mask-lower-case 0 --min-ungapped-score 15 "
              "--add-self-matches 0 --spaced-kmer-mode 1 --db-load-mode 0 --pca 1 --pcb 1.5 --spaced-kmer-pattern 11011101 --threads 8 "
              "--compressed 0 -v 3 -s 5.7")
ALN_FLAGS = ("--sub-mat nucl:nucleotide.out,aa:VTML40.out -a 1 --alignment-mode 2 --alignment-output-mode 0 --wrapped-scoring 0 -e 200 "
             "--min-seq-id 0 --min-aln-len 0 --seq-id-mode 0 --alt-ali 0 -c 0 --cov-mode 0 --max-seq-len 65535 --comp-bias-corr 1 "
             "--max-rejected 2147483647 --max-accept 2147483647 --add-self-matches 0 --db-load-mode 0 --pca 1 --pcb 1.5 --score-bias 0 "
             "--realign 0 --realign-score-bias -0.2 --realign-max-seqs 2147483647 --gap-open nucl:10,aa:16 --gap-extend 2 --zdrop 40 "
             "--threads 8 --compressed 0 -v 3")


def write_seq_db(path, seqs, key0=0, split=False, keys=None, order=None):
    """MMseqs2 sequence DB: residues + '\\n' + NUL per entry; index key\\toffset\\tlength; 4-byte dbtype (0 = amino acids).
    keys: DB key of sequence i (default key0 + i); order: the sequences' order in the data file (default 0..n-1) -- the index is
    always sorted by key, as DBWriter leaves it"""
    if keys is not None or order is not None:
        keys = keys if keys is not None else [key0 + i for i in range(len(seqs))]
        order = order if order is not None else list(range(len(seqs)))
        off, idx = 0, []
        with open(path, "wb") as f:
            for i in order:
                f.write(seqs[i].encode() + b"\n\0")
                idx.append((keys[i], off, len(seqs[i]) + 2))
                off += len(seqs[i]) + 2
        with open(path + ".index", "w") as f:
            for k, o, l in sorted(idx):
                f.write("%d\t%d\t%d\n" % (k, o, l))
        open(path + ".dbtype", "wb").write(struct.pack("<i", 0))
        return
    data = b"".join(s.encode() + b"\n\0" for s in seqs)
    if split:     # exercise the <p>.0 / <p>.1 split data files the real workflow leaves behind
        half = len(data) // 2
        open(path + ".0", "wb").write(data[:half])
        open(path + ".1", "wb").write(data[half:])
    else:
        open(path, "wb").write(data)
    off = 0
    with open(path + ".index", "w") as f:
        for i, s in enumerate(seqs):
            f.write("%d\t%d\t%d\n" % (key0 + i, off, len(s) + 2))
            off += len(s) + 2
    open(path + ".dbtype", "wb").write(struct.pack("<i", 0))


def read_db(path):
    if os.path.exists(path):
        data = open(path, "rb").read()
    else:                                   # split data files <path>.0, <path>.1, ... read back to back
        data, i = b"", 0
        while os.path.exists("%s.%d" % (path, i)):
            data += open("%s.%d" % (path, i), "rb").read()
            i += 1
    out = {}
    for line in open(path + ".index"):
        k, o, l = (int(x) for x in line.split("\t"))
        out[k] = data[o:o + l - 1].decode()
        assert data[o + l - 1] == 0
    return out, struct.unpack("<i", open(path + ".dbtype", "rb").read(4))[0]


@pytest.mark.parametrize("name", ["small", "family"])
def test_prefilter_align_modules_on_db_triples(name, tmp_path):
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    q, t, pref, aln = (str(tmp_path / x) for x in ("qdb", "tdb", "pref", "aln"))
    write_seq_db(q, g["queries"])
    write_seq_db(t, g["targets"], split=True)
    sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
    got, dbtype = read_db(pref)
    assert dbtype == 7
    assert sorted(got) == list(range(len(g["queries"])))            # every query key present, empty ones too
    for qq in r["queries"]:
        assert got[qq["id"]] == qq["pref"], qq["id"]
    sa.module_align([q, t, pref, aln] + ALN_FLAGS.split())
    got, dbtype = read_db(aln)
    assert dbtype == 5
    for a in r["aln"]:
        assert got[a["id"]] == a["text"], a["id"]
    # index is sorted by key and .dbtype exists only after the data is complete
    keys = [int(l.split("\t")[0]) for l in open(aln + ".index")]
    assert keys == sorted(keys)


def test_rescorediagonal_module(tmp_path):
    import spacepharer_amd as sa
    g = load_golden("small")
    r = g["ref"]
    q, t, pref, out1 = (str(tmp_path / x) for x in ("qdb", "tdb", "pref", "resc1"))
    write_seq_db(q, g["queries"])
    write_seq_db(t, g["targets"])
    sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
    sa.module_rescorediagonal([q, t, pref, out1, "--sub-mat", "nucl:nucleotide.out,aa:VTML40.out", "--rescore-mode", "1", "-e", "200",
                               "-c", "0", "--min-seq-id", "0", "--threads", "4", "--compressed", "0", "-v", "3"])
    got, dbtype = read_db(out1)
    assert dbtype == 7
    for a in r["aln"]:
        exp = ""
        for raw in a["rescore"]:
            dbKey, s1, s2, st, en, dlen, dist, diag, ev, bits = raw
            if float.fromhex(ev) <= 200.0:
                d16 = ((diag + 32768) % 65536) - 32768
                exp += "%d\t%d\t%d\n" % (dbKey, bits, d16)
        assert got[a["id"]] == exp, a["id"]


def test_module_fails_without_touching_dbtype_on_error(tmp_path):
    import spacepharer_amd as sa
    q, t, pref = (str(tmp_path / x) for x in ("qdb", "tdb", "pref"))
    write_seq_db(q, ["ACDEFGHIKL"])
    with pytest.raises(sa.SpError):
        sa.module_prefilter([q, t, pref])          # target DB missing
    assert not os.path.exists(pref + ".dbtype")


def test_modules_fan_out_over_gpus_with_identical_results(tmp_path):
    """SP_GPUS=N: the front-end binary forks one worker per GPU (here 3 workers sharing the one GPU of the box), each
    writes the data file <out>.<rank>, the parent merges the indexes -- the decoded DBs equal the reference's text"""
    import subprocess
    g = load_golden("family")
    r = g["ref"]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spacepharer_amd", "spacepharer-amd")
    q, t, pref, aln = (str(tmp_path / x) for x in ("qdb", "tdb", "pref", "aln"))
    write_seq_db(q, g["queries"])
    write_seq_db(t, g["targets"])
    env = dict(os.environ, SP_GPUS="3")
    p = subprocess.run([exe, "prefilter", q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split(), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert all(os.path.exists("%s.%d" % (pref, i)) for i in range(3)) and not os.path.exists(pref)
    got, dbtype = read_db(pref)
    assert dbtype == 7 and sorted(got) == list(range(len(g["queries"])))
    for qq in r["queries"]:
        assert got[qq["id"]] == qq["pref"], qq["id"]
    p = subprocess.run([exe, "align", q, t, pref, aln] + ALN_FLAGS.split(), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got, dbtype = read_db(aln)
    assert dbtype == 5
    for a in r["aln"]:
        assert got[a["id"]] == a["text"], a["id"]
    keys = [int(l.split("\t")[0]) for l in open(aln + ".index")]
    assert keys == sorted(keys) and len(keys) == len(g["queries"])
    assert not [f for f in os.listdir(tmp_path) if f.endswith("part")]       # the private indexes are gone


def test_module_fan_out_from_a_process_that_already_initialised_hip(tmp_path):
    """SP_GPUS=2 through the C ABI of THIS process, which has initialised the HIP runtime long ago (sp_init, live contexts in
    other tests): the workers are fresh processes of the front-end binary, never forks of the initialised runtime.  Stale
    outputs of an earlier run (an unsplit data file, a part of a wider split) must not survive."""
    import spacepharer_amd as sa
    g = load_golden("small")
    r = g["ref"]
    q, t, pref = (str(tmp_path / x) for x in ("qdb", "tdb", "pref"))
    write_seq_db(q, g["queries"])
    write_seq_db(t, g["targets"])
    c = sa.SearchContext()                     # the runtime is certainly up
    open(pref, "wb").write(b"stale unsplit data file")
    open(pref + ".5", "wb").write(b"stale part of an 8-way split")
    os.environ["SP_GPUS"] = "2"
    try:
        sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
    finally:
        os.environ.pop("SP_GPUS", None)
    c.close()
    assert not os.path.exists(pref) and not os.path.exists(pref + ".5") and os.path.exists(pref + ".0") and os.path.exists(pref + ".1")
    got, dbtype = read_db(pref)
    assert dbtype == 7 and sorted(got) == list(range(len(g["queries"])))
    for qq in r["queries"]:
        assert got[qq["id"]] == qq["pref"], qq["id"]
    # a later unsplit run over the same path removes the parts again
    sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
    assert os.path.exists(pref) and not os.path.exists(pref + ".0")
    got, _ = read_db(pref)
    for qq in r["queries"]:
        assert got[qq["id"]] == qq["pref"], qq["id"]


def test_rescorediagonal_module_alignment_format(tmp_path):
    """--rescore-mode 2 -a 1, what `search --alignment-mode 4` passes (lib/mmseqs/src/workflow/Search.cpp:339): the alignment-format
    branch of rescorediagonal (rescorediagonal.cpp:254-342) -- Matcher::resultToBuffer lines with an `<n>M` backtrace -- unsorted and
    with --sort-results 1"""
    import spacepharer_amd as sa
    for name in ("small", "longt"):
        g = load_golden(name)
        r = g["ref"]
        q, t, pref = (str(tmp_path / (name + x)) for x in ("qdb", "tdb", "pref"))
        write_seq_db(q, g["queries"])
        write_seq_db(t, g["targets"])
        sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
        for sort, key in (("0", "rescore2"), ("1", "rescore2sorted")):
            out = str(tmp_path / (name + "resc2_" + sort))
            sa.module_rescorediagonal([q, t, pref, out, "--sub-mat", "nucl:nucleotide.out,aa:VTML40.out", "--rescore-mode", "2", "-a", "1", "-e", "200", "-c", "0",
                                       "--min-seq-id", "0", "--sort-results", sort, "--threads", "4", "--compressed", "0", "-v", "3"])
            got, dbtype = read_db(out)
            assert dbtype == 5
            n = 0
            for a in r["aln"]:
                assert got[a["id"]] == a[key], (name, key, a["id"])
                n += a[key].count("\n")
            assert n > 10


def test_rescorediagonal_module_with_the_reference_defaults_and_the_other_modes(tmp_path):
    """a bare `rescorediagonal q t pref out` -- RESCORE_MODE_HAMMING, -e 0.001, no coverage / identity / length thresholds (commons/Parameters.cpp:2219-2260) -- writes
    prefilter-format lines whose score is 100 x the fraction of equal letters on the diagonal (rescorediagonal.cpp:243-252, 354-360); --rescore-mode 3 (global) and 4
    (window quality) write alignment records like mode 2 (DistanceCalculator.h:203-272)"""
    import spacepharer_amd as sa
    for name in ("small", "longt", "spacer"):
        g = load_golden(name)
        r = g["ref"]
        q, t, pref = (str(tmp_path / (name + x)) for x in ("qdb", "tdb", "pref"))
        write_seq_db(q, g["queries"])
        write_seq_db(t, g["targets"])
        sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
        out = str(tmp_path / (name + "resc0"))
        sa.module_rescorediagonal([q, t, pref, out])
        got, dbtype = read_db(out)
        assert dbtype == 7
        n = 0
        for a in r["aln"]:
            assert got[a["id"]] == a["rescore0"], (name, a["id"])
            n += a["rescore0"].count("\n")
        assert n > 10
        for mode in (3, 4):
            out = str(tmp_path / (name + "resc%d" % mode))
            sa.module_rescorediagonal([q, t, pref, out, "--sub-mat", "nucl:nucleotide.out,aa:VTML40.out", "--rescore-mode", str(mode), "-a", "1", "-e", "200", "-c", "0",
                                       "--min-seq-id", "0", "--threads", "4", "--compressed", "0", "-v", "3"])
            got, dbtype = read_db(out)
            assert dbtype == 5
            for a in r["aln"]:
                assert got[a["id"]] == a["rescore%d" % mode], (name, mode, a["id"])


def test_modules_with_sparse_keys_and_shuffled_entries(tmp_path):
    """DB keys are not positions: target keys sparse (the sorted-list branch of the key lookup), query keys neither dense nor monotone, the
    query entries shuffled in the data file, the index sorted by key as DBWriter leaves it.  Results are keyed by query key and name
    targets by their keys; inside an entry nothing changes (sequence ids are positions in data order, Prefiltering.cpp:164)."""
    import random
    import spacepharer_amd as sa
    g = load_golden("family")
    r = g["ref"]
    rng = random.Random(5)
    nq, nt = len(g["queries"]), len(g["targets"])
    tkeys = [1000 * i + 17 for i in range(nt)]
    qkeys = rng.sample(range(3, 5 * 10 ** 6), nq)
    order = list(range(nq))
    rng.shuffle(order)
    q, t, pref, aln, resc = (str(tmp_path / x) for x in ("qdb", "tdb", "pref", "aln", "resc"))
    write_seq_db(q, g["queries"], keys=qkeys, order=order)
    write_seq_db(t, g["targets"], keys=tkeys)
    tmap = lambda text: "".join(str(tkeys[int(ln.split("\t", 1)[0])]) + "\t" + ln.split("\t", 1)[1] + "\n" for ln in text.splitlines())      # noqa: E731
    sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
    got, _ = read_db(pref)
    assert sorted(got) == sorted(qkeys)
    for qq in r["queries"]:
        assert got[qkeys[qq["id"]]] == tmap(qq["pref"]), qq["id"]
    sa.module_align([q, t, pref, aln] + ALN_FLAGS.split())
    got, _ = read_db(aln)
    for a in r["aln"]:
        assert got[qkeys[a["id"]]] == tmap(a["text"]), a["id"]
    sa.module_rescorediagonal([q, t, pref, resc, "--sub-mat", "nucl:nucleotide.out,aa:VTML40.out", "--rescore-mode", "2", "-a", "1", "-e", "200", "-c", "0",
                               "--min-seq-id", "0", "--threads", "4", "--compressed", "0", "-v", "3"])
    got, _ = read_db(resc)
    for a in r["aln"]:
        assert got[qkeys[a["id"]]] == tmap(a["rescore2"]), a["id"]


@pytest.mark.parametrize("name", ["small", "family", "longq"])
def test_fused_search_module_writes_the_align_db(name, tmp_path):
    """`search <queryDB> <targetDB> <alnDB> <tmp>` = prefilter + align on one resident context, with the reference's search flag list
    (lib/mmseqs/src/workflow/Search.cpp): the alignment DB must be the one the reference's two modules produce -- the golden text --
    and byte-identical, index included, to this implementation's own two-step run"""
    import spacepharer_amd as sa
    g = load_golden(name)
    r = g["ref"]
    q, t, pref, aln, fused = (str(tmp_path / x) for x in ("qdb", "tdb", "pref", "aln", "fused"))
    write_seq_db(q, g["queries"], key0=11)
    write_seq_db(t, g["targets"], split=True)
    flags = (PREF_FLAGS % r["maxSeqs"]).split() + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--min-seq-id", "0", "--min-aln-len", "0", "--max-rejected", "2147483647",
                                                   "--max-accept", "2147483647", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2", "--num-iterations", "1", "--search-type", "0",
                                                   "--start-sens", "4", "--sens-steps", "1", "--exhaustive-search", "0", "--remove-tmp-files", "0", "--rescore-mode", "0", "--strand", "1"]
    sa.module_search([q, t, fused, str(tmp_path)] + flags)
    got, dbtype = read_db(fused)
    assert dbtype == 5 and sorted(got) == [11 + i for i in range(len(g["queries"]))]
    for a in r["aln"]:
        assert got[11 + a["id"]] == a["text"], a["id"]
    sa.module_prefilter([q, t, pref] + (PREF_FLAGS % r["maxSeqs"]).split())
    sa.module_align([q, t, pref, aln] + ALN_FLAGS.split())
    assert open(aln, "rb").read() == open(fused, "rb").read() and open(aln + ".index").read() == open(fused + ".index").read()
    # the entries above were formatted on the device (csrc/sp_gpu_text.inc) where the batch allows it; the host's formatters (SP_DEVICE_TEXT=0) must write the same files
    os.environ["SP_DEVICE_TEXT"] = "0"
    try:
        hpref, hfused = str(tmp_path / "hpref"), str(tmp_path / "hfused")
        sa.module_search([q, t, hfused, str(tmp_path)] + flags)
        sa.module_prefilter([q, t, hpref] + (PREF_FLAGS % r["maxSeqs"]).split())
    finally:
        del os.environ["SP_DEVICE_TEXT"]
    assert open(hfused, "rb").read() == open(fused, "rb").read() and open(hfused + ".index").read() == open(fused + ".index").read()
    assert open(hpref, "rb").read() == open(pref, "rb").read() and open(hpref + ".index").read() == open(pref + ".index").read()
    with pytest.raises(sa.SpError):
        sa.module_search([q, t, fused + "x", str(tmp_path)] + flags + ["--num-iterations", "3"])      # iterative profile search: not this path
    assert not os.path.exists(fused + "x.dbtype")


@pytest.mark.parametrize("name,gpus", [("small", 2), ("family", 3), ("longq", 4)])
def test_search_module_over_target_shards(name, gpus, tmp_path):
    """SP_GPUS=N SP_SHARD=targets: the N workers of the `search` module each index 1/N of the TARGET set, match all queries against their shard, all-gather the
    candidates through the library's communicator (sp_dist: RCCL with a GPU per worker; here the workers share the box's one GPU and the host transport carries the
    device buffers) and finish their slice of the queries -- the merged alignment DB equals the reference's text and the unsplit run's entries"""
    import subprocess
    g = load_golden(name)
    r = g["ref"]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spacepharer_amd", "spacepharer-amd")
    q, t, fused = (str(tmp_path / x) for x in ("qdb", "tdb", "fused"))
    write_seq_db(q, g["queries"], key0=5)
    write_seq_db(t, g["targets"])
    flags = (PREF_FLAGS % r["maxSeqs"]).split() + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--min-seq-id", "0", "--min-aln-len", "0", "--max-rejected", "2147483647",
                                                   "--max-accept", "2147483647", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2", "--num-iterations", "1", "--search-type", "0",
                                                   "--start-sens", "4", "--sens-steps", "1", "--exhaustive-search", "0", "--remove-tmp-files", "0", "--rescore-mode", "0", "--strand", "1"]
    env = dict(os.environ, SP_GPUS=str(gpus), SP_SHARD="targets")
    p = subprocess.run([exe, "search", q, t, fused, str(tmp_path)] + flags, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    got, dbtype = read_db(fused)
    assert dbtype == 5 and sorted(got) == [5 + i for i in range(len(g["queries"]))]
    for a in r["aln"]:
        assert got[5 + a["id"]] == a["text"], (name, a["id"])
    assert all(os.path.exists("%s.%d" % (fused, i)) for i in range(gpus))


def test_module_calls_through_a_resident_server(tmp_path):
    """`spacepharer-amd serve <socket>` + clients with SP_SERVER set: the client forwards argv and working directory, the server runs
    the module with its runtime up and the target set cached in HBM, the exit status and the error text come back.  Same DB bytes as
    a direct call; relative paths resolve in the client's directory; an error leaves no .dbtype and a non-zero status; without a
    reachable server the client runs the module itself."""
    import subprocess
    import time
    g = load_golden("small")
    r = g["ref"]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spacepharer_amd", "spacepharer-amd")
    write_seq_db(str(tmp_path / "qdb"), g["queries"])
    write_seq_db(str(tmp_path / "tdb"), g["targets"])
    sock = str(tmp_path / "sp.sock")
    srv = subprocess.Popen([exe, "serve", sock], stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(3000):
            if os.path.exists(sock):
                break
            time.sleep(0.01)
        assert os.path.exists(sock)
        env = dict(os.environ, SP_SERVER=sock)
        flags = (PREF_FLAGS % r["maxSeqs"]).split() + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]
        for rep in range(3):          # the second and third call find the target set resident
            p = subprocess.run([exe, "search", "qdb", "tdb", "aln%d" % rep, "."] + flags, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr
            got, dbtype = read_db(str(tmp_path / ("aln%d" % rep)))
            assert dbtype == 5
            for a in r["aln"]:
                assert got[a["id"]] == a["text"], a["id"]
        p = subprocess.run([exe, "prefilter", "qdb", "missing", "pref"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        assert p.returncode != 0 and "missing" in p.stderr and not os.path.exists(str(tmp_path / "pref.dbtype"))
        p = subprocess.run([exe, "translatenucs", "qdb", "x", "--no-such-flag", "1"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        assert p.returncode != 0 and "Unrecognized" in p.stderr
        subprocess.run([exe, "shutdown"], env=env, capture_output=True, timeout=60)
        assert srv.wait(timeout=120) == 0 and not os.path.exists(sock)
        # no server any more: the same command line runs in the client process
        p = subprocess.run([exe, "search", "qdb", "tdb", "aln_direct", "."] + flags, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        assert open(str(tmp_path / "aln_direct"), "rb").read() == open(str(tmp_path / "aln0"), "rb").read()
    finally:
        if srv.poll() is None:
            srv.kill()


def test_auto_started_server_and_align_on_resident_hits(tmp_path):
    """SP_SERVER=auto: the first module process finds no server on the per-user socket, starts one (detached, idle time-out) and forwards; `align`
    on the result DB the server's `prefilter` just wrote aligns the hits that are still resident (no DB read back) -- same bytes as fresh
    processes; a result DB that changed on disk, other align parameters or another query DB take the ordinary path; shuffled / sparse query keys;
    the server leaves by itself when idle; a stale socket file is replaced, a regular file at the socket path is left alone."""
    import random
    import subprocess
    import time
    g = load_golden("spacer")
    r = g["ref"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "spacepharer_amd", "spacepharer-amd")
    rnd = random.Random(5)
    qkeys = rnd.sample(range(10, 10 * len(g["queries"])), len(g["queries"]))      # sparse, shuffled keys: the index is sorted by key, the data is not
    write_seq_db(str(tmp_path / "qdb"), g["queries"], keys=qkeys)
    write_seq_db(str(tmp_path / "qdb2"), g["queries"][::-1], keys=qkeys)
    write_seq_db(str(tmp_path / "tdb"), g["targets"])
    pf = (PREF_FLAGS % r["maxSeqs"]).split()
    af = ALN_FLAGS.split()
    direct = dict(os.environ)
    direct.pop("SP_SERVER", None)

    def pair(env, tag, qdb="qdb", aflags=af):
        for mod, args in (("prefilter", [qdb, "tdb", "pref" + tag] + pf), ("align", [qdb, "tdb", "pref" + tag, "aln" + tag] + aflags)):
            p = subprocess.run([exe, mod] + args, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, (mod, p.stderr)
        return open(str(tmp_path / ("aln" + tag)), "rb").read(), open(str(tmp_path / ("aln" + tag + ".index"))).read()

    ref = pair(direct, "_ref")
    expect = {qkeys[a["id"]]: a["text"] for a in r["aln"]}
    got, _ = read_db(str(tmp_path / "aln_ref"))
    assert all(got[k] == v for k, v in expect.items()) and len(got) == len(qkeys)
    rt = tmp_path / "rt"
    rt.mkdir()
    sock = str(rt / "spacepharer-amd.sock")
    open(sock, "w").write("not a socket")      # something else at the path: the server refuses, the clients run the modules themselves
    env = dict(os.environ, SP_SERVER="auto:4", XDG_RUNTIME_DIR=str(rt), SP_SERVER_LOG=str(tmp_path / "server.log"), SP_MODULE_TIMES="1")
    assert pair(env, "_nosrv") == ref and open(sock).read() == "not a socket"
    os.remove(sock)
    assert pair(env, "_a") == ref                       # starts the server, prefilter + align through it
    assert os.path.exists(sock) and (os.stat(sock).st_mode & 0o077) == 0
    log = open(str(tmp_path / "server.log")).read()
    assert "resident hits" in log                        # the align call used what prefilter left in HBM
    assert pair(env, "_b") == ref
    # the ordinary path behind the same server: another E-value threshold (not what the context aligns with) ...
    af2 = ALN_FLAGS.replace("-e 200", "-e 0.5").split()
    assert "-e 0.5" in " ".join(af2)
    n_res = open(str(tmp_path / "server.log")).read().count("resident hits")
    other = pair(env, "_c", aflags=af2)
    assert other == pair(direct, "_c_ref", aflags=af2) and other != ref
    assert open(str(tmp_path / "server.log")).read().count("resident hits") == n_res
    # ... a result DB somebody rewrote after the prefilter call ...
    p = subprocess.run([exe, "prefilter", "qdb", "tdb", "pref_d"] + pf, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    data = open(str(tmp_path / "pref_d"), "rb").read()
    time.sleep(0.02)
    open(str(tmp_path / "pref_d"), "wb").write(data)
    p = subprocess.run([exe, "align", "qdb", "tdb", "pref_d", "aln_d"] + af, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and open(str(tmp_path / "aln_d"), "rb").read() == ref[0]
    assert open(str(tmp_path / "server.log")).read().count("resident hits") == n_res
    # ... and another query DB under the memo's result name
    assert pair(env, "_e", qdb="qdb2") == pair(direct, "_e_ref", qdb="qdb2")
    # ... a query DB rewritten IN PLACE (same path, same sizes, other sequences) behind the server's cache of flattened sequence DBs, and back
    import shutil
    for ext in ("", ".index", ".dbtype"):
        shutil.copy(str(tmp_path / ("qdb" + ext)), str(tmp_path / ("qdb_keep" + ext)))
    time.sleep(0.02)
    write_seq_db(str(tmp_path / "qdb"), g["queries"][::-1], keys=qkeys)
    assert pair(env, "_g") == pair(direct, "_g_ref") and pair(env, "_g2") != ref
    time.sleep(0.02)
    for ext in ("", ".index", ".dbtype"):
        shutil.copy(str(tmp_path / ("qdb_keep" + ext)), str(tmp_path / ("qdb" + ext)))
    assert pair(env, "_h") == ref
    # idle: the server leaves and takes its socket with it; the next call starts a new one
    for _ in range(200):
        if not os.path.exists(sock):
            break
        time.sleep(0.1)
    assert not os.path.exists(sock)
    assert pair(env, "_f") == ref and os.path.exists(sock)
    subprocess.run([exe, "shutdown"], env=env, capture_output=True, timeout=60)
    for _ in range(100):
        if not os.path.exists(sock):
            break
        time.sleep(0.1)
    assert not os.path.exists(sock)


def test_modules_take_a_query_beyond_the_hit_stream_limit_through_target_shards(tmp_path):
    """a query with more than 65 535 index hits (tests/test_gpu_sharded.py builds the same case): the in-memory call refuses its batch,
    the module drop-ins search the batch without it and take it through target shards -- every entry equals the oracle's / the
    two-step run's, the refused query included"""
    import random
    import spacepharer_amd as sa
    from oracle_lib import Oracle
    rnd = random.Random(11)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    core = "".join(rnd.choice(aa) for _ in range(60))
    T = []
    for i in range(2400):
        s = list(core)
        for _ in range(3):
            s[rnd.randrange(60)] = rnd.choice(aa)
        T.append("".join(rnd.choice(aa) for _ in range(rnd.randrange(0, 20))) + "".join(s) + "".join(rnd.choice(aa) for _ in range(20)))
    Q = ["".join(rnd.choice(aa) for _ in range(12)), core, "".join(rnd.choice(aa) for _ in range(10)), core[5:55]]
    q, t, pref, aln, fused = (str(tmp_path / x) for x in ("qdb", "tdb", "pref", "aln", "fused"))
    write_seq_db(q, Q, key0=3)
    write_seq_db(t, T)
    sa.module_prefilter([q, t, pref] + (PREF_FLAGS % 300).split())
    got, dbtype = read_db(pref)
    o = Oracle()
    o.set_targets(T)
    assert dbtype == 7 and sorted(got) == [3, 4, 5, 6]
    for i, s in enumerate(Q):
        assert got[3 + i] == o.prefilter_text(s), i
    assert got[4].count("\n") == 300
    o.close()
    sa.module_align([q, t, pref, aln] + ALN_FLAGS.split())
    flags = (PREF_FLAGS % 300).split() + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--min-seq-id", "0", "--min-aln-len", "0", "--max-rejected", "2147483647",
                                          "--max-accept", "2147483647", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]
    sa.module_search([q, t, fused, str(tmp_path)] + flags)
    two, _ = read_db(aln)
    one, dbtype = read_db(fused)
    assert dbtype == 5 and one == two and one[4].count("\n") > 100


def test_a_fresh_process_recycles_device_blocks_and_writes_the_same_db(tmp_path):
    """a one-shot front-end process hands device blocks it is done with (similar-3-mer table scratch, index-build scratch) to later buffers (DevPool, csrc/sp_gpu.hip)
    instead of allocating again: the alignment DB must be byte-identical with the pool off, and the trace must show that blocks WERE recycled"""
    import subprocess
    import bench
    tres, toff, qres, qoff = bench.make_workload(20000, 50000, 200, 4321)
    T = [tres[int(toff[i]):int(toff[i + 1])].tobytes().decode() for i in range(len(toff) - 1)]
    Q = [qres[int(qoff[i]):int(qoff[i + 1])].tobytes().decode() for i in range(len(qoff) - 1)]
    q, t = str(tmp_path / "q"), str(tmp_path / "t")
    write_seq_db(q, Q)
    write_seq_db(t, T)
    flags = (PREF_FLAGS % 300).split() + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spacepharer_amd", "spacepharer-amd")
    outs = {}
    for name, extra in (("pool", {}), ("nopool", {"SP_NO_DEVICE_POOL": "1"})):
        env = dict(os.environ, SP_MEM_TRACE="1", **extra)
        env.pop("SP_SERVER", None)
        p = subprocess.run([exe, "search", q, t, str(tmp_path / name), str(tmp_path)] + flags, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = (open(str(tmp_path / name), "rb").read(), open(str(tmp_path / name) + ".index").read(), p.stderr)
    assert outs["pool"][0] == outs["nopool"][0] and outs["pool"][1] == outs["nopool"][1] and len(outs["pool"][0]) > 100000
    assert "recycled" in outs["pool"][2] and "recycled" not in outs["nopool"][2]


@pytest.mark.parametrize("gpus", [2, 3])
def test_target_sharded_search_takes_an_overflowing_query_through_sub_shards(gpus, tmp_path):
    """SP_GPUS=N SP_SHARD=targets with a query whose hit stream overflows ONE rank's shard (a family of near-identical targets sits in the first shard only): the ranks agree
    on the refused queries, search the rest as usual and take the refused ones through sub-shards of the whole target set -- every entry equals the unsplit run's
    (round 5 ended the job on all ranks; the reference's own overflow path is QueryMatcher.cpp:254-283)"""
    import random
    import subprocess
    import spacepharer_amd as sa
    rnd = random.Random(23)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    core = "".join(rnd.choice(aa) for _ in range(60))
    T = []
    for i in range(2600):      # the family: shard 0 of 2 (and of 3) holds all of it
        s = list(core)
        for _ in range(3):
            s[rnd.randrange(60)] = rnd.choice(aa)
        T.append("".join(rnd.choice(aa) for _ in range(rnd.randrange(0, 20))) + "".join(s) + "".join(rnd.choice(aa) for _ in range(20)))
    background = ["".join(rnd.choice(aa) for _ in range(rnd.randrange(60, 140))) for _ in range(6000)]
    T += background
    Q = ["".join(rnd.choice(aa) for _ in range(12)), core, background[17][10:24], core[5:55], background[4000][30:42], background[5999][5:20]]
    q, t, one, many = (str(tmp_path / x) for x in ("qdb", "tdb", "one", "many"))
    write_seq_db(q, Q, key0=3)
    write_seq_db(t, T)
    flags = (PREF_FLAGS % 300).split() + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--min-seq-id", "0", "--min-aln-len", "0", "--max-rejected", "2147483647",
                                          "--max-accept", "2147483647", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]
    sa.module_search([q, t, one, str(tmp_path)] + flags)      # unsplit, in this process
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spacepharer_amd", "spacepharer-amd")
    env = dict(os.environ, SP_GPUS=str(gpus), SP_SHARD="targets", SP_MODULE_TIMES="1")
    env.pop("SP_SERVER", None)
    p = subprocess.run([exe, "search", q, t, many, str(tmp_path)] + flags, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "refused queries through sub-shards" in p.stderr      # (the fallback did run)
    a, dbtype = read_db(one)
    b, dbtype2 = read_db(many)
    assert dbtype == 5 and dbtype2 == 5 and sorted(a) == [3 + i for i in range(len(Q))]
    assert a == b
    assert a[4].count("\n") > 100 and a[5].count("\n") >= 1
