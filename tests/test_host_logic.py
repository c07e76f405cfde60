"""CPU-side checks of the PRODUCT library (no GPU, no compute calls on the device): the shared object loads, exports every
symbol the public header declares, rejects bad command lines, and its host set-up code (matrices, similar-k-mer table,
tantan masking, k-mer index, E-value statistics) reproduces the reference-generated golden digests."""
import ctypes as C
import os
import re

import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import spacepharer_amd as sa
    if not os.path.exists(sa.library_path()):
        sa.build_library()
    return sa.load_library()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "spacepharer_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.sp_version()


def test_modules_reject_bad_command_lines(lib):
    def run(fn, args):
        arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
        return getattr(lib, fn)(len(args), arr), lib.sp_last_error().decode()
    rc, err = run("sp_prefilter_run", ["q", "t", "o", "--no-such-flag", "1"])
    assert rc != 0 and "Unrecognized parameter" in err
    rc, err = run("sp_prefilter_run", ["q", "t"])
    assert rc != 0 and "database arguments" in err
    rc, err = run("sp_prefilter_run", ["q", "t", "o", "--exact-kmer-matching", "1"])
    assert rc != 0 and "unsupported value" in err
    rc, err = run("sp_align_run", ["q", "t", "r", "o", "--alt-ali", "3"])
    assert rc != 0 and "unsupported value" in err
    rc, err = run("sp_rescorediagonal_run", ["q", "t", "r", "o", "--rescore-mode", "5"])
    assert rc != 0 and "unsupported value" in err
    rc, err = run("sp_rescorediagonal_run", ["q", "t", "r", "o", "--filter-hits", "1"])
    assert rc != 0 and "unsupported value" in err
    rc, err = run("sp_prefilter_run", ["/nonexistent/q", "/nonexistent/t", "/tmp/o"])
    assert rc != 0 and "cannot open" in err


def test_host_matrices(lib):
    g = load_golden("small")["ref"]
    lib.sp_host_matrix.argtypes = [C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_int16), C.POINTER(C.c_double)]
    for spec, bf, sb, key in ((b"VTML80.out", 8.0, float(C.c_float(-0.2).value), "seedMat"),
                              (b"VTML40.out", 2.0, float(C.c_float(-0.2).value), "ungappedMat"), (b"VTML40.out", 2.0, 0.0, "alnMat")):
        sub = (C.c_int16 * 441)()
        pb = (C.c_double * 21)()
        assert lib.sp_host_matrix(spec, bf, sb, sub, pb) == 0
        assert list(sub) == g[key]["sub"]
        assert list(pb) == [float.fromhex(x) for x in g[key]["pBack"]]
    assert lib.sp_host_matrix(b"NOPE.out", 2.0, 0.0, (C.c_int16 * 441)(), (C.c_double * 21)()) != 0


def test_host_ext3_table(lib):
    g = load_golden("small")["ref"]["ext3"]
    lib.sp_host_ext3_digest.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int16), C.POINTER(C.c_uint32), C.c_uint32]
    dig = C.c_uint64()
    sc = (C.c_int16 * 64)()
    ix = (C.c_uint32 * 64)()
    assert lib.sp_host_ext3_digest(b"VTML80.out", C.byref(dig), sc, ix, 4321) == 0
    assert "%016x" % dig.value == g["digest"]
    assert list(sc) == g["rows"]["4321"]["score"] and list(ix) == g["rows"]["4321"]["index"]


@pytest.mark.parametrize("name", ["small", "spacer", "family", "longq", "longt"])
def test_host_masking_and_index(lib, name):
    g = load_golden(name)
    r = g["ref"]
    lib.sp_host_index_digest.argtypes = [C.c_char_p, C.c_float, C.c_int, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p]
    T = [t.encode() for t in g["targets"]]
    offs = (C.c_uint64 * (len(T) + 1))()
    for i, t in enumerate(T):
        offs[i + 1] = offs[i] + len(t)
    buf = b"".join(T)
    ent, msk, dig = C.c_uint64(), C.c_uint64(), C.c_uint64()
    masked = C.create_string_buffer(len(buf) + 1)
    assert lib.sp_host_index_digest(b"VTML80.out", r["sens"], 1, buf, offs, len(T), C.byref(ent), C.byref(msk), C.byref(dig), masked) == 0
    assert ent.value == r["index"]["entries"]
    assert msk.value == r["maskedResidues"]
    assert "%016x" % dig.value == r["index"]["digest"]
    assert masked.raw[:len(buf)].decode() == "".join(r["masked"])


def test_host_evalues(lib):
    r = load_golden("spacer")["ref"]
    lib.sp_host_evalue.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for key, v in r["evalueProbe"].items():
        ql, s = (int(x) for x in key.split("_"))
        ev, bits, evg, bitsg = (float.fromhex(x) for x in v)
        e, b = C.c_double(), C.c_double()
        assert lib.sp_host_evalue(b"VTML40.out", 16, 2, r["aaDbSize"], float(s), float(ql), C.byref(e), C.byref(b)) == 0
        assert (e.value, b.value) == (ev, bits)
        assert lib.sp_host_evalue(b"VTML40.out", -1, -1, r["aaDbSize"], float(s), float(ql), C.byref(e), C.byref(b)) == 0
        assert (e.value, b.value) == (evg, bitsg)
    # other matrix / gap pairs come from the table generated from the reference's EvalueComputation (tools/gen_alp_table.py -> csrc/sp_alp_table.h); probes from the
    # same classes: tests/golden/alp_probes.json
    import json
    probes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alp_probes.json")))["cases"]
    assert len(probes) >= 5
    for key, case in probes.items():
        name, go, ge = key.split()
        for s, ql, ev, bits in case["probes"]:
            assert lib.sp_host_evalue(name.encode(), int(go), int(ge), case["db_residues"], float(s), float(ql), C.byref(e), C.byref(b)) == 0, (key, lib.sp_last_error())
            assert (e.value, int(b.value)) == (float.fromhex(ev), bits), (key, s, ql)
    assert lib.sp_host_evalue(b"VTML40.out", 5, 1, 1000, 10.0, 10.0, C.byref(e), C.byref(b)) != 0   # a pair ALP cannot fit (the reference fails on it too): loud failure
    assert lib.sp_host_evalue(b"VTML40.out", 30, 7, 1000, 10.0, 10.0, C.byref(e), C.byref(b)) != 0  # outside the table
    # a file that carries a bundled matrix's NAME but other scores is not given that matrix's parameters
    from oracle_lib import matrix_text
    other = matrix_text("VTML40").replace("A\t6\t", "A\t9\t", 1)      # the A-A score
    assert other != matrix_text("VTML40")
    assert lib.sp_host_evalue(("VTML40.out:" + matrix_text("VTML40")).encode(), 11, 1, 1000, 10.0, 10.0, C.byref(e), C.byref(b)) == 0, lib.sp_last_error()      # (the same table spelled out: accepted)
    assert lib.sp_host_evalue(("VTML40.out:" + other).encode(), 11, 1, 1000, 10.0, 10.0, C.byref(e), C.byref(b)) != 0 and b"other scores" in lib.sp_last_error()


def test_bench_reads_the_committed_counter_summary():
    """bench.py takes roofline.traffic from the newest profiles/rNN_pmc_traffic.json: the file must parse and carry the
    calibration the DESIGN describes (FETCH_SIZE counts half of a streaming read on gfx950)"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    traffic, src = bench.pmc_traffic("k_match")
    assert traffic and src.startswith("profiles/r")
    doc = json.load(open(os.path.join(ROOT, src)))
    cal = doc["calibration_factor_known_over_counted"]
    assert 1.9 < cal["4B_per_lane"] < 2.1 and 1.9 < cal["16B_per_lane"] < 2.1
    k = doc["kernels"]["k_match"]
    assert abs(traffic - (k["fetch_bytes_corrected"] + k["write_bytes_raw"])) < 1.0
    assert 0.5e9 < traffic < 20e9


def test_front_end_does_not_link_the_library_or_the_hip_runtime():
    """the front end loads libspacepharer_amd.so when it runs a module ITSELF: a client that forwards its call to the resident server must not map the HIP runtime first
    (30 ms per process start against 2; csrc/sp_main.cpp, DESIGN.md section 6 / DESIGN_HISTORY.md section 6c)"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spacepharer_amd", "spacepharer-amd")
    if not os.path.exists(exe):
        pytest.skip("front end not built")
    libs = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libamdhip64" not in libs and "libspacepharer_amd" not in libs and "libhsa-runtime" not in libs, libs
    p = subprocess.run([exe, "no_such_module"], capture_output=True, text=True)       # (answers without loading anything)
    assert p.returncode == 1 and "unknown module" in p.stderr


def test_flat_db_reader(lib, tmp_path):
    """sp_flat_db_open: a sequence DB triple (split data files, keys that are not 0..n-1, a lookup with the set column) comes back as residues / offsets / keys / sets
    in the order of the data file (the reference's DBReader in its LINEAR_ACCCESS sort mode) -- what sp_ctx_set_targets / sp_ctx_set_queries and sp_set_layout take"""
    import struct
    import spacepharer_amd as sa
    seqs = {7: "MKTAYIAKQR", 3: "ACDEFGHIK", 12: "WWWW", 5: "LLLLLLLLLLLLLLLLLLLL"}
    order = [7, 3, 12, 5]                                  # data order differs from key order
    path = str(tmp_path / "db")
    parts, idx, off = [b"", b""], {}, 0
    for i, k in enumerate(order):
        e = seqs[k].encode() + b"\n\0"
        parts[0 if i < 2 else 1] += e
        idx[k] = (off, len(e))
        off += len(e)
    open(path + ".0", "wb").write(parts[0])
    open(path + ".1", "wb").write(parts[1])
    open(path + ".index", "w").write("".join("%d\t%d\t%d\n" % (k, idx[k][0], idx[k][1]) for k in sorted(seqs)))
    open(path + ".dbtype", "wb").write(struct.pack("<i", 0))
    open(path + ".lookup", "w").write("".join("%d\tname%d\t%d\n" % (k, k, 100 + k % 3) for k in sorted(seqs)))
    f = sa.FlatDb(path)
    keys = order
    assert f.n == 4 and [f.keys[i] for i in range(4)] == keys
    raw = C.string_at(C.cast(f.residues, C.c_void_p).value, int(f.offsets[4]))
    assert [raw[int(f.offsets[i]):int(f.offsets[i + 1])].decode() for i in range(4)] == [seqs[k] for k in keys]
    assert [f.sets[i] for i in range(4)] == [100 + k % 3 for k in keys]
    f.close()
    with pytest.raises(sa.SpError):
        sa.FlatDb(str(tmp_path / "missing"))
    os.remove(path + ".lookup")
    g = sa.FlatDb(path)
    assert g.n == 4 and not g.sets
    g.close()
