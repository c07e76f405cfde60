"""Malformed input to the host-side DB layer and the set-level modules (sp_db.h, sp_sets.cpp, sp_modules.cpp): truncated / garbled `.index` files, wrong `.dbtype`,
data files shorter than their index says, result entries with missing / non-numeric / oversized columns, missing terminators.  A module may succeed or fail with a
message -- it must never crash (signal), hang, or read outside its buffers.  The same cases run under AddressSanitizer + UBSan with tools/asan_tests.sh
(`make -C spacepharer_amd/csrc asan`), where any out-of-bounds read aborts the process.

Every case is one run of the `spacepharer-amd` front end in its own process (it starts in milliseconds and needs no device for these modules)."""
import os
import random
import shutil
import struct
import subprocess
import zlib

import pytest

from conftest import load_golden
from test_set_modules import write_db

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONT = os.path.join(ROOT, "spacepharer_amd", "spacepharer-amd")
THREADS = ["--threads", "2", "--compressed", "0", "-v", "1"]


@pytest.fixture(scope="module")
def ex(tmp_path_factory):
    g = load_golden("examples")
    d = tmp_path_factory.mktemp("fuzz")
    for name, db in g["db"].items():
        write_db(str(d / name), db)
    for name, text in list(g["lookup"].items()) + list(g["lookup_x"].items()):
        open(str(d / (name + ".lookup")), "w").write(text)
    return g, d


def run(args, cwd):
    env = dict(os.environ)
    p = subprocess.run([FRONT] + [str(a) for a in args], cwd=str(cwd), capture_output=True, timeout=120, env=env)
    assert p.returncode >= 0 and p.returncode not in (134, 139), (args, p.returncode, p.stderr[-2000:].decode(errors="replace"))      # no signal, no abort (ASan / UBSan abort)
    assert b"AddressSanitizer" not in p.stderr and b"runtime error:" not in p.stderr, p.stderr[-3000:].decode(errors="replace")
    return p.returncode


# module -> (input DBs in argv order, number of output DBs, extra flags)
MODULES = {
    "extractorfs": (["q_nucl"], 1, ["--min-length", "9", "--orf-start-mode", "1", "--create-lookup", "1"]),
    "translatenucs": (["q_nucl_orf"], 1, []),
    "reverseseqbycodon": (["t_nucl_orf"], 1, []),
    "proteinaln2nucl": (["q_nucl_orf", "t_nucl_orf", "q", "t", "prot_result"], 1, []),
    "combineprotnuclaln": (["prot_result", "nucl_result"], 1, []),
    "besthitperset": (["q", "t", "result"], 1, ["--simple-best-hit", "1"]),
    "mergeresultsbyset": (["q_set_to_member", "aggregate"], 1, []),
    "combinescore": (["q", "t", "aggregate_merged"], 1, ["."]),      # (the command line also names a tmpDir behind the output DB)
    "filtermatchbyfdr": (["cScore", "cScore_rev"], 1, ["--fdr", "0.05"]),
    "summarizeresults": (["t_nucl", "match", "aln"], 1, ["--fmt", "1"]),
    # the device modules read, flatten and -- align, rescorediagonal -- parse their DBs BEFORE they ask for the device: without a GPU they end with "no ROCm-capable device"
    # after the code under test has run (with one, they search)
    "prefilter": (["q", "t"], 1, ["--max-seqs", "300", "-s", "5.7"]),
    "align": (["q", "t", "prot_result_pref"], 1, ["-a", "1", "--alignment-mode", "2", "-e", "200", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]),
    "rescorediagonal": (["q", "t", "prot_result_pref"], 1, ["--rescore-mode", "2", "-a", "1", "-e", "200"]),
    "search": (["q", "t"], 1, ["tmp", "--max-seqs", "300", "-s", "5.7", "-a", "1", "--alignment-mode", "2", "-e", "200"]),
}
DEVICE_MODULES = ("prefilter", "align", "rescorediagonal", "search")


def stage(src_dir, dst_dir, private):
    """every file of the examples directory as a link, the files of DB `private` (the one about to be damaged) as copies"""
    for f in os.listdir(str(src_dir)):
        src, dst = os.path.join(str(src_dir), f), os.path.join(str(dst_dir), f)
        if private is not None and (f == private or f.startswith(private + ".")):
            shutil.copy(src, dst)
        else:
            os.symlink(src, dst)


def mutate_index(path, rng, kind):
    lines = open(path).read().splitlines(True)
    if kind == "truncate":
        text = "".join(lines)
        open(path, "w").write(text[:rng.randrange(1, max(2, len(text)))])
    elif kind == "empty":
        open(path, "w").write("")
    elif kind == "garbage":
        i = rng.randrange(len(lines))
        lines[i] = rng.choice(["x\ty\tz\n", "\t\t\n", "12\n", "-5\t-7\t-9\n", "1\t2\n", "99999999999999999999\t1\t1\n", "\0\0\0\n", "5 6 7\n"])
        open(path, "w").write("".join(lines))
    elif kind == "offset":
        i = rng.randrange(len(lines))
        k, o, l = lines[i].split("\t")
        lines[i] = "%s\t%d\t%s" % (k, int(o) + rng.choice([1 << 20, 1 << 40, (1 << 63) - 1, 7]), l)
        open(path, "w").write("".join(lines))
    elif kind == "length":
        i = rng.randrange(len(lines))
        k, o, l = lines[i].split("\t")
        lines[i] = "%s\t%s\t%d\n" % (k, o, rng.choice([0, 1, 1 << 30, (1 << 62), 18446744073709551615]))
        open(path, "w").write("".join(lines))
    elif kind == "no-newline":
        open(path, "w").write("".join(lines).rstrip("\n"))
    elif kind == "duplicate-keys":
        open(path, "w").write("".join(lines + lines[:3]))


def mutate_data(path, rng, kind):
    data = bytearray(open(path, "rb").read())
    if kind == "truncate-data":
        data = data[:rng.randrange(0, max(1, len(data)))]
    elif kind == "no-terminators":
        data = bytearray(b if b else 65 for b in data)
    elif kind == "bytes":
        for _ in range(max(1, len(data) // 200)):
            if data:
                data[rng.randrange(len(data))] = rng.choice([0, 9, 10, 45, 48, 57, 65, 255])
    elif kind == "columns":      # drop or double tab-separated columns of text entries
        text = bytes(data).split(b"\n")
        for _ in range(max(1, len(text) // 20)):
            i = rng.randrange(len(text))
            cols = text[i].split(b"\t")
            if len(cols) > 1:
                j = rng.randrange(len(cols))
                text[i] = b"\t".join(cols[:j] + rng.choice([[], [cols[j], cols[j]], [b"nan"], [b"-1"], [b"1e999"], [b"9" * 40]]) + cols[j + 1:])
        data = bytearray(b"\n".join(text))
    open(path, "wb").write(bytes(data))


INDEX_KINDS = ["truncate", "empty", "garbage", "offset", "length", "no-newline", "duplicate-keys"]
DATA_KINDS = ["truncate-data", "no-terminators", "bytes", "columns"]


@pytest.mark.parametrize("module", sorted(MODULES))
def test_malformed_dbs_never_crash_a_host_module(ex, tmp_path, module):
    g, d = ex
    inputs, n_out, flags = MODULES[module]
    rounds = int(os.environ.get("SP_FUZZ_ROUNDS", "2"))      # (a campaign: SP_FUZZ_ROUNDS=200 SP_FUZZ_SEED=<n>; round 6 ran 300 rounds per module under ASan + UBSan)
    seed0 = zlib.crc32(module.encode()) + int(os.environ.get("SP_FUZZ_SEED", "0"))      # (not hash(): that differs from process to process)
    # the unmodified call first: it must work, so that the mutations below are what fails (or not)
    work = tmp_path / "ok"
    work.mkdir()
    stage(d, work, None)
    rc = run([module] + inputs + ["fuzz_out"] + flags + THREADS, work)
    assert rc == 0 or (module in DEVICE_MODULES and not os.path.exists("/dev/kfd")), rc
    n_cases = 0
    for rnd in range(rounds):
      rng = random.Random(seed0 + rnd)
      for which in range(len(inputs)):
        for kind in INDEX_KINDS + DATA_KINDS + ([] if rnd else ["dbtype", "missing-dbtype", "missing-index"]):
            work = tmp_path / ("%s_%d_%d_%s" % (module, rnd, which, kind))
            work.mkdir()
            stage(d, work, inputs[which])
            target = str(work / inputs[which])
            if kind in INDEX_KINDS:
                mutate_index(target + ".index", rng, kind)
            elif kind in DATA_KINDS:
                mutate_data(target, rng, kind)
            elif kind == "dbtype":
                open(target + ".dbtype", "wb").write(rng.choice([b"", b"\x01", struct.pack("<i", 12345), b"\xff" * 4, b"x" * 9]))
            elif kind == "missing-dbtype":
                os.remove(target + ".dbtype")
            else:
                os.remove(target + ".index")
            run([module] + inputs + ["fuzz_out"] + flags + THREADS, work)
            n_cases += 1
            shutil.rmtree(str(work))
    assert n_cases >= 14
