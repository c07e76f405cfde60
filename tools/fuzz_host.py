#!/usr/bin/env python3
"""Fuzz campaign over the host-side modules (the cases of tests/test_host_fuzz.py, more rounds): stops at the first crash and prints the damaged DB.
usage: tools/fuzz_host.py <module|all> [rounds] [seed]        (SP_LIB_NAME / LD_PRELOAD select the sanitized build: tools/asan_tests.sh)"""
import os
import random
import shutil
import struct
import subprocess
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_host_fuzz as F  # noqa: E402
from conftest import load_golden  # noqa: E402
from test_set_modules import write_db  # noqa: E402


def main():
    which_module = sys.argv[1] if len(sys.argv) > 1 else "all"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    g = load_golden("examples")
    d = tempfile.mkdtemp(prefix="sp_fuzz_src")
    for name, db in g["db"].items():
        write_db(os.path.join(d, name), db)
    for name, text in list(g["lookup"].items()) + list(g["lookup_x"].items()):
        open(os.path.join(d, name + ".lookup"), "w").write(text)
    base = tempfile.mkdtemp(prefix="sp_fuzz_work")
    n = 0
    for module in sorted(F.MODULES) if which_module == "all" else [which_module]:
        inputs, _, flags = F.MODULES[module]
        for rnd in range(rounds):
            rng = random.Random(zlib.crc32(module.encode()) + seed + rnd)
            for which in range(len(inputs)):
                for kind in F.INDEX_KINDS + F.DATA_KINDS:
                    work = os.path.join(base, "w")
                    shutil.rmtree(work, ignore_errors=True)
                    os.mkdir(work)
                    F.stage(d, work, inputs[which])
                    target = os.path.join(work, inputs[which])
                    (F.mutate_index if kind in F.INDEX_KINDS else F.mutate_data)(target + (".index" if kind in F.INDEX_KINDS else ""), rng, kind)
                    p = subprocess.run([F.FRONT, module] + inputs + ["fuzz_out"] + flags + F.THREADS, cwd=work, capture_output=True, timeout=120)
                    n += 1
                    if p.returncode < 0 or p.returncode in (134, 139) or b"AddressSanitizer" in p.stderr or b"runtime error:" in p.stderr:
                        print("CRASH %s round %d input %d (%s) kind %s rc %d" % (module, rnd, which, inputs[which], kind, p.returncode))
                        print(p.stderr[-1500:].decode(errors="replace"))
                        print("index:", repr(open(target + ".index").read()[:400]))
                        print("data: ", repr(open(target, "rb").read()[:600]))
                        print("kept in", work)
                        return 1
        print("%s: %d rounds clean" % (module, rounds), flush=True)
    print("%d cases, no crash" % n)
    shutil.rmtree(base, ignore_errors=True)
    shutil.rmtree(d, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
