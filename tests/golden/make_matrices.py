#!/usr/bin/env python3
"""One-off: read the published VTML40/VTML80 half-bit tables (+ background, lambda) from the reference's data
directory and store them as plain numbers in tests/golden/matrices.json.  Both the oracle tests (which render
the numbers back into `.out` text) and tools/gen_matrix_inc.py (which emits the product's embedded C tables)
read that JSON, so the matrix numbers exist once in the repo."""
import json, os, sys
REF = os.environ.get("SP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for name in ("VTML40", "VTML80"):
    bg = lam = None; letters = None; rows = {}
    for line in open(os.path.join(REF, "lib/mmseqs/data", name + ".out")):
        if line.startswith("# Background (precomputed optional):"):
            bg = line.split()[4:]
        elif line.startswith("# Lambda     (precomputed optional):"):
            lam = line.split()[4]
        elif line.startswith("#") or not line.strip():
            continue
        elif letters is None:
            letters = line.split()
        else:
            w = line.split(); rows[w[0]] = [int(x) for x in w[1:]]
    out[name] = {"letters": "".join(letters), "background": bg, "lambda": lam,
                 "scores": [rows[l] for l in letters]}
json.dump(out, open(os.path.join(HERE, "matrices.json"), "w"), separators=(",", ":"))
print("wrote matrices.json")
