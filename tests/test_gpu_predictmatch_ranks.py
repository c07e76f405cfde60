"""GPU: BASELINE configs[4]'s chain on N ranks (tools/predictmatch_ranks.py: queries partitioned by CRISPR array, one process per rank, the library's communicator
for the one exchange) gives the one-rank run's match list, byte for byte."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(ranks, tmp):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "predictmatch_ranks.py"), "--ranks", str(ranks), "--same-gpu", "--dir", str(tmp), "60", "12000", "3", "400", "12"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_predictmatch_chain_on_three_ranks_equals_one_rank(tmp_path):
    one = _run(1, tmp_path / "one")
    three = _run(3, tmp_path / "three")
    assert one["ranks"] == 1 and three["ranks"] == 3
    assert one["matches_reported"] > 100 and one["true_matches_found"] > 0.8 * one["matches_reported"]
    for k in ("query_peptides", "alignments_target", "alignments_control", "matches_reported", "true_matches_found", "match_db_crc"):
        assert one[k] == three[k], (k, one[k], three[k])
    assert open(tmp_path / "one" / "match", "rb").read() == open(tmp_path / "three" / "match", "rb").read()
