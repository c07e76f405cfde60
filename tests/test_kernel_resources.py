"""Register / LDS budget of the kernels whose speed hangs on occupancy (cross-compiled for gfx950, no GPU needed).

k_match is latency-bound: its large footprint (dense index) runs three waves per SIMD only while it stays at or below 168 VGPRs and
53 KB of LDS per block of four waves, its small footprint (sparse index) four waves at 128 VGPRs and 40 KB.  A
change that pushes either over the edge costs a third / a quarter of the occupancy and 15-50 % of the kernel time without failing any
parity test -- this test fails instead."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (shutil.which(HIPCC) or os.path.exists(HIPCC)), reason="hipcc not available")
def test_match_and_sw_kernels_keep_their_occupancy(tmp_path):
    src = os.path.join(ROOT, "spacepharer_amd", "csrc", "sp_gpu.hip")
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-c", src,
                          "-o", str(tmp_path / "sp_gpu.o"), "-Rpass-analysis=kernel-resource-usage"],
                         cwd=os.path.dirname(src), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    info, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            info[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
        if m and name:
            info[name][m.group(1).strip()] = int(m.group(2))

    def kernel(fragment):
        hits = [v for k, v in info.items() if fragment in k]
        assert hits, (fragment, list(info)[:5])
        return hits[0]

    match = kernel("k_matchILi512ELb0ELb0")                   # k_match<512, false, false>: the unsharded search on a dense index (large footprint)
    assert match["Occupancy [waves/SIMD]"] >= 3, match
    # (up to 8 spilled registers: four wave-constant pointers parked in the kernel's first instructions and fetched back in a query's tail -- none inside the window loop)
    assert match["VGPRs"] <= 168 and match.get("VGPRs Spill", 0) <= 8, match
    assert match["LDS Size [bytes/block]"] * 3 <= 160 * 1024, match
    small = kernel("k_matchILi256ELb0ELb1")                   # k_match<256, false, true>: sparse index -- four waves per SIMD, or three and room for an aligner
    assert small["Occupancy [waves/SIMD]"] >= 4 and small["VGPRs"] <= 128 and small.get("VGPRs Spill", 0) <= 4, small
    assert small["LDS Size [bytes/block]"] * 4 <= 160 * 1024, small
    assert small["LDS Size [bytes/block]"] * 3 + 2 * 25 * 1024 <= 160 * 1024, small      # three blocks per CU leave two blocks of the forward aligner their LDS
    grp = kernel("k_match_grpILb0ELi0E")                            # short queries, sparse index, unsharded: four queries per pass of a wave -- four blocks per CU,
    assert grp["Occupancy [waves/SIMD]"] >= 4 and grp["VGPRs"] <= 128 and grp.get("VGPRs Spill", 0) == 0, grp      # or three and room for an aligner
    assert grp["LDS Size [bytes/block]"] * 4 <= 160 * 1024, grp
    grpo = kernel("k_match_grpILb1ELi0E")                     # ... and with k-mer ordinals (the target-sharded search, round 5): three blocks per CU, no spills
    assert grpo["Occupancy [waves/SIMD]"] >= 3 and grpo.get("VGPRs Spill", 0) == 0 and grpo["LDS Size [bytes/block]"] * 3 <= 160 * 1024, grpo
    for side in ("k_match_grpILb1ELi1E", "k_match_grpILb1ELi2E"):      # the two sides of the probe-list cut (enumerate / fetch): laid out for four blocks per CU
        pc = kernel(side)
        assert pc["Occupancy [waves/SIMD]"] >= 4 and pc.get("VGPRs Spill", 0) == 0 and pc["LDS Size [bytes/block]"] * 4 <= 160 * 1024, (side, pc)
    dense2 = kernel("k_match_dense")                          # round 6's opt-in kernel for the dense index: three blocks per CU, nothing spilled
    assert dense2["Occupancy [waves/SIMD]"] >= 3 and dense2.get("VGPRs Spill", 0) == 0 and dense2["LDS Size [bytes/block]"] * 3 <= 160 * 1024, dense2
    shard = kernel("k_matchILi256ELb1ELb1")                   # k_match<256, true, true>: the target-sharded search (ordinals recorded): three blocks per CU
    assert shard["Occupancy [waves/SIMD]"] >= 3 and shard["LDS Size [bytes/block]"] * 3 <= 160 * 1024, shard
    for gl in (10, 12, 16, 32):                               # the packed Smith-Waterman kernels: eight waves per SIMD
        sw = kernel("k_sw_byte1pILi%dE" % gl)
        assert sw["Occupancy [waves/SIMD]"] >= 8, (gl, sw)
        fwd = kernel("k_sw_fwd1pILi%dE" % gl)                 # the forward pass: one 128-byte profile line per residue for both halves, 13.6 KB of LDS per block
        assert fwd["Occupancy [waves/SIMD]"] >= (8 if gl <= 16 else 6) and fwd.get("VGPRs Spill", 0) == 0 and fwd["VGPRs"] <= 80, (gl, fwd)
        assert fwd["LDS Size [bytes/block]"] <= 14 * 1024, (gl, fwd)      # (four blocks fit beside two blocks of the dense matching kernel)
