"""CPU: host-side logic of the engine (partition arithmetic, synthetic generator,
workload definitions, the bit-vector DP core compiled for the host)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H


def test_shard_range_partitions_exactly():
    apm = H.pkg()
    for n, k, g in [(0, 0, 1), (5, 7, 3), (1000, 3, 3), (1 << 20, 0, 8), (12345677, 5, 7), (1 << 33, 2, 8), (17, 0, 8)]:
        prev = 0
        for s in range(g):
            b, e = apm.shard_range(n, k, s, g)
            assert b == prev and e >= b
            if 0 < s:
                assert b % 16 == 0 or b == max(0, n - k)
            prev = e
        assert prev == max(0, n - k)


def test_shard_range_rejects_bad_arguments():
    apm = H.pkg()
    for args in [(10, 0, 3, 3), (10, 0, -1, 3), (10, -1, 0, 1), (10, 0, 0, 0)]:
        with pytest.raises(apm.ApmError):
            apm.shard_range(*args)


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    return x ^ (x >> 31)


def test_synth_generator_matches_spec():
    """byte i = "ACGT"[(splitmix64(seed ^ (i>>5)) >> (2*(i&31))) & 3]  """
    apm = H.pkg()
    seed = 0x5EED0002
    for off, ln in [(0, 100), (31, 70), (123456789, 65), ((1 << 33) - 40, 40)]:
        got = apm.synth_fill_host(off, ln, seed)
        want = bytes(b"ACGT"[(_splitmix64(seed ^ (i >> 5)) >> (2 * (i & 31))) & 3] for i in range(off, off + ln))
        assert got == want
    big = np.frombuffer(apm.synth_fill_host(0, 1 << 16, seed), dtype=np.uint8)
    freq = np.bincount(big, minlength=256)[[65, 67, 71, 84]] / big.size
    assert freq.sum() == 1.0 and abs(freq - 0.25).max() < 0.02


def test_workload_definitions_match_baseline():
    wl = H.workloads()
    c = wl.CONFIGS
    assert c["cfg2"]["n"] == 1 << 28 and c["cfg2"]["lens"] == [32] * 8 and c["cfg2"]["k"] == 0
    assert c["cfg3"]["n"] == 1 << 30 and len(c["cfg3"]["lens"]) == 32 and c["cfg3"]["k"] == 3
    assert min(c["cfg3"]["lens"]) == 16 and max(c["cfg3"]["lens"]) == 128
    assert sum(m * m for m in c["cfg3"]["lens"]) == 201456
    assert c["cfg4"]["lens"] == [64] * 16 and c["cfg4"]["k"] == 2 and c["cfg4"]["n"] == 1 << 33
    assert c["cfg5"]["lens"] == [50] * 256 and c["cfg5"]["k"] == 5
    assert abs(wl.algorithmic_cells(1 << 28, [32] * 8, 0) - 2.199e12) / 2.199e12 < 1e-3


def test_planted_patterns_are_found_by_the_oracle():
    wl = H.workloads()
    apm = H.pkg()
    n, k = 1 << 14, 3
    lens = [16, 32, 50, 64, 100, 128, 20, 40]
    seed = wl.seed_of(3)
    pats, planted = wl.make_patterns(n, lens, k, seed)
    text = apm.synth_fill_host(0, n, seed)
    counts = H.oracle_counts(text, pats, k, banded=True)
    for (o, d), cnt, p in zip(planted, counts, pats):
        assert H.window_distance(p, text[o:o + len(p)]) == d
        if d <= k:
            assert cnt >= 1


def test_bitvector_core_on_host(tmp_path):
    """apm_core.h's Myers/Hyyro column == the oracle's window distance (compiled with g++)."""
    src = os.path.join(H.ROOT, "tests", "host_core_test.cpp")
    exe = str(tmp_path / "host_core_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(H.PKG_DIR, "csrc"), "-I", os.path.join(H.ROOT, "oracle"),
                    src, os.path.join(H.ROOT, "oracle", "apm_oracle.c"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_no_kernel_spills_to_scratch():
    """csrc/apm_kernels.resources.txt (written by the Makefile from the compiler's resource-usage
    remarks): a spill reload inside a streaming loop is a vector-memory op that drains every prefetch
    queued behind it (vmcnt counts in order) -- it cost 10 % on the headline kernel once."""
    import glob
    paths = sorted(glob.glob(os.path.join(H.PKG_DIR, "csrc", "*.resources.txt")))
    if len(paths) < 2:
        pytest.skip("library was not built by the Makefile in this checkout")
    name, seen = None, 0
    for line in (l for p in paths for l in open(p)):
        if line.startswith("Function Name:"):
            name = line.split(":", 1)[1].strip()
        elif line.startswith("ScratchSize"):
            seen += 1
            spilled = int(line.rsplit(":", 1)[1])
            # No kernel may spill (round 3: the two LDS-tile instantiations that kept 8 bytes per lane gave up a wave of
            # occupancy instead -- they are the fallback for unaligned text only).
            assert spilled == 0, "%s spills %d bytes to scratch" % (name, spilled)
        elif line.startswith("LDS Size") and any(t in name for t in ("apm_filter_kernel", "apm_stream_kernel", "apm_sieve2_kernel", "apm_sieve2cf_kernel",
                                                                      "apm_sieve8_kernel", "apm_fused_kernel")):
            # their key bitmap is addressed as a compile-time LDS constant: dynamic LDS must start at 0
            assert int(line.rsplit(":", 1)[1]) == 0, "%s owns static LDS" % name
    assert seen >= 40
