"""GPU (-m gpu): the HIP path, called THROUGH THE C ABI, against
  (1) the golden vectors produced by the reference binary (tests/golden),
  (2) the oracle restatement on seeded inputs,
  (3) size-independent properties at BASELINE.json's full sizes.
Bit-exact integer counts everywhere (no tolerance)."""
import json
import os
import random
import subprocess

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

VARIANTS = ["auto", "banded", "bitpar", "wavefront", "generic", "nfa"]
MAX_M = {"auto": 65535, "generic": 65535, "bitpar": 4096, "wavefront": 256, "banded": 512, "nfa": 32}


def _supported(variant, m, k):
    """mirror of the library's documented limits (include/apm.h, apm_set_kernel -> UNSUPPORTED); m: a length or the pattern"""
    pat = None
    if isinstance(m, (bytes, bytearray)):
        pat, m = m, len(m)
    if m > MAX_M[variant]:
        return False
    if variant == "banded":
        return k <= 7 and m // (k + 1) >= 4
    if variant == "nfa":
        return k <= 7 and m + k // 2 <= 32 and (pat is None or len(set(pat)) <= 16)
    return True

CASES = H.golden()["cases"]


@pytest.fixture(scope="module")
def apm():
    return H.pkg()


@pytest.fixture(scope="module")
def ctx(apm):
    assert apm.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    c = apm.ApmContext(device=0)
    yield c
    c.close()


def _run(ctx, apm, variant, patterns, k, text):
    ctx.set_kernel("auto")
    ctx.set_patterns(patterns, k)
    ctx.set_kernel(variant)
    return ctx.count_buffer(text)


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden(ctx, apm, case, variant):
    pats, k = case["patterns"], case["k"]
    if not all(_supported(variant, p, k) for p in pats):
        ctx.set_kernel("auto")
        ctx.set_patterns(pats, k)
        with pytest.raises(apm.ApmError) as e:      # error behaviour: UNSUPPORTED, state unchanged
            ctx.set_kernel(variant)
        assert e.value.status == -6
        assert ctx.count_buffer(H.case_text(case)) == case["counts"]
        return
    if variant == "generic" and H.case_cells(case) > 8e9:
        pytest.skip("literal global-memory DP: keep the GPU suite short")
    assert _run(ctx, apm, variant, pats, k, H.case_text(case)) == case["counts"]


def test_count_file_equals_golden(ctx):
    for name in ("cfg1_basic_test", "x100_k3", "chrY_k5", "bigger_k0", "run_tests_easy"):
        c = next(c for c in CASES if c["name"] == name)
        ctx.set_kernel("auto")
        ctx.set_patterns(c["patterns"], c["k"])
        assert ctx.count_file(c["path"]) == c["counts"]


# ---------------------------------------------------------------- oracle on seeded inputs
@pytest.mark.parametrize("variant", VARIANTS)
def test_random_vs_oracle(ctx, apm, variant):
    rnd = random.Random(1234)
    for trial in range(40):
        alpha = rnd.choice([b"ab", b"ACGT", b"ACGT\n", bytes(range(256))])
        n = rnd.choice([0, 1, 2, 15, 16, 17, 255, 256, 257, 1023, 1024, 1025, 1500, 4095, 4097, 9000])
        text = bytes(rnd.choice(alpha) for _ in range(n))
        pats = []
        for _ in range(rnd.randint(1, 7)):
            m = rnd.choice([1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 95, 96, 97, 127, 128])
            if variant in ("auto", "generic", "wavefront") and rnd.random() < 0.15:
                m = rnd.choice([129, 150, 200, 256])
            if variant in ("auto", "generic") and rnd.random() < 0.1:
                m = rnd.choice([257, 300, 700])
            if n > m and rnd.random() < 0.7:
                o = rnd.randrange(0, n - m + 1)
                p = bytearray(text[o:o + m])
                for _e in range(rnd.randint(0, 4)):
                    r = rnd.random()
                    pos = rnd.randrange(len(p))
                    if r < 0.5:
                        p[pos] = rnd.choice(alpha)
                    elif r < 0.75 and len(p) > 1:
                        del p[pos]
                        p.append(rnd.choice(alpha))
                    else:
                        p.insert(pos, rnd.choice(alpha))
                        p.pop()
                p = bytes(p)
            else:
                p = bytes(rnd.choice(alpha) for _ in range(m))
            pats.append(p)
        k = rnd.choice([0, 0, 1, 2, 3, 4, 5, 6, 9])
        pats = [p for p in pats if _supported(variant, p, k)]
        if not pats:
            continue
        want = H.oracle_counts(text, pats, k)
        got = _run(ctx, apm, variant, pats, k, text)
        assert got == want, (trial, n, k, [len(p) for p in pats])


@pytest.mark.parametrize("variant", ["auto", "banded", "bitpar", "wavefront"])
def test_tile_boundaries(ctx, apm, variant):
    """a planted occurrence straddling every tile seam is found exactly once"""
    rnd = random.Random(99)
    n = 9000
    base = bytearray(rnd.choice(b"ACGT") for _ in range(n))
    pat = bytes(rnd.choice(b"ACGT") for _ in range(40))
    for seam in (512, 1024, 2048, 3072, 4000, 4040, 4064, 4080, 4096):
        for off in (-39, -20, -1, 0, 1):
            text = bytearray(base)
            o = seam + off
            text[o:o + 40] = pat
            text = bytes(text)
            for k in (0, 2):
                want = H.oracle_counts(text, [pat], k)
                assert want[0] >= 1
                assert _run(ctx, apm, variant, [pat], k, text) == want


def test_duplicate_and_many_patterns(ctx, apm):
    """P = 300 (several LDS batches) incl. duplicates -> counts[] order is argv order"""
    rnd = random.Random(5)
    text = bytes(rnd.choice(b"ACGT") for _ in range(6000))
    pats = []
    for i in range(300):
        m = rnd.choice([8, 20, 50])
        o = rnd.randrange(0, len(text) - m)
        pats.append(text[o:o + m])
    pats[17] = pats[3]
    want = H.oracle_counts(text, pats, 1, banded=True)
    assert _run(ctx, apm, "auto", pats, 1, text) == want
    sub = [i for i, p in enumerate(pats) if _supported("banded", len(p), 1)]
    assert _run(ctx, apm, "banded", [pats[i] for i in sub], 1, text) == [want[i] for i in sub]
    assert _run(ctx, apm, "bitpar", pats, 1, text) == want
    assert _run(ctx, apm, "wavefront", pats, 1, text) == want


# ---------------------------------------------------------------- shards (device-resident API)
def _device_shard_counts(ctx, text, pats, k, n_shards, front_pad=0):
    """owner-computes partition through apm_count_shard_device; returns the summed counts"""
    n = len(text)
    P = len(pats)
    apm = H.pkg()
    m_max = max(len(p) for p in pats)
    d_counts = ctx.device_alloc(8 * P)
    ctx.device_memset(d_counts, 0, 8 * P)
    for s in range(n_shards):
        b, e = apm.shard_range(n, k, s, n_shards)
        if e <= b:
            continue
        lo = max(0, b - front_pad)
        hi = min(n, e + m_max - 1)
        d_text = ctx.device_alloc(hi - lo + 16)
        ctx.device_upload(d_text, text[lo:hi])
        ctx.count_shard_device(d_text, lo, hi - lo, n, b, e, d_counts)
        ctx.synchronize()
        ctx.device_free(d_text)
    raw = ctx.device_download(d_counts, 8 * P)
    ctx.device_free(d_counts)
    return [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(P)]


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_sharded_equals_unsharded(ctx, n_shards):
    """DB_OVER_RANKS done right: no double count at shard seams (the reference's own
    database_over_ranks.c:339-343 over-counts there: 6 vs 3 on this very input)."""
    ctx.set_kernel("auto")
    text, pats = b"A" * 16, [b"AAAB", b"AAAA"]
    ctx.set_patterns(pats, 0)
    assert _device_shard_counts(ctx, text, pats, 0, min(n_shards, 1)) == [3, 16]
    for name in ("chrY_k3", "x100_k2", "dna20k_k5"):
        c = next(c for c in CASES if c["name"] == name)
        text = H.case_text(c)
        for variant in ("auto", "wavefront", "bitpar"):
            ctx.set_kernel("auto")
            ctx.set_patterns(c["patterns"], c["k"])
            ctx.set_kernel(variant)
            assert _device_shard_counts(ctx, text, c["patterns"], c["k"], n_shards) == c["counts"]
            # text slice starting before own_begin at a non-16-byte-aligned address
            assert _device_shard_counts(ctx, text, c["patterns"], c["k"], n_shards, front_pad=5) == c["counts"]


def test_shard_halo_is_enforced(ctx, apm):
    ctx.set_kernel("auto")
    ctx.set_patterns([b"ACGTACGTAC"], 0)
    d_text = ctx.device_alloc(64)
    d_counts = ctx.device_alloc(8)
    with pytest.raises(apm.ApmError) as e:       # own range [0,40) of a 100-byte text needs 49 bytes
        ctx.count_shard_device(d_text, 0, 40, 100, 0, 40, d_counts)
    assert e.value.status == -1
    ctx.device_free(d_text)
    ctx.device_free(d_counts)


def test_argument_errors(ctx, apm):
    with pytest.raises(apm.ApmError):
        ctx.set_patterns([b"ACGT"], -1)
    with pytest.raises(apm.ApmError):
        ctx.set_patterns([b"ACGT", b""], 0)
    with pytest.raises(apm.ApmError):
        ctx.set_patterns([], 0)
    with pytest.raises(apm.ApmError):
        ctx.set_kernel(17)
    ctx.set_patterns([b"ACGT"], 0)
    with pytest.raises(apm.ApmError) as e:
        ctx.count_file("/nonexistent/file.fa")
    assert e.value.status == -4


# ---------------------------------------------------------------- synthetic generator + workloads
def test_device_generator_equals_host_generator(ctx, apm):
    seed = 0x5EED0002
    for off, ln in [(0, 4096), (48, 1000), (123456789, 777), ((1 << 33) - 4096, 4096)]:
        d = ctx.device_alloc(ln + 16)
        ctx.synth_fill_device(d, off, ln, seed)
        ctx.synchronize()
        assert ctx.device_download(d, ln) == apm.synth_fill_host(off, ln, seed)
        ctx.device_free(d)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_baseline_workloads_small_vs_oracle(ctx, apm, cfg):
    """each BASELINE config's pattern set / k on a 1 MiB synthetic text: every kernel variant
    == oracle (banded form, itself pinned to the reference on all golden cases)."""
    wl = H.workloads()
    c = wl.CONFIGS[cfg]
    n = 1 << 20
    seed = wl.seed_of(c["cid"])
    lens = c["lens"] if cfg != "cfg5" else c["lens"][:48]
    pats, planted = wl.make_patterns(n, lens, c["k"], seed)
    text = apm.synth_fill_host(0, n, seed)
    want = H.oracle_counts(text, pats, c["k"], banded=True)
    for (o, d), w in zip(planted, want):
        assert (w >= 1) or d > c["k"]
    for variant in ("auto", "banded", "bitpar", "wavefront"):
        sub = [i for i, p in enumerate(pats) if _supported(variant, p, c["k"])]
        assert sub
        ctx.set_kernel("auto")
        ctx.set_patterns([pats[i] for i in sub], c["k"])
        ctx.set_kernel(variant)
        assert ctx.count_synthetic(n, seed) == [want[i] for i in sub], variant


def test_cfg2_full_size_properties(ctx, apm):
    """BASELINE cfg2 at FULL size (256 MiB, 8x32, k=0): planted exact copies are found,
    kernel variants agree bit for bit, and an 8-way shard sum equals the whole."""
    wl = H.workloads()
    c = wl.CONFIGS["cfg2"]
    n, k, seed = c["n"], c["k"], wl.seed_of(c["cid"])
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    ctx.set_kernel("auto")
    ctx.set_patterns(pats, k)
    auto = ctx.count_synthetic(n, seed)
    # planted copy (a random 32-mer recurs with p ~ 2^-36) + the reference's truncated tail
    # windows at the very end of the text (a 1-byte window matches with p = 1/4, ...)
    assert auto == wl.expected_counts_k0(n, pats, planted, seed)
    assert [c >= (1 if d == 0 else 0) for c, (o, d) in zip(auto, planted)] == [True] * len(pats)
    ctx.set_kernel("bitpar")
    assert ctx.count_synthetic(n, seed) == auto
    # CPU slice check: first 256 KiB by the oracle (literal DP)
    sl = 1 << 18
    text = apm.synth_fill_host(0, sl + 31, seed)
    want = H.oracle_counts(text, pats, k, j_end=sl)
    ctx.set_kernel("auto")
    d_text = ctx.device_alloc(len(text) + 16)
    d_counts = ctx.device_alloc(64)
    ctx.device_upload(d_text, text)
    ctx.device_memset(d_counts, 0, 64)
    ctx.count_shard_device(d_text, 0, len(text), n, 0, sl, d_counts)
    ctx.synchronize()
    raw = ctx.device_download(d_counts, 64)
    assert [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(8)] == want
    ctx.device_free(d_text)
    ctx.device_free(d_counts)


def test_cfg3_4mib_vs_oracle(ctx, apm):
    """cfg3's pattern set (32 patterns, m = 16..128, k = 3) on 4 MiB against the banded oracle."""
    wl = H.workloads()
    c = wl.CONFIGS["cfg3"]
    n, k, seed = 1 << 22, c["k"], wl.seed_of(c["cid"])
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    text = apm.synth_fill_host(0, n, seed)
    want = H.oracle_counts(text, pats, k, banded=True)
    ctx.set_kernel("auto")
    ctx.set_patterns(pats, k)
    assert ctx.count_synthetic(n, seed) == want
    assert ctx.count_buffer(text) == want


@pytest.mark.parametrize("cfg,idx", [("cfg3", [0, 1, 3, 4, 8, 12, 20, 31]), ("cfg5", [0, 17, 101, 255]), ("cfg4", [0, 7, 15])])
def test_baseline_sets_64mib_subset_vs_oracle(ctx, apm, cfg, idx):
    """BASELINE pattern sets on 64 MiB of the synthetic text: the FULL set runs on the GPU (its real launch
    structure: sieve pass, verify-only launches, dense tile class ...), a subset of the patterns is checked
    against the multi-threaded banded oracle (the whole set would take the CPU minutes)."""
    wl = H.workloads()
    c = wl.CONFIGS[cfg]
    n, k, seed = 1 << 26, c["k"], wl.seed_of(c["cid"])
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    ctx.set_kernel("auto")
    ctx.set_patterns(pats, k)
    got = ctx.count_synthetic(n, seed)
    text = apm.synth_fill_host(0, n, seed)
    want = H.oracle_counts(text, [pats[i] for i in idx], k, banded=True)
    assert [got[i] for i in idx] == want
    for i, (o, d) in enumerate(planted):
        if d <= k:
            assert got[i] >= 1


def _counts_from(raw, P):
    return [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(P)]


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_the_real_8gib_workloads_range_by_range(apm, cfg):
    """BASELINE.json configs[3..4] as they are: ONE text of 2^33 bytes, the pattern set made for THAT text
    (make_patterns(n = 2^33): occurrences planted all over the 8 GiB), cut by apm_shard_range into the eight owner ranges of
    an 8-GPU node.  The 8 GiB text is resident on this one GPU; every range is scanned as its rank would scan it (its own text
    window with the m_max - 1 halo, text_off = the range's global offset, all beyond 4 GiB for the upper half) by AUTO and
    by the full-DP BITPAR kernel (cfg5: every 8th pattern), pattern by pattern equal; every planted occurrence within k is
    found by the range that owns its position; and the sum over the ranges equals ONE scan of the whole resident text
    (which the pipeline cuts into 3 GiB pieces itself).  Checker at this size: BITPAR + the sum property; the CPU oracle
    anchors the same pattern set at the sizes it can reach (two windows per range here)."""
    import torch
    wl = H.workloads()
    c = wl.CONFIGS[cfg]
    n, k, seed = c["n"], c["k"], wl.seed_of(c["cid"])
    assert n == 1 << 33
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    P = len(pats)
    m_max = max(len(p) for p in pats)
    sub = list(range(P)) if P <= 32 else list(range(0, P, 8))       # patterns the full-DP kernel re-evaluates
    text = torch.empty(n + 4096, dtype=torch.uint8, device="cuda:0")
    cnt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    cnt_sub = torch.zeros(len(sub), dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c2, apm.ApmContext(device=0) as c3:
        c2.set_patterns(pats, k)
        c3.set_patterns([pats[i] for i in sub], k)
        c3.set_kernel("bitpar")
        c2.synth_fill_device(text.data_ptr(), 0, n, seed)
        c2.synchronize()
        total = [0] * P
        for r in range(8):
            ob, oe = apm.shard_range(n, k, r, 8)
            lo, hi = ob, min(n, oe + m_max - 1)
            assert oe > ob and (r == 0 or ob % 16 == 0)
            cnt.zero_(); cnt_sub.zero_()
            torch.cuda.synchronize()
            c2.count_shard_device(text.data_ptr() + lo, lo, hi - lo, n, ob, oe, cnt.data_ptr())
            c3.count_shard_device(text.data_ptr() + lo, lo, hi - lo, n, ob, oe, cnt_sub.data_ptr())
            c2.synchronize(); c3.synchronize()
            got, full = cnt.cpu().tolist(), cnt_sub.cpu().tolist()
            assert [got[i] for i in sub] == full, (cfg, r)
            for i, (o, d) in enumerate(planted):                      # the planted copy lies in exactly one owner range
                if d <= k and ob <= o < oe:
                    assert got[i] >= 1, (cfg, r, i)
            total = [a + b for a, b in zip(total, got)]
            # a window of the range against the CPU oracle (as the rank would see it: global offsets >= 4 GiB from r = 4 on)
            w0, wlen = ob + ((oe - ob) // 2 & ~15) + (5 if r % 2 else 0), 1 << 20
            host = apm.synth_fill_host(w0, wlen + m_max - 1, seed)
            want = H.oracle_counts(host, [pats[i] for i in sub], k, banded=True, j_end=wlen)
            cnt.zero_()
            torch.cuda.synchronize()
            c2.count_shard_device(text.data_ptr() + w0, w0, wlen + m_max - 1, n, w0, w0 + wlen, cnt.data_ptr())
            c2.synchronize()
            one = cnt.cpu().tolist()
            assert [one[i] for i in sub] == want, (cfg, r, w0)
        for cc, (o, d) in zip(total, planted):
            assert cc >= (1 if d <= k else 0)
        cnt.zero_()
        torch.cuda.synchronize()
        c2.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr())   # the whole text in one call
        c2.synchronize()
        assert cnt.cpu().tolist() == total, cfg
    del text


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_full_size_counts_pinned_by_full_dp_kernel(ctx, apm, cfg):
    """BASELINE cfg3 (1 GiB) and the per-GPU shards of cfg4 / cfg5 (2^33 / 8 = 1 GiB) at FULL size: the counts of
    AUTO (the exact-shortcut path: sieve / verify / tile kernels) must equal, pattern by pattern, those of the forced
    BITPAR kernel, which evaluates every DP cell of every window (no filter, no candidate list), on the same
    device-generated text; planted occurrences must be found; and a 4 MiB window in the middle of the text is
    checked against the CPU oracle through apm_count_shard_device."""
    import torch
    wl = H.workloads()
    c = wl.CONFIGS[cfg]
    n, k, seed = 1 << 30, c["k"], wl.seed_of(c["cid"])
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    P = len(pats)
    m_max = max(len(p) for p in pats)
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    cnt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c2:
        c2.set_patterns(pats, k)
        c2.synth_fill_device(text.data_ptr(), 0, n, seed)
        c2.synchronize()
        got = {}
        for variant in ("auto", "bitpar"):
            c2.set_kernel(variant)
            cnt.zero_()
            torch.cuda.synchronize()
            c2.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr())
            c2.synchronize()
            got[variant] = cnt.cpu().tolist()
        assert got["auto"] == got["bitpar"], cfg
        for cc, (o, d) in zip(got["auto"], planted):
            assert cc >= (1 if d <= k else 0)
        # two mid-text windows against the oracle (banded form, pinned to the reference by the goldens): one at a
        # 16-byte aligned device address (the sieve / stream kernels), one unaligned (register-staged tile kernel)
        idx = list(range(P)) if P <= 32 else list(range(0, P, 8))
        c2.set_kernel("auto")
        for w0, wl_len in ((n // 2 - (1 << 21), 1 << 22), (n // 4 + 5, 1 << 21)):
            host = apm.synth_fill_host(w0, wl_len + m_max - 1, seed)
            want = H.oracle_counts(host, [pats[i] for i in idx], k, banded=True, j_end=wl_len)
            cnt.zero_()
            torch.cuda.synchronize()
            c2.count_shard_device(text.data_ptr() + w0, w0, wl_len + m_max - 1, n, w0, w0 + wl_len, cnt.data_ptr())
            c2.synchronize()
            sub = cnt.cpu().tolist()
            assert [sub[i] for i in idx] == want, (cfg, w0)
    del text


def test_auto_routes_long_loose_patterns_to_bitpar(ctx, apm):
    """m <= 512 where BANDED does not apply (k > 7 or pieces shorter than 4 bytes): the bit-vector kernel with columns of
    up to 16 words, not the global-memory GENERIC one (nor the 14 x slower wavefront kernel, which AUTO no longer picks);
    longer patterns too, up to 4096 bytes (24 / 32-word columns, then one window per wave); beyond that, and for long
    patterns over big alphabets, GENERIC; counts = the reference's (golden)."""
    for name in ("chrY_loose_long_k60", "chrY_loose_long_k8", "dna20k_loose_long_k9"):
        c = next(c for c in CASES if c["name"] == name)
        ctx.set_kernel("auto")
        ctx.set_patterns(c["patterns"], c["k"])
        for i, p in enumerate(c["patterns"]):
            assert ctx.pattern_kernel(i) == 3, (name, len(p))
        assert ctx.count_buffer(H.case_text(c)) == c["counts"]
    ctx.set_patterns([b"A" * 300], 100)
    assert ctx.pattern_kernel(0) == 3
    for m in (600, 1024, 1025, 2048, 4096):
        ctx.set_patterns([b"A" * m], 100)
        assert ctx.pattern_kernel(0) == 3, m
    ctx.set_patterns([b"A" * 4097], 100)
    assert ctx.pattern_kernel(0) == 1
    ctx.set_patterns([bytes(range(256)) * 8], 100)      # 2048 bytes over 256 distinct ones: the Eq rows do not fit LDS
    assert ctx.pattern_kernel(0) == 1


@pytest.mark.parametrize("m,k", [(129, 40), (200, 9), (256, 64), (257, 3), (300, 20), (400, 150), (512, 8),
                                 (513, 5), (600, 30), (700, 12), (768, 200), (769, 9), (1000, 40), (1024, 6),
                                 (1025, 6), (1300, 25), (2047, 11), (2048, 300), (2049, 8), (3000, 50), (4096, 16)])
def test_wide_bitvector_columns_vs_oracle(ctx, apm, m, k):
    """BITPAR beyond 128 bytes: 8- and 16-word columns (<= 512), 24 / 32 words with the one-pass column step (<= 1024), one
    window per wave (<= 4096: 32- and 64-bit rows per lane).  Planted occurrences with substitutions and indels in 60 KB of
    DNA, truncated tail windows included; AUTO and forced BITPAR == the CPU oracle (and == GENERIC)."""
    rnd = random.Random(7 * m + k)
    banded = 8 * k <= m                                      # the oracle's banded form (pinned by the goldens) where the band is narrow
    n = 60000 if m <= 512 else max(3 * m + 1000, min(60000, int((2e10 if banded else 4e10) / (m * (2 * k + 1) if banded else m * m))))
    text = bytearray(rnd.choice(b"ACGT") for _ in range(n))
    pat = bytes(rnd.choice(b"ACGT") for _ in range(m))
    for i in range(12):
        w = bytearray(pat)
        for _e in range(rnd.randrange(0, min(k, 12) + 1)):
            r, pos = rnd.random(), rnd.randrange(len(w))
            if r < 0.6:
                w[pos] = rnd.choice(b"ACGT")
            elif r < 0.8:
                del w[pos]
                w.append(rnd.choice(b"ACGT"))
            else:
                w.insert(pos, rnd.choice(b"ACGT"))
                w.pop()
        o = n - m - 3 if i == 0 else rnd.randrange(0, n - m)     # one right at the end of the text
        text[o:o + m] = w
    text = bytes(text)
    want = H.oracle_counts(text, [pat], k, banded=banded)
    assert want[0] >= 1
    for variant in ("auto", "bitpar", "generic"):
        ctx.set_kernel(variant)
        ctx.set_patterns([pat], k)
        assert ctx.count_buffer(text) == want, (variant, m, k)
    ctx.set_kernel("auto")


@pytest.mark.parametrize("m,k,alphabet", [(600, 9, b"etaoin shrdlucmfwypvbgkqjxz\n"), (1500, 12, bytes(range(32, 112))), (3000, 20, b"ACGTN\n"),
                                          (4096, 7, bytes(range(1, 41)))])
def test_long_patterns_over_bigger_alphabets(ctx, apm, m, k, alphabet):
    """The long bit-vector forms keep one Eq row per distinct pattern byte in LDS (24/32 words per row up to 1024 bytes, 64
    or 128 beyond): prose-like, 80-letter and 40-letter alphabets, text bytes the pattern does not contain, an occurrence cut
    by the end of the text (truncated windows).  AUTO == forced BITPAR == the CPU oracle's banded form."""
    rnd = random.Random(m + k)
    n = 3 * m + 2000
    text = bytearray(rnd.choice(alphabet + b"\x00\xff") for _ in range(n))
    pat = bytearray(rnd.choice(alphabet) for _ in range(m))
    for o in (17, n - m - 1, n - m + 40):                      # the last one leaves only a truncated window
        w = bytearray(pat)
        for _e in range(k - 2):
            w[rnd.randrange(m)] = rnd.choice(alphabet)
        del w[m // 3]
        w.insert(2 * m // 3, rnd.choice(alphabet))
        text[o:o + m] = w[:max(0, min(m, n - o))]
    text, pat = bytes(text[:n]), bytes(pat)
    want = H.oracle_counts(text, [pat], k, banded=True)
    assert want[0] >= 2
    for variant in ("auto", "bitpar"):
        ctx.set_kernel(variant)
        ctx.set_patterns([pat], k)
        assert ctx.pattern_kernel(0) == 3
        assert ctx.count_buffer(text) == want, (variant, m, k)
    ctx.set_kernel("auto")


@pytest.mark.parametrize("alphabet", [b"ACGT", b"ab", b"ACGTN\n", bytes(range(65, 81))])
def test_short_loose_patterns_through_the_automaton(ctx, apm, alphabet):
    """Short patterns with many errors (pieces too short for BANDED's filter): AUTO routes them to the k-error automaton
    over 32 window starts per lane (apm_nfa.hip).  Every (m, k) with m + k/2 <= 32, k <= 7 and m / (k+1) < 4 on texts
    with planted edited occurrences, shard cuts at odd offsets, text shorter than a lane's 64 bytes: AUTO == forced NFA ==
    forced BITPAR == the CPU oracle."""
    import torch
    rnd = random.Random(len(alphabet) * 7919)
    n = 70000
    text = bytearray(rnd.choice(alphabet) for _ in range(n))
    sets = {}
    for k in range(0, 8):
        for m in range(1, 33):
            if m + k // 2 > 32 or m // (k + 1) >= 4 or k >= m:
                continue
            if rnd.random() > (1.0 if m in (1, 2, 3, 12, 15, 28, 29) or k in (0, 7) else 0.25):
                continue
            o = rnd.randrange(0, n - m)
            p = bytearray(text[o:o + m])
            for _e in range(rnd.randrange(0, k + 1)):
                r, pos = rnd.random(), rnd.randrange(len(p))
                if r < 0.5:
                    p[pos] = rnd.choice(alphabet)
                elif r < 0.75 and len(p) > 1:
                    del p[pos]
                    p.append(rnd.choice(alphabet))
                else:
                    p.insert(pos, rnd.choice(alphabet))
                    p.pop()
            sets.setdefault(k, []).append(bytes(p))
    text = bytes(text)
    assert len(sets) == 8
    for k, pats in sorted(sets.items()):
        want = H.oracle_counts(text, pats, k)
        ctx.set_kernel("auto")
        ctx.set_patterns(pats, k)
        assert all(ctx.pattern_kernel(i) == 5 for i in range(len(pats))), k
        assert ctx.count_buffer(text) == want, ("auto", k)
        for variant in ("nfa", "bitpar"):
            ctx.set_kernel(variant)
            assert ctx.count_buffer(text) == want, (variant, k)
        ctx.set_kernel("auto")
        assert ctx.count_buffer(text[:37]) == H.oracle_counts(text[:37], pats, k), ("short text", k)
        # owner ranges cut at odd places, device text at an unaligned address
        t = torch.empty(n + 64, dtype=torch.uint8, device="cuda:0")
        t[3:3 + n] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to("cuda:0")
        cnt = torch.zeros(len(pats), dtype=torch.int64, device="cuda:0")
        m_max = max(len(p) for p in pats)
        total = [0] * len(pats)
        cuts = [0, 1, 33, 4097, 40001, n]
        for ob, oe in zip(cuts[:-1], cuts[1:]):
            hi = min(n, oe + m_max - 1)
            cnt.zero_()
            torch.cuda.synchronize()
            ctx.count_shard_device(t.data_ptr() + 3 + ob, ob, hi - ob, n, ob, oe, cnt.data_ptr())
            ctx.synchronize()
            total = [a + b for a, b in zip(total, cnt.cpu().tolist())]
        assert total == want, ("shards", k)
    ctx.set_kernel("auto")
    ctx.set_patterns([bytes(range(40, 60))], 5)           # 20 distinct bytes: no automaton
    assert ctx.pattern_kernel(0) == 3
    with pytest.raises(apm.ApmError):
        ctx.set_kernel("nfa")
    ctx.set_kernel("auto")


def test_reference_gpu_entry_points_link_level(tmp_path):
    """include/apm_refshim.h: a C program calls getDeviceCount/setDevice, invoke_kernel/write_kernel_result and
    initializeGPU/getGPUResult exactly as the reference's host files do, linked against libapm_hip.so only."""
    exe = str(tmp_path / "refshim_test")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(H.ROOT, "include"), os.path.join(H.ROOT, "tests", "refshim_test.c"),
                    "-o", exe, "-L", H.PKG_DIR, "-lapm_hip", "-Wl,-rpath," + H.PKG_DIR, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    for name in ("x100_k2", "cfg1_basic_test", "chrY_k5"):
        c = next(c for c in CASES if c["name"] == name)
        text, pats, k = H.case_text(c), c["patterns"], c["k"]
        n, P = len(text), len(pats)
        r = subprocess.run([exe, str(k), c["path"]] + [p.decode() for p in pats], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lines = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in r.stdout.decode().splitlines() if l.strip()}
        assert lines["devices"] == [1]
        assert lines["invoke"] == c["counts"]                     # whole text = src/sequential.c's count
        want34 = []
        for p in pats:                                            # the GPU's share in patterns_over_ranks.c:316-326
            part = min(n, 3 * n // 4 + len(p) - 1)
            want34 += H.oracle_counts(text[:part], [p], k)
        assert lines["invoke34"] == want34
        last = P - 1 if P > 1 else P
        for rank, (start, end) in enumerate([(0, n // 2), (n // 2, n)]):
            want = []
            for i, p in enumerate(pats):                          # searchPattern's per-rank arithmetic
                if i >= last:
                    want.append(0)
                    continue
                e = min(n, end + (len(p) - 1 if rank == 0 else 0))
                want += H.oracle_counts(text[:e], [p], k, j_begin=start)
            assert lines["db%d" % rank] == want, (name, rank)


# ---------------------------------------------------------------- the C host (reference CLI contract)
CLI = os.path.join(H.PKG_DIR, "host", "apm_parallel")


def _cli(args):
    return subprocess.run([CLI] + args, capture_output=True)


@pytest.mark.skipif(not os.path.exists(CLI), reason="host/apm_parallel not built")
def test_cli_matches_reference_stdout():
    """scripts/basic_test.batch:10-18 / README.md:54-92: same banner and result lines."""
    c = next(c for c in CASES if c["name"] == "cfg1_basic_test")
    for extra in ([], ["DB_OVER_RANKS"], ["PATTERNS_OVER_RANKS"], ["--gpus", "1"], ["--kernel", "wavefront"]):
        r = _cli(["0", c["path"]] + [p.decode() for p in c["patterns"]] + extra)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.decode().splitlines()
        assert lines[0] == ("Approximate Pattern Mathing: looking for 6 pattern(s) in file %s w/ distance of 0" % c["path"])
        assert lines[1].startswith("APM done in ") and lines[1].endswith(" s")
        want = ["Number of matches for pattern <%s>: %d" % (p.decode(), n) for p, n in zip(c["patterns"], c["counts"])]
        assert lines[2:] == want
    c = next(c for c in CASES if c["name"] == "x100_k3")
    r = _cli([str(c["k"]), c["path"]] + [p.decode() for p in c["patterns"]])
    got = [int(l.rsplit(": ", 1)[1]) for l in r.stdout.decode().splitlines() if l.startswith("Number of matches")]
    assert got == c["counts"]


@pytest.mark.skipif(not os.path.exists(CLI), reason="host/apm_parallel not built")
def test_batch_runner_script():
    """scripts/run_tests.sh = the reference's scripts/basic_test.batch + scripts/run_tests invocations,
    diffing the result lines against fixtures written from the reference binary."""
    r = subprocess.run(["bash", os.path.join(H.ROOT, "scripts", "run_tests.sh"), "1"], capture_output=True)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert r.stdout.decode().count("result OK") == 5


@pytest.mark.skipif(not os.path.exists(CLI), reason="host/apm_parallel not built")
def test_cli_error_paths_match_reference():
    errs = H.golden()["cli_errors"]
    dna = os.path.join(H.GOLDEN_DIR, "dna")
    for e in errs:
        args = [a.replace("<dna>", dna) for a in e["args"]]
        r = _cli(args)
        assert r.returncode == e["rc"]
        assert r.stderr.decode("latin-1") == e["stderr"]
        assert r.stdout.decode("latin-1").replace(CLI, "<exe>").replace(dna, "<dna>") == e["stdout"]


# ---------------------------------------------------------------- adversarial inputs for the filter
@pytest.mark.parametrize("variant", ["auto", "banded", "bitpar"])
def test_low_entropy_text_every_window_is_a_candidate(ctx, apm, variant):
    """runs of one letter / short periods: every position hits the q-gram filter, every window is
    verified; counts must still equal the oracle (rule 26: force the rare branch)."""
    texts = [b"A" * 9000, b"AC" * 4500, (b"ACGT" * 8 + b"T") * 280, b"A" * 4095 + b"C" + b"A" * 4904]
    for text in texts:
        for k in (0, 1, 2, 3, 5):
            pats = [text[100:100 + m] for m in (24, 32, 50, 64, 128)] + [b"A" * 31 + b"C", b"C" + b"A" * 40]
            pats = [p for p in pats if _supported(variant, p, k)]
            want = H.oracle_counts(text, pats, k, banded=True)
            assert _run(ctx, apm, variant, pats, k, text) == want, (len(text), k)


def test_custom_unaligned_shard_cuts(ctx, apm):
    """own ranges cut at arbitrary (non 16-byte) positions add up to the whole"""
    c = next(c for c in CASES if c["name"] == "x100_k2")
    text, pats, k = H.case_text(c), c["patterns"], c["k"]
    n = len(text)
    m_max = max(len(p) for p in pats)
    for variant in ("auto", "bitpar", "wavefront"):
        ctx.set_kernel("auto")
        ctx.set_patterns(pats, k)
        ctx.set_kernel(variant)
        d_counts = ctx.device_alloc(8 * len(pats))
        ctx.device_memset(d_counts, 0, 8 * len(pats))
        cuts = [0, 7, 1001, 4099, 65537, 100003, n]
        for b, e in zip(cuts[:-1], cuts[1:]):
            lo, hi = max(0, b - 3), min(n, e + m_max - 1)
            d_text = ctx.device_alloc(hi - lo + 16)
            ctx.device_upload(d_text, text[lo:hi])
            ctx.count_shard_device(d_text, lo, hi - lo, n, b, e, d_counts)
            ctx.synchronize()
            ctx.device_free(d_text)
        raw = ctx.device_download(d_counts, 8 * len(pats))
        ctx.device_free(d_counts)
        assert [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(len(pats))] == c["counts"], variant


# ---------------------------------------------------------------- scale: 64-bit offsets, big files, RCCL
def test_text_beyond_4gib_on_device(ctx, apm):
    """5 GiB synthetic text on one device (the reference's int / single read() stop at 2 GiB):
    planted exact copies + truncated-tail matches, BASELINE cfg2's pattern set."""
    wl = H.workloads()
    c = wl.CONFIGS["cfg2"]
    n, k, seed = 5 << 30, c["k"], wl.seed_of(c["cid"])
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    ctx.set_kernel("auto")
    ctx.set_patterns(pats, k)
    got = ctx.count_synthetic(n, seed)
    assert got == wl.expected_counts_k0(n, pats, planted, seed)
    t = ctx.timing()
    assert t["text_bytes"] >= n


def test_shard_beyond_4gib_runs_the_pipeline_in_pieces(apm):
    """5 GiB of device text with BASELINE cfg3's pattern set as ONE shard: beyond the sieve + verify pipeline's 32-bit
    offsets, so the runtime scans it in pieces of 3 GiB of window starts (a piece seam in the middle of the text).
    Same counts as the caller's own cut into two shards at another place, and as the full-DP BITPAR kernel on the whole
    5 GiB (64-bit positions, both seams, two kernel families); planted occurrences found."""
    import torch
    wl = H.workloads()
    c = wl.CONFIGS["cfg3"]
    n, k, seed = 5 << 30, c["k"], wl.seed_of(c["cid"])
    pats, planted = wl.make_patterns(n, c["lens"], k, seed)
    P, m_max = len(pats), max(len(p) for p in pats)
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    cnt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c2:
        c2.set_patterns(pats, k)
        assert c2.stat("sieve_on") == 1
        c2.synth_fill_device(text.data_ptr(), 0, n, seed)
        c2.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr())
        c2.synchronize()
        whole = cnt.cpu().tolist()
        assert c2.stat("sieve_candidates") > 0                      # the pipeline ran (the last piece's masks)
        assert [l for l, _ in c2.launch_times()].count("tile") == 0
        cnt.zero_()
        torch.cuda.synchronize()
        cut = (n // 2 + 12345) & ~15
        for lo, hi in ((0, cut), (cut, n)):
            end = min(n, hi + m_max - 1)
            c2.count_shard_device(text.data_ptr() + lo, lo, end - lo, n, lo, hi, cnt.data_ptr())
            c2.synchronize()
        assert cnt.cpu().tolist() == whole
        c2.set_kernel("bitpar")
        cnt.zero_()
        torch.cuda.synchronize()
        c2.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr())
        c2.synchronize()
        assert cnt.cpu().tolist() == whole
    for cc, (o, d) in zip(whole, planted):
        assert cc >= (1 if d <= k else 0)
    del text


def test_cli_file_larger_than_2gib(tmp_path_factory, apm):
    """apm_parallel on a 2.25 GiB file (64-bit chunked ingest) == device-resident scan of the same bytes"""
    if not os.path.exists(CLI):
        pytest.skip("host/apm_parallel not built")
    wl = H.workloads()
    n, k, seed = (9 << 28) + 12345, 1, wl.seed_of(4)
    lens = [40, 64, 100]
    pats, planted = wl.make_patterns(n, lens, k, seed)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path_factory.mktemp("big"))
    path = os.path.join(base, "apm_big_%d.fa" % os.getpid())
    try:
        with open(path, "wb") as f:
            step = 1 << 26
            for off in range(0, n, step):
                f.write(apm.synth_fill_host(off, min(step, n - off), seed))
        with apm.ApmContext(device=0) as c2:
            c2.set_patterns(pats, k)
            want = c2.count_synthetic(n, seed)
            assert all(w >= 1 for w, (o, d) in zip(want, planted) if d <= k)
            assert c2.count_file(path) == want
        r = _cli([str(k), path] + [p.decode() for p in pats])
        assert r.returncode == 0, r.stderr
        got = [int(l.rsplit(": ", 1)[1]) for l in r.stdout.decode().splitlines() if l.startswith("Number of matches")]
        assert got == want
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_rccl_allreduce_path_single_device(apm):
    """single-process mode: counts go through ncclAllReduce (dlopen'ed librccl) when APM_FORCE_RCCL=1"""
    c = next(c for c in CASES if c["name"] == "chrY_k3")
    env = dict(os.environ, APM_FORCE_RCCL="1")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import helpers as H; apm = H.pkg();"
            "c = next(c for c in H.golden()['cases'] if c['name'] == 'chrY_k3');"
            "ctx = apm.ApmContext(n_devices=1); ctx.set_patterns(c['patterns'], c['k']);"
            "print(ctx.count_buffer(H.case_text(c)))") % (H.ROOT, os.path.join(H.ROOT, "tests"))
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().strip().splitlines()[-1] == str(c["counts"])


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_multi_device_context_rehearsed_on_one_gpu(apm, tmp_path, devices):
    """The single-process multi-device path (apm_create(G): one DeviceState per shard -- own stream, text buffer, plan copy
    and count vector --, concurrent staging threads, partial counts summed at the end) rehearsed on this one GPU:
    APM_DEVICES=0,0 makes a context of two (three) shards that share device 0.  Counts must equal the single-device
    counts for the reference's data files (host buffer and file ingest), for a 64 MiB cfg3 text through the pinned staging
    ring, and for the device-side generator."""
    os.environ["APM_DEVICES"] = devices
    try:
        multi = apm.ApmContext(n_devices=0)
    finally:
        del os.environ["APM_DEVICES"]
    G = len(devices.split(","))
    try:
        assert multi.timing()["n_devices"] in (0, G)
        for name in ("x100_k2", "chrY_k3"):
            c = next(c for c in CASES if c["name"] == name)
            text = H.case_text(c)
            multi.set_patterns(c["patterns"], c["k"])
            assert multi.count_buffer(text) == c["counts"], (name, devices)
            assert multi.timing()["n_devices"] == G
            f = tmp_path / (name + ".txt")
            f.write_bytes(text)
            assert multi.count_file(str(f)) == c["counts"], (name, devices)
        wl = H.workloads()
        cfg = wl.CONFIGS["cfg3"]
        n, k, seed = 64 << 20, cfg["k"], wl.seed_of(cfg["cid"])
        pats, planted = wl.make_patterns(n, cfg["lens"], k, seed)
        text = apm.synth_fill_host(0, n, seed)
        with apm.ApmContext(device=0) as one:
            one.set_patterns(pats, k)
            want = one.count_buffer(text)
        multi.set_patterns(pats, k)
        assert multi.count_buffer(text) == want
        assert multi.count_synthetic(n, seed) == want
        f = tmp_path / "cfg3_64m.txt"
        f.write_bytes(text)
        assert multi.count_file(str(f)) == want
        assert sum(want) > 0 and all(cc >= 1 for cc, (o, d) in zip(want, planted) if d <= k)
    finally:
        multi.close()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_pattern_sharded_context_rehearsed_on_one_gpu(apm, tmp_path, devices):
    """apm_set_partition(APM_PARTITION_PATTERNS): the pattern list cut into contiguous slices, one single-device child
    context per device, each scanning the WHOLE text (the replacement of the reference's PATTERNS_OVER_RANKS,
    src/patterns_over_ranks.c:160-182) -- rehearsed with children sharing this one GPU (APM_DEVICES).  Same counts as the
    golden vectors / the single-device context for buffer, file and generator ingest, more devices than patterns, a
    forced kernel, match positions, and through the C CLI's trailing PATTERNS_OVER_RANKS."""
    os.environ["APM_DEVICES"] = devices
    try:
        multi = apm.ApmContext(n_devices=0)
    finally:
        del os.environ["APM_DEVICES"]
    G = len(devices.split(","))
    try:
        multi.set_partition("patterns")
        for name in ("x100_k2", "chrY_k3", "cfg1_basic_test"):
            c = next(c for c in CASES if c["name"] == name)
            text = H.case_text(c)
            multi.set_patterns(c["patterns"], c["k"])
            assert multi.count_buffer(text) == c["counts"], (name, devices)
            assert multi.timing()["n_devices"] == G
            f = tmp_path / (name + ".txt")
            f.write_bytes(text)
            assert multi.count_file(str(f)) == c["counts"], (name, devices)
            multi.set_kernel("bitpar")
            assert multi.count_buffer(text) == c["counts"], (name, "bitpar")
            multi.set_kernel("auto")
            multi.set_patterns(c["patterns"][:1], c["k"])                  # fewer patterns than devices
            assert multi.count_buffer(text) == c["counts"][:1]
        c = next(c for c in CASES if c["name"] == "chrY_k3")
        text = H.case_text(c)
        multi.set_patterns(c["patterns"], c["k"])
        last = len(c["patterns"]) - 1
        got_pos, total = multi.find_buffer(text, last)
        assert got_pos == _oracle_positions(text, c["patterns"][last], c["k"]) and total == c["counts"][last]
        multi.set_partition("text")                                          # and back: the same context, text-sharded
        assert multi.count_buffer(text) == c["counts"]
        multi.set_partition("patterns")
        wl = H.workloads()
        cfg = wl.CONFIGS["cfg5"]
        n, k, seed = 16 << 20, cfg["k"], wl.seed_of(cfg["cid"])
        pats, planted = wl.make_patterns(n, cfg["lens"], k, seed)
        with apm.ApmContext(device=0) as one:
            one.set_patterns(pats, k)
            want = one.count_synthetic(n, seed)
        multi.set_patterns(pats, k)
        assert multi.count_synthetic(n, seed) == want and sum(want) > 0
    finally:
        multi.close()
    c = next(c for c in CASES if c["name"] == "x100_k2")
    cli = os.path.join(H.PKG_DIR, "host", "apm_parallel")
    r = subprocess.run([cli, str(c["k"]), c["path"]] + [p.decode() for p in c["patterns"]] + ["PATTERNS_OVER_RANKS", "--gpus", str(G)],
                       capture_output=True, env=dict(os.environ, APM_DEVICES=devices), timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    got = [int(l.rsplit(":", 1)[1]) for l in r.stdout.decode().splitlines() if l.startswith("Number of matches")]
    assert got == c["counts"]


def test_bench_two_ranks_equal_one_rank():
    """bench.py's N > 1 path (owner-computes shards + halo + all-reduce of the partial counts), rehearsed with two
    ranks sharing this one GPU over gloo (RCCL refuses two ranks on one device): the summed counts must equal the
    one-rank counts of the same text.  The ranks are started by torch.distributed.run from a fresh interpreter."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    per_gpu = 48 << 20
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants", "--no-per-config"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([os.sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "1", "--bytes-per-gpu", str(2 * per_gpu)] + common,
                         capture_output=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    two = subprocess.run([os.sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(H.ROOT, "bench.py"),
                          "--gpus", "2", "--dist-backend", "gloo", "--bytes-per-gpu", str(per_gpu)] + common,
                         capture_output=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    a = json.loads(one.stdout.decode().strip().splitlines()[-1])
    b = json.loads(two.stdout.decode().strip().splitlines()[-1])
    assert a["config"]["workload"].startswith("cfg3") and b["config"]["workload"].startswith("cfg3")
    assert a["config"]["text_bytes_total"] == b["config"]["text_bytes_total"] == 2 * per_gpu
    assert b["n_gpus"] == 2 and a["counts"] == b["counts"] and sum(a["counts"]) > 0
    assert a["planted_occurrences_found"] and b["planted_occurrences_found"]


def test_bench_two_ranks_run_the_whole_cfg4_and_cfg5():
    """bench.py at N > 1 also times BASELINE.json configs[3..4] as they are -- ONE text of 2^33 bytes cut into N owner ranges,
    pattern sets planted in that text, the partial counts all-reduced -- beside the weak-scaled headline.  Rehearsed with two
    ranks on this one GPU over gloo (4 GiB of text per rank: the pipeline scans it in pieces)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([os.sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(H.ROOT, "bench.py"),
                          "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants"],
                         capture_output=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    b = json.loads(two.stdout.decode().strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and b["config"]["workload"].startswith("cfg3") and b["planted_occurrences_found"]
    for name in ("cfg4", "cfg5"):
        r = b["per_config"][name]
        assert "the whole text" in r["workload"] and r["text_bytes_total"] == 1 << 33 and r["text_bytes_per_gpu"] == 1 << 32
        assert r["planted_occurrences_found"] and r["value"] > 0 and r["roofline"]["frac"] > 0


# ---------------------------------------------------------------- filter corner cases (chains, overflow, big k)
@pytest.mark.parametrize("variant", ["auto", "banded"])
def test_repeated_pieces_and_identical_patterns(ctx, apm, variant):
    """patterns whose pigeonhole pieces are equal (fingerprint chains in the hash table), identical
    patterns (every key duplicated), and periodic texts where every piece matches at every shift:
    the stateless dedup must still count each window exactly once"""
    rnd = random.Random(77)
    unit = b"ACGTTGCA"
    texts = [unit * 600, b"".join(rnd.choice([unit, b"ACGTTGCC", b"TTTTTTTT"]) for _ in range(700))]
    for text in texts:
        for k in (0, 1, 2, 3, 5, 7):
            pats = [unit * 4, unit * 4, unit * 8, (unit * 8)[3:35], unit * 16, unit * 32]
            pats = [p for p in pats if _supported(variant, p, k)]
            want = H.oracle_counts(text, pats, k, banded=True)
            assert _run(ctx, apm, variant, pats, k, text) == want, (k, [len(p) for p in pats])


def test_many_keys_one_launch_and_split_launches(ctx, apm):
    """> 4096 sub-keys / > 16 KiB of pattern bytes force several BANDED launches and full hash buckets"""
    rnd = random.Random(2024)
    text = bytes(rnd.choice(b"ACGT") for _ in range(30000))
    pats = []
    for i in range(700):
        m = rnd.choice([32, 40, 64, 100])
        o = rnd.randrange(0, len(text) - m)
        p = bytearray(text[o:o + m])
        if i % 3 == 0:
            p[rnd.randrange(m)] = rnd.choice(b"ACGT")
        pats.append(bytes(p))
    for k in (1, 3):
        want = H.oracle_counts(text, pats, k, banded=True)
        assert _run(ctx, apm, "auto", pats, k, text) == want
        assert sum(want) > 500


@pytest.mark.parametrize("k", [4, 6, 7])
def test_wide_bands(ctx, apm, k):
    """band half-width 2 and 3 (k up to 7), pattern lengths up to 256, indels near the window ends"""
    rnd = random.Random(100 + k)
    text = bytearray(rnd.choice(b"ACGT") for _ in range(20000))
    pats = []
    for m in (64, 100, 128, 200, 256):
        o = rnd.randrange(100, len(text) - m - 100)
        p = bytearray(text[o:o + m])
        for _e in range(k):
            r, pos = rnd.random(), rnd.randrange(len(p))
            if r < 0.4:
                p[pos] = rnd.choice(b"ACGT")
            elif r < 0.7:
                del p[pos]
                p.append(rnd.choice(b"ACGT"))
            else:
                p.insert(pos, rnd.choice(b"ACGT"))
                p.pop()
        pats.append(bytes(p))
    text = bytes(text)
    want = H.oracle_counts(text, pats, k, banded=True)
    assert sum(want) >= 3
    for variant in ("auto", "banded", "wavefront"):
        assert _run(ctx, apm, variant, pats, k, text) == want, variant


# ---------------------------------------------------------------- match positions
def _oracle_positions(text, p, k):
    n, m = len(text), len(p)
    out = []
    for j in range(0, max(0, n - k)):
        size = min(m, n - j)
        if H.window_distance(p[:size], text[j:j + size]) <= k:
            out.append(j)
    return out


def test_find_positions_equal_oracle(ctx, apm):
    rnd = random.Random(31)
    text = bytes(rnd.choice(b"ACGT") for _ in range(3000)) + b"ACGTACGTAC"
    pats = [text[100:132], text[500:516], b"ACGTACGTACGT", text[1000:1200], b"GG",
            text[2600:2900], text[-260:] + b"TTTTTT", text[2000:2512]]   # 16-word columns; a truncated tail window of a long pattern
    for k in (0, 2, 3):
        ctx.set_kernel("auto")
        ctx.set_patterns(pats, k)
        counts = ctx.count_buffer(text)
        for i, p in enumerate(pats):
            want = _oracle_positions(text, p, k)
            got, total = ctx.find_buffer(text, i, capacity=8192)
            assert total == len(want) == counts[i]
            assert got == want
        assert ctx.count_buffer(text) == counts          # the pattern set survived the find calls
    got, total = ctx.find_buffer(text, 4, capacity=3)     # capacity smaller than the number of matches
    assert total == counts[4] and len(got) == 3 and set(got) <= set(_oracle_positions(text, pats[4], 3))


@pytest.mark.skipif(not os.path.exists(CLI), reason="host/apm_parallel not built")
def test_cli_positions_flag():
    c = next(c for c in CASES if c["name"] == "chrY_k2")
    text = H.case_text(c)
    r = _cli([str(c["k"]), c["path"]] + [p.decode() for p in c["patterns"]] + ["--positions"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.decode().splitlines()
    pos_lines = [l for l in lines if l.startswith("Positions for pattern")]
    assert len(pos_lines) == len(c["patterns"])
    for l, p, cnt in zip(pos_lines, c["patterns"], c["counts"]):
        got = [int(x) for x in l.split(">:", 1)[1].split()]
        assert len(got) == cnt and got == _oracle_positions(text, p, c["k"])


# ---------------------------------------------------------------- every kernel form of the BANDED path
_FORM_WORKER = r"""
import json, random, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import helpers as H
apm = H.pkg()
rng = random.Random(20241)
lrng = random.Random(777)   # (its own stream: the mixed sets stay what they were)
out = {}
for name, alphabet, n in (("dna", b"ACGT", 300000), ("prose", b"etaoin shrdlucETAOIN\n.,", 200000)):
    trng = random.Random(name)                              # the parent regenerates the text from this seed
    text = bytearray(trng.choice(alphabet) for _ in range(n))
    for k in (0, 1, 3, 4):
        pats = []
        for m in (16, 20, 27, 30, 40, 59, 64, 100, 128):
            o = rng.randrange(0, len(text) - m)
            p = bytearray(text[o:o + m])
            for _ in range(rng.randrange(0, k + 2)):          # substitutions
                p[rng.randrange(m)] = rng.choice(alphabet)
            if k >= 2 and rng.random() < 0.5:                  # one deletion + one insertion (keeps the length)
                i, j = sorted(rng.sample(range(1, m - 1), 2))
                del p[i]; p.insert(j, rng.choice(alphabet))
            pats.append(bytes(p))
        with apm.ApmContext(device=0) as ctx:
            ctx.set_patterns(pats, k)
            out["%s:%d" % (name, k)] = dict(patterns=[p.decode("latin-1") for p in pats], counts=ctx.count_buffer(bytes(text)),
                                            kernels=[ctx.pattern_kernel(i) for i in range(len(pats))], clist=ctx.stat("sieve_clist"))
    for k in (2, 3, 4):   # long pieces only (>= 15 bytes): the sampled (stride-8) sieve
        pats = []
        for m in (80, 96, 100, 128):
            o = lrng.randrange(0, len(text) - m)
            p = bytearray(text[o:o + m])
            for _ in range(lrng.randrange(0, k - 1)):           # <= k - 2 substitutions
                p[lrng.randrange(m)] = lrng.choice(alphabet)
            if lrng.random() < 0.5:                               # + one deletion and one insertion: still within k
                i, j = sorted(lrng.sample(range(1, m - 1), 2))
                del p[i]; p.insert(j, lrng.choice(alphabet))
            pats.append(bytes(p))
        with apm.ApmContext(device=0) as ctx:
            ctx.set_patterns(pats, k)
            out["%s:%d:long" % (name, k)] = dict(patterns=[p.decode("latin-1") for p in pats], counts=ctx.count_buffer(bytes(text)),
                                                 kernels=[ctx.pattern_kernel(i) for i in range(len(pats))],
                                                 stride=ctx.stat("sieve_stride"), fused=ctx.stat("sieve_fused"))
print(json.dumps(out))
"""


@pytest.mark.parametrize("env", [{}, {"APM_FILTER_STREAM": "0"}, {"APM_FILTER_STREAM": "2"}, {"APM_FILTER_STREAM": "3"},
                                 {"APM_FILTER_DMA": "0"}, {"APM_FILTER_STREAM": "2", "APM_FILTER_DMA": "0"},
                                 {"APM_SIEVE": "0"}, {"APM_SIEVE": "0", "APM_FILTER_STREAM": "2"},
                                 {"APM_FUSED": "1"},          # sieve + verify in one kernel for every sieved set (default: sampled sets only)
                                 {"APM_FUSED": "0"},          # ... for none
                                 {"APM_SIEVE_CF": "0"},       # the sieve without its second stage (the code filter)
                                 {"APM_FUSED_RC": "0"},       # fused sampled form: the register compare packs its operands per hit (sets of > 128 units always do)
                                 {"APM_SIEVE_CLIST": "0"},    # the code filter's survivors handed over as mask rows + block list
                                 {"APM_CLIST_REGION_CAP": "1"},   # candidate-list regions of one / five entries: what does not fit leaves
                                 {"APM_CLIST_REGION_CAP": "5"}],  # through the rows of its block (list and rows mixed)
                         ids=lambda e: ",".join("%s=%s" % (k[4:], v) for k, v in e.items()) or "default")
def test_every_filter_kernel_form_agrees_with_oracle(env):
    """The BANDED path picks between the LDS-tile kernel (LDS-DMA or register-staged) and the wave-autonomous
    stream kernel per launch; the APM_FILTER_* switches force each form (they are read once per process,
    hence the subprocess).  All forms must give the oracle's counts."""
    r = subprocess.run([os.sys.executable, "-c", _FORM_WORKER, H.ROOT, os.path.join(H.ROOT, "tests")],
                       capture_output=True, env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = json.loads(r.stdout.decode().strip().splitlines()[-1])
    texts = {}
    for name, alphabet, n in (("dna", b"ACGT", 300000), ("prose", b"etaoin shrdlucETAOIN\n.,", 200000)):
        trng = random.Random(name)
        texts[name] = bytes(bytearray(trng.choice(alphabet) for _ in range(n)))
    assert len(got) == 14
    for key, res in got.items():
        name, k = key.split(":")[:2]
        if key.endswith(":long") and env.get("APM_SIEVE") != "0":
            assert res["stride"] == 8 and res["fused"] == (0 if env.get("APM_FUSED") == "0" else 1), (key, env, res["stride"], res["fused"])
        pats = [p.encode("latin-1") for p in res["patterns"]]
        for p, kern in zip(pats, res["kernels"]):
            want_kern = 4 if len(p) // (int(k) + 1) >= 4 else (5 if int(k) <= 7 and len(p) + int(k) // 2 <= 32 and len(set(p)) <= 16 else 3)
            assert kern == want_kern, "BANDED wherever the pieces are long enough, the automaton for short loose patterns, else the bit-vector kernel"
        assert res["counts"] == H.oracle_counts(texts[name], pats, int(k), banded=True), (key, env)
        assert sum(res["counts"]) >= (4 if key.endswith(":long") else 5)
    # the candidate list is the hand-over of the code-filter sieve unless a switch takes the filter, the list or the sieve away
    with_list = [key for key, res in got.items() if res.get("clist")]
    if any(env.get(sw) is not None for sw in ("APM_SIEVE", "APM_FUSED", "APM_SIEVE_CF", "APM_SIEVE_CLIST")):  # (APM_FUSED_RC touches sampled sets only)
        assert not with_list or env.get("APM_FUSED") == "0", (env, with_list)
    else:
        assert with_list, env


@pytest.mark.parametrize("m,k", [(16, 3), (20, 3), (36, 3), (50, 5), (24, 2)])
def test_edited_occurrences_at_every_offset_around_tile_seams(ctx, apm, m, k):
    """Occurrences carrying one deletion + one insertion (so every piece shift of the band is exercised, forward
    and backward partner pre-checks included), planted so that their starts sweep all offsets around the tile
    kernel's tile boundaries (tile width = (4096 - 16 - m - k//2) & ~31 window starts)."""
    rng = random.Random(1000 * m + k)
    tile_w = (4096 - 16 - m - k // 2) & ~31
    pat = bytes(rng.choice(b"ACGT") for _ in range(m))
    n_occ = 150
    text = bytearray(rng.choice(b"ACGT") for _ in range(tile_w * (n_occ + 2)))
    for i in range(n_occ):
        w = bytearray(pat)
        kind = i % 5
        if kind >= 1:  # a deletion and an insertion at varying places (distance 2, length kept)
            a = 1 + (i * 7) % (m - 3)
            b = 1 + (i * 11) % (m - 3)
            if kind in (1, 2):
                del w[min(a, b)]
                w.insert(max(a, b), rng.choice(b"ACGT"))
            else:
                w.insert(min(a, b), rng.choice(b"ACGT"))
                del w[max(a, b) + 1 if max(a, b) + 1 < len(w) else len(w) - 1]
            if kind == 2 and k >= 3:  # plus a substitution
                w[(i * 5) % m] = rng.choice(b"ACGT")
        assert len(w) == m
        pos = tile_w * (i + 1) - 75 + i  # sweeps [-75, +75) around a boundary
        text[pos:pos + m] = w
    text = bytes(text)
    want = H.oracle_counts(text, [pat], k, banded=True)
    assert want[0] >= n_occ // 2
    for variant in ("auto", "banded"):
        ctx.set_kernel(variant)
        ctx.set_patterns([pat], k)
        assert ctx.count_buffer(text) == want, (variant, m, k)
    # unaligned text pointer -> register-staged tile kernel
    ctx.set_kernel("auto")
    ctx.set_patterns([pat], k)
    d = ctx.device_alloc(len(text) + 64)
    try:
        ctx.device_upload(d + 3, text)
        cnt = ctx.device_alloc(8)
        ctx.device_memset(cnt, 0, 8)
        ctx.count_shard_device(d + 3, 0, len(text), len(text), 0, len(text), cnt)
        ctx.synchronize()
        assert int.from_bytes(ctx.device_download(cnt, 8), "little") == want[0]
        ctx.device_free(cnt)
    finally:
        ctx.device_free(d)


@pytest.mark.parametrize("m", [16, 17, 19])
def test_window_nominated_only_by_a_shifted_piece_at_tile_start(ctx, apm, m):
    """k = 3, four pieces: the window = piece 0 with one byte deleted | piece 1 intact (hence found one position
    early) | piece 2 with a substitution | piece 3 with an inserted byte.  Only (piece 1, shift -1) can nominate
    it, through the BACKWARD partner pre-check, whose text bytes lie in front of the piece -- placed on the first
    window starts of several tiles, where those bytes sit at the very beginning of the staged tile."""
    k = 3
    rng = random.Random(m)
    pat = bytes(rng.choice(b"ACGT") for _ in range(m))
    a = [q * m // 4 for q in range(5)]                     # piece offsets
    pieces = [bytearray(pat[a[q]:a[q + 1]]) for q in range(4)]
    del pieces[0][1]
    other = {65: 67, 67: 71, 71: 84, 84: 65}
    pieces[2][1] = other[pieces[2][1]]
    pieces[3].insert(2, other[pieces[3][2]])
    w = bytes(pieces[0] + pieces[1] + pieces[2] + pieces[3])
    assert len(w) == m and H.window_distance(pat, w) == 3
    tile_w = (4096 - 16 - m - k // 2) & ~31
    text = bytearray(rng.choice(b"ACGT") for _ in range(tile_w * 12))
    planted = []
    for t in range(1, 11):
        pos = tile_w * t + (t % 5)                         # window starts 0..4 of tile t
        text[pos:pos + m] = w
        planted.append(pos)
    text = bytes(text)
    want = H.oracle_counts(text, [pat], k, banded=True)
    assert want[0] >= 10
    for variant in ("auto", "banded", "bitpar"):
        ctx.set_kernel(variant)
        ctx.set_patterns([pat], k)
        assert ctx.count_buffer(text) == want, (variant, m)
    ctx.set_kernel("auto")


@pytest.mark.parametrize("P,m,k", [(490, 50, 5), (800, 30, 3), (200, 16, 3), (600, 20, 3), (1000, 32, 0), (700, 64, 1)])
def test_large_pattern_sets_vs_full_dp(apm, P, m, k):
    """Hundreds of patterns: the verify launch's LDS image passes 64 KiB with its wave buffers (490 x 50, k = 5: 55 KB of
    image, 512-thread workgroups); dense key sets -- a fifth (800 x 30, k = 3), half (200 x 16) and two thirds (600 x 20:
    several verify launches) of all 16-bit code words set -- stay on the sieve pipeline (no density limit); long pieces
    with k <= 1 leave the stream kernel for the fused sampled pipeline once one stream launch cannot hold the set.
    AUTO == forced full-DP BITPAR on 8 MiB of random DNA with planted occurrences."""
    import torch
    rnd = random.Random(1000 * P + m)
    n = 8 << 20
    g = torch.Generator().manual_seed(P)
    host = torch.tensor(list(b"ACGT"), dtype=torch.uint8)[torch.randint(0, 4, (n,), generator=g)]
    tb = host.numpy().tobytes()
    pats = []
    for _ in range(P):
        o = rnd.randrange(0, n - m)
        p = bytearray(tb[o:o + m])
        for _e in range(rnd.randrange(0, max(k, 1))):
            p[rnd.randrange(m)] = rnd.choice(b"ACGT")
        pats.append(bytes(p))
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    text[:n] = host.to("cuda:0")
    cnt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c2:
        c2.set_patterns(pats, k)
        got = {}
        for variant in ("auto", "bitpar"):
            c2.set_kernel(variant)
            cnt.zero_()
            torch.cuda.synchronize()
            c2.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr())
            c2.synchronize()
            got[variant] = cnt.cpu().tolist()
            if variant == "auto":
                assert c2.stat("sieve_on") == 1 and c2.stat("sieve_stride") == (8 if k <= 1 else 1)
                if k <= 1 and "APM_FUSED" not in os.environ:
                    assert c2.stat("sieve_fused") == 1
                if P == 490:
                    assert c2.stat("verify_image_bytes") > 50000
                if P == 600:
                    assert c2.stat("verify_launches") >= 2 and c2.stat("sieve_rate") > 0.5
        assert got["auto"] == got["bitpar"]
        assert sum(got["auto"]) >= P


@pytest.mark.parametrize("seed", [301, 302, 303, 304, 305, 306])
def test_sieve_pipeline_vs_full_dp_at_scale(apm, seed):
    """Random pattern sets (lengths 12..128 mixed, k = 2..5, patterns cut from the text and edited) on 48 MiB of device
    text over small alphabets -- millions of sieve hits, thousands of verify batches per wave, real DP work and bursts
    of matches: the sieve + verify pipeline (AUTO/BANDED) must give the counts of the forced full-DP BITPAR kernel.
    One seed in three runs the sampled form (all pieces >= 15 bytes)."""
    import torch
    rnd = random.Random(seed)
    n = 48 << 20
    alpha = rnd.choice([b"ACGT", b"ACGT", b"ACG", b"ACGTN"])
    k = rnd.choice([2, 3, 3, 4, 5])
    sampled = seed % 3 == 0
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randint(0, len(alpha), (n,), generator=g, dtype=torch.uint8)
    lut = torch.tensor(list(alpha), dtype=torch.uint8)
    host = lut[idx.long()]
    if seed % 2 == 0:  # tandem repeats: stretches where every window is a candidate and matches come in bursts
        unit = host[:37].clone()
        for off in range(1 << 20, n - (1 << 16), 5 << 20):
            host[off:off + 37 * 800] = unit.repeat(800)
    text_bytes = host.numpy().tobytes()
    pats = []
    for i in range(rnd.randint(8, 40)):
        m = rnd.randint(15 * (k + 1), 128) if sampled else rnd.randint(4 * (k + 1), 128)
        o = rnd.randrange(0, n - m)
        p = bytearray(text_bytes[o:o + m])
        for _e in range(rnd.randint(0, k)):
            r, pos = rnd.random(), rnd.randrange(len(p))
            if r < 0.5:
                p[pos] = rnd.choice(alpha)
            elif r < 0.75 and len(p) > 1:
                del p[pos]
                p.append(rnd.choice(alpha))
            else:
                p.insert(pos, rnd.choice(alpha))
                p.pop()
        pats.append(bytes(p))
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    text[:n] = host.to("cuda:0")
    cnt = torch.zeros(len(pats), dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c2:
        c2.set_patterns(pats, k)
        assert c2.stat("sieve_on") == 1 and c2.stat("sieve_stride") == (8 if sampled else 1)
        got = {}
        for variant in ("auto", "bitpar"):
            c2.set_kernel(variant)
            cnt.zero_()
            torch.cuda.synchronize()
            c2.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr())
            c2.synchronize()
            got[variant] = cnt.cpu().tolist()
            if variant == "auto" and "APM_FUSED" not in os.environ:
                assert c2.stat("sieve_fused") == (1 if sampled else 0)   # sampled sets run sieve + verify as one kernel
        assert got["auto"] == got["bitpar"], (seed, k, sampled, [len(p) for p in pats])
        assert sum(got["auto"]) >= len(pats) // 2


def _soak_seeds(default):
    """APM_SOAK_SEEDS=a-b widens the soak tests for a bug hunt (the suite itself runs the default seeds)."""
    spec = os.environ.get("APM_SOAK_SEEDS")
    if not spec:
        return default
    lo, hi = spec.split("-")
    return list(range(int(lo), int(hi) + 1))


@pytest.mark.parametrize("seed", _soak_seeds([11, 12, 13]))
def test_banded_path_soak_vs_oracle(ctx, apm, seed):
    """Randomised soak of the BANDED path on texts spanning many tiles / chunks: small alphabets (many candidates
    and real DP work), patterns cut from the text and edited (substitutions, deletions, insertions), every k the
    path supports, aligned and unaligned device text, whole text and owner-computes shards."""
    rnd = random.Random(seed)
    for trial in range(50):
        alpha = rnd.choice([b"ACGT", b"ACGT", b"AC", b"ACG", b"ACGTN", b"acgtACGT", bytes(range(32, 48))])
        n = rnd.choice([5000, 12345, 20000, 33333, 50000, 81920])
        if rnd.random() < 0.3:  # repetitive text: tandem repeats stress the dedup of shifted nominations
            unit = bytes(rnd.choice(alpha) for _ in range(rnd.choice([3, 7, 19, 64])))
            text = bytearray((unit * (n // len(unit) + 1))[:n])
            for _ in range(n // 50):
                text[rnd.randrange(n)] = rnd.choice(alpha)
            text = bytes(text)
        else:
            text = bytes(rnd.choice(alpha) for _ in range(n))
        k = rnd.choice([0, 1, 1, 2, 2, 3, 3, 4, 5, 6, 7])
        pats = []
        for _ in range(rnd.randint(1, 8)):
            m = rnd.randint(4 * (k + 1), min(256, 40 * (k + 1)))
            o = rnd.randrange(0, n - m)
            p = bytearray(text[o:o + m])
            for _e in range(rnd.randint(0, k + 1)):
                r, pos = rnd.random(), rnd.randrange(len(p))
                if r < 0.4:
                    p[pos] = rnd.choice(alpha)
                elif r < 0.7 and len(p) > 1:
                    del p[pos]
                    p.append(rnd.choice(alpha))
                else:
                    p.insert(pos, rnd.choice(alpha))
                    p.pop()
            pats.append(bytes(p))
        want = H.oracle_counts(text, pats, k, banded=True)
        ctx.set_kernel("auto")
        ctx.set_patterns(pats, k)
        ctx.set_kernel("banded")
        assert all(ctx.pattern_kernel(i) == 4 for i in range(len(pats)))
        mode = trial % 3
        if mode == 0:
            got = ctx.count_buffer(text)
        else:  # device text at a random byte alignment, one shard or three owner-computes shards
            shift = rnd.randrange(16) if mode == 1 else 0
            d = ctx.device_alloc(n + 64)
            cnt = ctx.device_alloc(8 * len(pats))
            try:
                ctx.device_upload(d + shift, text)
                ctx.device_memset(cnt, 0, 8 * len(pats))
                cuts = [0, n] if mode == 1 else [0, rnd.randrange(1, n // 2), rnd.randrange(n // 2, n - 1), n]
                for lo, hi in zip(cuts, cuts[1:]):
                    end = min(n, hi + 255)
                    ctx.count_shard_device(d + shift + lo, lo, end - lo, n, lo, hi, cnt)
                ctx.synchronize()
                raw = ctx.device_download(cnt, 8 * len(pats))
                got = [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(len(pats))]
            finally:
                ctx.device_free(cnt)
                ctx.device_free(d)
        assert got == want, (seed, trial, mode, k, [len(p) for p in pats], alpha)
    ctx.set_kernel("auto")


@pytest.mark.parametrize("seed", _soak_seeds([21, 22]))
def test_all_kernels_soak_vs_oracle(ctx, apm, seed):
    """The same soak for AUTO (mixed kernels per pattern set, trivial and generic patterns included) and the
    forced full-DP kernels, on device text at random alignments and over owner-computes shards."""
    rnd = random.Random(seed)
    for trial in range(40):
        alpha = rnd.choice([b"ACGT", b"AC", b"ACGTN\n", bytes(range(256))])
        n = rnd.choice([300, 4095, 4097, 9000, 20011, 40000])
        text = bytes(rnd.choice(alpha) for _ in range(n))
        k = rnd.choice([0, 1, 2, 3, 5, 8])
        variant = rnd.choice(["auto", "auto", "bitpar", "wavefront", "generic"])
        pats = []
        for _ in range(rnd.randint(1, 6)):
            m = rnd.choice([1, 3, 8, 16, 17, 31, 32, 33, 50, 64, 65, 100, 128])
            if variant in ("auto", "generic", "wavefront") and rnd.random() < 0.2:
                m = rnd.choice([129, 200, 256])
            if variant in ("auto", "generic") and rnd.random() < 0.1:
                m = rnd.choice([257, 400])
            m = min(m, n // 2)
            o = rnd.randrange(0, n - m)
            p = bytearray(text[o:o + m])
            for _e in range(rnd.randint(0, k + 1)):
                r, pos = rnd.random(), rnd.randrange(len(p))
                if r < 0.4:
                    p[pos] = rnd.choice(alpha)
                elif r < 0.7 and len(p) > 1:
                    del p[pos]
                    p.append(rnd.choice(alpha))
                else:
                    p.insert(pos, rnd.choice(alpha))
                    p.pop()
            pats.append(bytes(p))
        pats = [p for p in pats if _supported(variant, p, k)]
        if not pats:
            continue
        want = H.oracle_counts(text, pats, k, banded=True)
        ctx.set_kernel("auto")      # (set_kernel re-plans the patterns already loaded: load the new ones first)
        ctx.set_patterns(pats, k)
        ctx.set_kernel(variant)
        m_max = max(len(p) for p in pats)
        shift = rnd.randrange(16)
        d = ctx.device_alloc(n + 64)
        cnt = ctx.device_alloc(8 * len(pats))
        try:
            ctx.device_upload(d + shift, text)
            ctx.device_memset(cnt, 0, 8 * len(pats))
            cuts = sorted({0, n} | {rnd.randrange(1, n) for _ in range(rnd.choice([0, 1, 3]))})
            for lo, hi in zip(cuts, cuts[1:]):
                end = min(n, hi + m_max - 1)
                ctx.count_shard_device(d + shift + lo, lo, end - lo, n, lo, hi, cnt)
            ctx.synchronize()
            raw = ctx.device_download(cnt, 8 * len(pats))
            got = [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(len(pats))]
        finally:
            ctx.device_free(cnt)
            ctx.device_free(d)
        assert got == want, (seed, trial, variant, k, shift, cuts, [len(p) for p in pats])
    ctx.set_kernel("auto")
