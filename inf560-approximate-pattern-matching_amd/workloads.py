"""Synthetic workloads of BASELINE.json `configs`.

Text: counter-based DNA, byte i = "ACGT"[(splitmix64(seed ^ (i>>5)) >> (2*(i&31))) & 3]
(generated on the device by apm_synth_fill_device / on the host by
apm_synth_fill_host -- same bytes).  Patterns: pattern_p = text[o_p : o_p+m_p],
o_p = floor((p + 1/2) * n / P), then d_p = p mod (k+2) substitutions at
positions (7t+3) mod m_p (t < d_p) to the next base of "ACGT": some patterns
occur exactly, some within distance k, some (d_p = k+1) usually not at all.
"""
from . import synth_fill_host  # noqa: E402  (package-relative; see __init__)

SEED_BASE = 0x5EED0000

CONFIGS = {
    # name: n bytes (whole text), pattern lengths, k, config id
    "cfg2": dict(n=1 << 28, lens=[32] * 8, k=0, cid=2,
                 desc="256 MB synthetic DNA text, 8 patterns of len 32, k=0"),
    "cfg3": dict(n=1 << 30, lens=[16 + round(112 * i / 31) for i in range(32)], k=3, cid=3,
                 desc="1 GB synthetic DNA text, 32 patterns of mixed len 16-128, k=3"),
    "cfg4": dict(n=1 << 33, lens=[64] * 16, k=2, cid=4,
                 desc="8 GB synthetic text, 16 patterns len 64, k=2 (8 GPUs, text-sharded)"),
    "cfg5": dict(n=1 << 33, lens=[50] * 256, k=5, cid=5,
                 desc="8 GB text, 256 patterns len 50, k=5 (8 GPUs, pattern-batched)"),
}

_NEXT = {ord("A"): ord("C"), ord("C"): ord("G"), ord("G"): ord("T"), ord("T"): ord("A")}


def seed_of(cid):
    return SEED_BASE + cid


def make_patterns(n, lens, k, seed):
    """Returns (patterns, planted) where planted[p] = (offset o_p, substitutions d_p)."""
    P = len(lens)
    pats, planted = [], []
    for p, m in enumerate(lens):
        o = ((2 * p + 1) * n) // (2 * P)
        if o + m > n:
            o = max(0, n - m)
        b = bytearray(synth_fill_host(o, m, seed))
        d = p % (k + 2)
        for t in range(d):
            pos = (7 * t + 3) % m
            b[pos] = _NEXT[b[pos]]
        pats.append(bytes(b))
        planted.append((o, d))
    return pats, planted


def scaled(cfg, n):
    """The same workload on a text of n bytes (parity tests use small n)."""
    c = dict(CONFIGS[cfg])
    c["n"] = n
    return c


def algorithmic_cells(n, lens, k):
    """BASELINE metric numerator: sum_p (n-k) * m_p^2 (full-size windows; the <m truncated
    tail windows are counted at m^2 too -- relative error < 1e-6 at these sizes)."""
    pos = max(0, n - k)
    return float(pos) * float(sum(m * m for m in lens))


def expected_counts_k0(n, pats, planted, seed):
    """Exact expected counts for k = 0 on the synthetic text, valid while no pattern recurs by
    chance (m >= 24 on <= 2^40 bytes): the planted copy (if unmutated) plus the reference's
    truncated tail windows (sequential.c:131-134): a window of size s < m at the very end of the
    text matches when the last s text bytes equal the pattern's first s bytes."""
    m_max = max(len(p) for p in pats)
    tail = synth_fill_host(max(0, n - m_max), min(n, m_max), seed)
    out = []
    for p, (o, d) in zip(pats, planted):
        c = 1 if d == 0 else 0
        for s in range(1, min(len(p), n + 1)):
            if s < len(p) and tail[len(tail) - s:] == p[:s]:
                c += 1
        out.append(c)
    return out
