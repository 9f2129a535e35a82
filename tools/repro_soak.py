"""Replay one trial of tests/test_gpu_parity.py::test_banded_path_soak_vs_oracle (debug aid)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
apm = H.pkg()
seed, want_trial = int(sys.argv[1]), int(sys.argv[2])
force_shift = int(sys.argv[3]) if len(sys.argv) > 3 else None
rnd = random.Random(seed)
for trial in range(want_trial + 1):
    alpha = rnd.choice([b"ACGT", b"ACGT", b"AC", b"ACG", b"ACGTN", b"acgtACGT", bytes(range(32, 48))])
    n = rnd.choice([5000, 12345, 20000, 33333, 50000, 81920])
    if rnd.random() < 0.3:
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
                del p[pos]; p.append(rnd.choice(alpha))
            else:
                p.insert(pos, rnd.choice(alpha)); p.pop()
        pats.append(bytes(p))
    mode = trial % 3
    shift = rnd.randrange(16) if mode == 1 else 0
    if mode == 2:
        rnd.randrange(1, n // 2); rnd.randrange(n // 2, n - 1)
print("n", n, "k", k, "mode", mode, "shift", shift, "lens", [len(p) for p in pats], "alpha", alpha, file=sys.stderr)
want = H.oracle_counts(text, pats, k, banded=True)
full = H.oracle_counts(text, pats, k)
ctx = apm.ApmContext(device=0)
for kern in ("banded", "bitpar"):
    ctx.set_kernel(kern)
    ctx.set_patterns(pats, k)
    for sh in ([force_shift] if force_shift is not None else [0, shift]):
        d = ctx.device_alloc(n + 64); cnt = ctx.device_alloc(8 * len(pats))
        ctx.device_upload(d + sh, text); ctx.device_memset(cnt, 0, 8 * len(pats))
        ctx.count_shard_device(d + sh, 0, n, n, 0, n, cnt); ctx.synchronize()
        raw = ctx.device_download(cnt, 8 * len(pats))
        got = [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(len(pats))]
        print(kern, "shift", sh, "OK" if got == want else "MISMATCH", [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w], "oracle banded==full:", want == full)
        ctx.device_free(cnt); ctx.device_free(d)
# single patterns alone, to see whether the miss depends on the launch mates
ctx.set_kernel("banded")
for i, p in enumerate(pats):
    ctx.set_patterns([p], k)
    g = ctx.count_buffer(text)[0]
    if g != want[i]:
        print("alone: pattern", i, "len", len(p), "got", g, "want", want[i])

# localise the missed windows of one pattern on the unaligned buffer
def shard_count(ctx, d, sh, lo, hi, npat):
    cnt = ctx.device_alloc(8 * npat); ctx.device_memset(cnt, 0, 8 * npat)
    end = min(n, hi + 255)
    ctx.count_shard_device(d + sh + lo, lo, end - lo, n, lo, hi, cnt); ctx.synchronize()
    raw = ctx.device_download(cnt, 8 * npat); ctx.device_free(cnt)
    return [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(npat)]
for pi in (7, 1):
    p = pats[pi]
    ctx.set_kernel("banded"); ctx.set_patterns([p], k)
    d = ctx.device_alloc(n + 64); ctx.device_upload(d + shift, text)
    whole = shard_count(ctx, d, shift, 0, n, 1)[0]
    print("pattern", pi, "len", len(p), "whole-buffer (offset 0 shard) got", whole, "want", want[pi])
    # the shard API moves the text pointer, so instead bisect with own ranges over the SAME text pointer
    def own_count(lo, hi):
        cnt = ctx.device_alloc(8); ctx.device_memset(cnt, 0, 8)
        ctx.count_shard_device(d + shift, 0, n, n, lo, hi, cnt); ctx.synchronize()
        v = int.from_bytes(ctx.device_download(cnt, 8), "little"); ctx.device_free(cnt); return v
    bad = [(0, n)]
    for _ in range(20):
        nxt = []
        for lo, hi in bad:
            if hi - lo <= 1: nxt.append((lo, hi)); continue
            mid = (lo + hi) // 2
            for a, b in ((lo, mid), (mid, hi)):
                if own_count(a, b) != H.oracle_counts(text, [p], k, banded=True, j_begin=a, j_end=b)[0]: nxt.append((a, b))
        bad = nxt[:8]
    print("  missed window starts:", [lo for lo, hi in bad], "tile_w", (4096 - 16 - len(p) - k // 2) & ~31)
    ctx.device_free(d)
