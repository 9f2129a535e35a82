"""Measurement aid (GPU box): scan time of pattern-set shapes outside the BASELINE configs (256 MiB of random DNA, planted
occurrences), to spot performance cliffs of the AUTO routing.  Prints ms per GiB and the launches."""
import importlib, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
n = (int(os.environ.get('SHAPE_MIB', '256'))) << 20
g = torch.Generator().manual_seed(1)
host = torch.tensor(list(b"ACGT"), dtype=torch.uint8)[torch.randint(0, 4, (n,), generator=g)]
tb = host.numpy().tobytes()
text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0"); text[:n] = host.to("cuda:0")
rnd = random.Random(3)
shapes = [(32, 16, 30, 1), (32, 16, 128, 1), (32, 24, 64, 2), (32, 64, 128, 5), (32, 64, 128, 6), (32, 64, 128, 7), (32, 40, 128, 4),
          (32, 129, 256, 3), (32, 129, 256, 7), (32, 12, 15, 3), (256, 100, 100, 3), (64, 16, 16, 2)]
if os.environ.get("SHAPE_SET") == "sparse":  # few candidates, long DPs
    shapes = [(32, 64, 128, 5), (32, 40, 128, 4), (32, 24, 64, 2)]
if os.environ.get("SHAPE_SET") == "long":   # long, loose patterns: full-DP kernels (use SHAPE_MIB=16)
    shapes = [(8, 200, 256, 40), (4, 300, 400, 3), (4, 300, 500, 100), (2, 600, 700, 10), (2, 900, 1024, 10), (2, 1300, 1300, 5), (1, 2048, 2048, 10), (1, 4096, 4096, 10)]
forced = os.environ.get("SHAPE_KERNEL")
for P, m0, m1, k in shapes:
    pats = []
    for i in range(P):
        m = m0 + (m1 - m0) * i // max(P - 1, 1)
        o = rnd.randrange(0, n - m)
        p = bytearray(tb[o:o + m])
        for _e in range(rnd.randrange(0, k + 1)):
            p[rnd.randrange(m)] = rnd.choice(b"ACGT")
        pats.append(bytes(p))
    cnt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c:
        if forced: c.set_kernel(forced)
        c.set_patterns(pats, k)
        for rep in range(3):
            cnt.zero_(); torch.cuda.synchronize()
            c.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr()); c.synchronize()
        lt = c.launch_times()
        kinds = sorted(set(c.pattern_kernel(i) for i in range(P)))
        print("MiB=%d P=%d m=%d..%d k=%d kernels=%s  %.3f ms per GiB  %s  matches=%d" % (n >> 20, P, m0, m1, k, kinds, (1 << 30) / n * sum(t for _, t in lt),
              [(l, round(t, 3)) for l, t in lt][:6], int(cnt.sum())), flush=True)
