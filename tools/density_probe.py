"""Measurement aid (GPU box): big pattern sets.  Scans 64 MiB of random DNA with P patterns of length m (k errors) through the
sieve pipeline (APM_SIEVE=1, the default) and through the tile / stream kernels alone (APM_SIEVE=0), and checks the counts
against each other.  Round 2 took the density limit off the pipeline on the strength of these numbers."""
import importlib, os, sys, random, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import torch
    apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
    P, m, k = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rnd = random.Random(7)
    n = 64 << 20
    g = torch.Generator().manual_seed(1)
    host = torch.tensor(list(b"ACGT"), dtype=torch.uint8)[torch.randint(0, 4, (n,), generator=g)]
    tb = host.numpy().tobytes()
    pats = []
    for _ in range(P):
        o = rnd.randrange(0, n - m)
        p = bytearray(tb[o:o + m])
        for _e in range(rnd.randrange(0, k + 1)):
            p[rnd.randrange(m)] = rnd.choice(b"ACGT")
        pats.append(bytes(p))
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0"); text[:n] = host.to("cuda:0")
    cnt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c:
        c.set_patterns(pats, k)
        for rep in range(3):
            cnt.zero_(); torch.cuda.synchronize()
            c.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr()); c.synchronize()
        lt = c.launch_times()
        print("P=%d m=%d k=%d APM_SIEVE=%s sieve_on=%d rate=%.4f weak=%.2f  %.3f ms per 64 MiB  %s  sum=%d crc=%d" % (P, m, k, os.environ.get("APM_SIEVE", "1"), c.stat("sieve_on"), c.stat("sieve_rate"), c.stat("sieve_weak_frac"),
              sum(t for _, t in lt), [(l, round(t, 3)) for l, t in lt][:6], int(cnt.sum()), int((cnt * torch.arange(1, P + 1, device=cnt.device)).sum() % 1000003)))
else:
    for P, m, k in ((800, 30, 3), (60, 16, 3), (200, 16, 3), (600, 20, 3), (2000, 50, 5), (1000, 64, 2), (4000, 24, 2)):
        for sv in ("1", "0"):
            subprocess.run([sys.executable, __file__, "worker", str(P), str(m), str(k)], env=dict(os.environ, APM_SIEVE=sv))
