"""Measurement aid (GPU box, measurement build): sets of long pieces with k <= 1 -- stream kernel (the product's choice) or
the fused sampled pipeline (APM_SAMPLED_MIN_K=0)?  256 MiB of random DNA."""
import importlib, os, sys, random, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    os.environ.setdefault("APM_LIB_PATH", os.path.join(ROOT, "inf560-approximate-pattern-matching_amd", "libapm_hip_measure.so"))
    import torch
    apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
    P, m, k = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rnd = random.Random(7)
    n = 256 << 20
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
        for rep in range(4):
            cnt.zero_(); torch.cuda.synchronize()
            c.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr()); c.synchronize()
        lt = c.launch_times()
        print("P=%d m=%d k=%d min_k=%s fused=%d  %.3f ms per 256 MiB  %s  sum=%d crc=%d" % (P, m, k, os.environ.get("APM_SAMPLED_MIN_K"), c.stat("sieve_fused"),
              sum(t for _, t in lt), [(l, round(t, 3)) for l, t in lt][:6], int(cnt.sum()), int((cnt * torch.arange(1, P + 1, device=cnt.device)).sum() % 1000003)))
else:
    for P, m, k in ((8, 32, 0), (32, 32, 0), (128, 32, 0), (1000, 32, 0), (8, 64, 1), (64, 64, 1), (1000, 64, 1)):
        for mk in ("2", "0"):
            subprocess.run([sys.executable, __file__, "worker", str(P), str(m), str(k)], env=dict(os.environ, APM_SAMPLED_MIN_K=mk))
