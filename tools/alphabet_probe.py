"""Measurement aid (GPU box): the BASELINE cfg3 shape (32 patterns of 16..128 bytes, k = 3) on texts that are not uniform DNA:
skewed DNA (60 % A/T), English-like letters (Zipf), uniform bytes.  256 MiB each; ms per GiB and candidates."""
import importlib, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
n = 256 << 20
rnd = random.Random(11)
def make(kind):
    g = torch.Generator().manual_seed(5)
    if kind == "dna_skew":
        w = torch.tensor([0.35, 0.15, 0.15, 0.35]); sym = torch.tensor(list(b"ACGT"), dtype=torch.uint8)
    elif kind == "prose":
        letters = b" etaoinshrdlcumwfgypbvkjxqz.,\n"
        w = torch.tensor([1.0 / (i + 1) for i in range(len(letters))]); sym = torch.tensor(list(letters), dtype=torch.uint8)
    else:
        w = torch.ones(256); sym = torch.arange(256, dtype=torch.uint8)
    idx = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
    return sym[idx]
for kind in ("dna_skew", "prose", "bytes"):
    host = make(kind)
    tb = host.numpy().tobytes()
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0"); text[:n] = host.to("cuda:0")
    pats = []
    for i in range(32):
        m = 16 + (128 - 16) * i // 31
        o = rnd.randrange(0, n - m)
        p = bytearray(tb[o:o + m])
        for _e in range(rnd.randrange(0, 4)):
            p[rnd.randrange(m)] = tb[rnd.randrange(n)]
        pats.append(bytes(p))
    cnt = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    with apm.ApmContext(device=0) as c:
        c.set_patterns(pats, 3)
        for rep in range(3):
            cnt.zero_(); torch.cuda.synchronize()
            c.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr()); c.synchronize()
        lt = c.launch_times()
        auto = cnt.cpu().tolist()
        cand = c.stat("sieve_candidates")
        c.set_kernel("bitpar"); cnt.zero_(); torch.cuda.synchronize()
        c.count_shard_device(text.data_ptr(), 0, n, n, 0, n, cnt.data_ptr()); c.synchronize()
        print("%-9s %.3f ms per GiB  %s  candidates per GiB %.1fM  rate %.4f  matches %d  equal_bitpar %s" % (kind, 4 * sum(t for _, t in lt),
              [(l, round(t, 3)) for l, t in lt], 4 * cand / 1e6, c.stat("sieve_rate"), sum(auto), auto == cnt.cpu().tolist()), flush=True)
    del text
