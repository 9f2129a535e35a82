"""Measurement aid (GPU box): GPU time of the sieve + verify launches when a 1 GiB shard is scanned in S slices one after
the other on one stream (slices small enough for the verify launch to find its text in the Infinity Cache?).  Sums the
per-launch event times (host overhead of the extra calls does not enter).  usage: slice_probe.py cfg3|cfg5 [S ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
wl = importlib.import_module("inf560-approximate-pattern-matching_amd.workloads")
cfg = wl.CONFIGS[sys.argv[1]]
n = 1 << 30
k, lens, seed = cfg["k"], cfg["lens"], wl.seed_of(cfg["cid"])
pats, planted = wl.make_patterns(n, lens, k, seed)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = apm.ApmContext(device=0); ctx.set_stream(stream.cuda_stream); ctx.set_patterns(pats, k)
text = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill_device(text.data_ptr(), 0, n, seed)
counts = torch.zeros(len(pats), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ref = None
for S in [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]:
    cuts = [((n * i // S) & ~15) for i in range(S)] + [n]
    best = None
    for rep in range(8):
        counts.zero_(); torch.cuda.synchronize()
        tot = {}
        for i in range(S):
            ctx.count_shard_device(text.data_ptr(), 0, n, n, cuts[i], cuts[i + 1], counts.data_ptr())
            for label, ms in ctx.launch_times():
                tot[label] = tot.get(label, 0.0) + ms
        torch.cuda.synchronize()
        if rep >= 2 and (best is None or sum(tot.values()) < sum(best.values())): best = tot
    c = counts.cpu().tolist()
    if ref is None: ref = c
    print("S=%d  %s  sum %.4f ms  counts_equal_S1 %s" % (S, {k: round(v, 4) for k, v in best.items()}, sum(best.values()), c == ref))
