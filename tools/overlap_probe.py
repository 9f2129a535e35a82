"""Measurement aid (GPU box): does slicing a shard and running the slices' sieve + verify launches on two streams
(sieve of slice i+1 beside verify of slice i, slices small enough to stay in the Infinity Cache) pay?
usage: overlap_probe.py cfg3|cfg5 [slices ...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
wl = importlib.import_module("inf560-approximate-pattern-matching_amd.workloads")
cfg = wl.CONFIGS[sys.argv[1]]
n = 1 << 30
k, lens, seed = cfg["k"], cfg["lens"], wl.seed_of(cfg["cid"])
pats, planted = wl.make_patterns(n, lens, k, seed)
m_max = max(lens)
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
ctxs = []
for s in streams:
    c = apm.ApmContext(device=0)
    c.set_stream(s.cuda_stream)
    c.set_patterns(pats, k)
    c.set_timing(False)
    ctxs.append(c)
text = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctxs[0].synth_fill_device(text.data_ptr(), 0, n, seed)
counts = torch.zeros(len(pats), dtype=torch.int64, device=dev)
torch.cuda.synchronize()

def run(n_slices, n_lanes, reps=12):
    cuts = [((n * i // n_slices) & ~15) for i in range(n_slices)] + [n]
    best = 1e9
    for r in range(reps):
        counts.zero_()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        for lane in range(1, n_lanes):
            streams[lane].wait_event(e0)
        for i in range(n_slices):
            lo, hi = cuts[i], cuts[i + 1]
            end = min(n, hi + m_max - 1)
            lane = i % n_lanes
            ctxs[lane].count_shard_device(text.data_ptr() + lo, lo, end - lo, n, lo, hi, counts.data_ptr())
        for lane in range(1, n_lanes):
            ev = torch.cuda.Event()
            ev.record(streams[lane])
            streams[0].wait_event(ev)
        e1.record(streams[0])
        torch.cuda.synchronize()
        if r >= 2:
            best = min(best, e0.elapsed_time(e1))
    return best, counts.tolist()[:6]

base = None
for spec in (sys.argv[2:] or ["1x1", "4x1", "8x1", "4x2", "8x2", "16x2", "8x3", "16x3"]):
    ns, nl = (int(x) for x in spec.split("x"))
    t, c = run(ns, nl)
    if base is None:
        base = c
    print("%2d slices on %d stream(s): %.4f ms per GiB   counts ok: %s" % (ns, nl, t, c == base), flush=True)
