"""Measurement aid (GPU box, product build): what the sieve hands to the verify launch -- hits that survive its code
filter (APM_SIEVE_CF=0: every lookup hit) -- and the launch times, for a BASELINE workload.  usage: cf_stats.py cfg3|cfg5 ..."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
wl = importlib.import_module("inf560-approximate-pattern-matching_amd.workloads")
for name in [a for a in sys.argv[1:]]:
    cfg = wl.CONFIGS[name]
    n = min(cfg["n"], 1 << 30)
    k, lens, seed = cfg["k"], cfg["lens"], wl.seed_of(cfg["cid"])
    pats, planted = wl.make_patterns(n, lens, k, seed)
    ctx = apm.ApmContext(device=0)
    ctx.set_patterns(pats, k)
    text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    counts = torch.zeros(len(pats), dtype=torch.int64, device="cuda:0")
    ctx.synth_fill_device(text.data_ptr(), 0, n, seed)
    ctx.synchronize(); torch.cuda.synchronize()
    best = None
    for _ in range(6):
        counts.zero_(); torch.cuda.synchronize()
        ctx.count_shard_device(text.data_ptr(), 0, n, n, 0, n, counts.data_ptr())
        ctx.synchronize()
        lt = ctx.launch_times()
        tot = sum(t for _, t in lt)
        if best is None or tot < best[0]:
            best = (tot, lt)
    print(name, "APM_SIEVE_CF=%s" % os.environ.get("APM_SIEVE_CF", "1"),
          {key: ctx.stat(key) for key in ("sieve_rate", "sieve_weak_frac", "sieve_cf", "sieve_cf_bytes", "sieve_candidates", "verify_image_bytes", "verify_blocks_per_cu", "verify_threads")},
          "best of 6: %.4f ms" % best[0], [(l, round(t, 4)) for l, t in best[1]], "sum(counts)=%d" % int(counts.sum().item()))
    del ctx, text
