"""Measurement aid (GPU box): time the bench workload's scan kernel with stages skipped and a
plain read of the same buffer.  Uses the MEASUREMENT build of the library (make measure ->
libapm_hip_measure.so, -DAPM_MEASURE): the product library has no such switches.
Not part of the product or the test-suite.
ABLATIONS=a,b,.. = values of APM_MEASURE_SKIP to run (bits, sieve + verify pipeline): 1 the sieve reports no hits,
8 no nomination predicate (mask walk and window loads only), 16 the predicate runs but nothing survives, 32 no dedup
test, 64 no DP result, 256 collect the counters of tools/verify_stats.py (atomics: distorts the timing), 512 per-wave
time stamps, 1024 the predicate takes its partner text out of the window in hand (what the dependent gather costs:
cfg3 2 %, cfg5 20 % of the verify launch).  APM_VERIFY_GRID_PCT=p launches p % of the verify workgroups."""
import importlib, os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("APM_LIB_PATH", os.path.join(ROOT, "inf560-approximate-pattern-matching_amd", "libapm_hip_measure.so"))
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
wl = importlib.import_module("inf560-approximate-pattern-matching_amd.workloads")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = wl.CONFIGS[cfgname]
n = min(cfg["n"], 1 << 30)
k, lens, seed = cfg["k"], cfg["lens"], wl.seed_of(cfg["cid"])
pats, planted = wl.make_patterns(n, lens, k, seed)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = apm.ApmContext(device=0); ctx.set_stream(stream.cuda_stream)
ctx.set_patterns(pats, k)
if len(sys.argv) > 2: ctx.set_kernel(sys.argv[2])
text = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill_device(text.data_ptr(), 0, n, seed)
counts = torch.zeros(len(pats), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
def run(reps=20):
    ms = []
    for _ in range(reps):
        counts.zero_()
        ctx.count_shard_device(text.data_ptr(), 0, n, n, 0, n, counts.data_ptr())
        ms.append(ctx.timing()["main_kernel_ms"])
    ms = ms[2:]
    global last_launches
    last_launches = ctx.launch_times()
    return min(ms), sum(ms) / len(ms)
v = text[:n].view(torch.int64)
for _ in range(3): v.sum()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): v.sum()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print("torch int64 sum over the text: %.3f ms  -> %.0f GB/s" % (t, n / t / 1e6))
for ab in os.environ.get("ABLATIONS", "0,1,2,3").split(","):
    os.environ["APM_MEASURE_SKIP"] = ab
    mn, av = run()
    print("skip=%s  kernel min %.4f ms avg %.4f ms -> %.0f GB/s" % (ab, mn, av, n / mn / 1e6), "counts", counts.tolist()[:8],
          [(l, round(t, 4)) for l, t in last_launches])
