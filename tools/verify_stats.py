"""Measurement aid (GPU box, measurement build): candidate / pre-check / survivor / DP counts of the sieve + verify
pipeline for a BASELINE workload.  usage: verify_stats.py cfg3|cfg5"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("APM_LIB_PATH", os.path.join(ROOT, "inf560-approximate-pattern-matching_amd", "libapm_hip_measure.so"))
os.environ.setdefault("APM_MEASURE_SKIP", "256")   # bit 8: collect the counters (slow: atomics); bit 9 (512): per-wave time stamps instead
import torch
apm = importlib.import_module("inf560-approximate-pattern-matching_amd")
wl = importlib.import_module("inf560-approximate-pattern-matching_amd.workloads")
cfg = wl.CONFIGS[sys.argv[1]]
n = min(cfg["n"], 1 << 30)
k, lens, seed = cfg["k"], cfg["lens"], wl.seed_of(cfg["cid"])
pats, planted = wl.make_patterns(n, lens, k, seed)
ctx = apm.ApmContext(device=0)
ctx.set_patterns(pats, k)
text = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
counts = torch.zeros(len(pats), dtype=torch.int64, device="cuda:0")
ctx.synth_fill_device(text.data_ptr(), 0, n, seed)
ctx.synchronize(); torch.cuda.synchronize()
for _ in range(2):
    counts.zero_(); torch.cuda.synchronize()
    ctx.count_shard_device(text.data_ptr(), 0, n, n, 0, n, counts.data_ptr())
    ctx.synchronize()
print(sys.argv[1], {key: ctx.stat(key) for key in ("sieve_rate", "sieve_candidates", "verify_survivors", "verify_dp_items",
                                                    "verify_counted", "verify_image_bytes", "verify_blocks_per_cu")}, ctx.launch_times())
if int(os.environ["APM_MEASURE_SKIP"]) & 512:   # per-wave time stamps (us from the first wave's start)
    print("waves", {key: round(ctx.stat("verify_wave_" + key), 1) for key in ("count", "start_max", "end_min", "end_p10", "end_p50", "end_p90", "end_max")})
    print("group end max", [round(ctx.stat("verify_wave_grpmax%d" % g)) for g in range(32)])
    print("group end min", [round(ctx.stat("verify_wave_grpmin%d" % g)) for g in range(32)])
    print("blockIdx%8 end avg", [round(ctx.stat("verify_wave_xcdavg%d" % x)) for x in range(8)])
