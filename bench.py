#!/usr/bin/env python3
"""bench.py -- the hot path (Levenshtein sliding-window DP + match count) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the rank's text shard (all patterns),
followed, for N > 1, by the RCCL all-reduce of the P partial counts.  Inputs are
resident in HBM before the timed region (device-side synthetic generator).
Workload at N = 1: BASELINE.json configs[1] (cfg2: 256 MB synthetic DNA, 8 patterns
of length 32, k = 0); for N > 1 the same per-GPU shard size (weak scaling): a text
of N x 256 MiB sharded by owner-computes ranges with an (m_max-1)-byte halo.

Prints ONE JSON line (rank 0).  `value` = algorithmic window-DP cells per second,
sum_p (n-k) * m_p^2 / wall, with bit-exact counts; `config.kernel` names the kernel
variant that produced it; `variants` reports every full-DP kernel separately (cells
really evaluated per second) so that the exact-shortcut number is never mistaken
for raw DP throughput.
"""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "inf560-approximate-pattern-matching_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz int32 lane-ops/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", help="workload: cfg2|cfg3|cfg4|cfg5 (per-GPU size = cfg n / its GPU count)")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--bytes-per-gpu", type=int, default=0, help="override the per-GPU text size")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    return ap.parse_args()


def cpu_baseline(apm, wl, pats, k, seed, gpu_slice_counts_fn):
    """The reference's sequential path timed on this box's host cores, on a bounded
    sample (first 1 MiB of the same synthetic text, all patterns): kind "reference" =
    oracle/_ref/apm_sequential (the reference's own sources compiled in the build
    container), else kind "port" = oracle/liboracle.so, 1 thread."""
    # ~8e9 DP cells (10-15 s on one core): 1 MiB for cfg2, proportionally less for heavier pattern sets
    per_pos = sum(len(p) ** 2 for p in pats)
    sample = max(4096, min(1 << 20, int(8.6e9 / per_pos) & ~4095))
    m_max = max(len(p) for p in pats)
    text = apm.synth_fill_host(0, sample + m_max - 1, seed)
    cells = float(sample) * sum(len(p) ** 2 for p in pats)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "apm_sequential")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    out = None
    if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
        with tempfile.NamedTemporaryFile(suffix=".fa", delete=False) as f:
            f.write(text[:sample])
            path = f.name
        try:
            r = subprocess.run([ref_bin, str(k), path] + [p.decode("latin-1") for p in pats],
                               capture_output=True, timeout=600)
            mt = re.search(rb"APM done in ([0-9.]+) s", r.stdout)
            if r.returncode == 0 and mt:
                secs = float(mt.group(1))
                n_pos = max(0, sample - k)
                out = dict(value=n_pos * sum(len(p) ** 2 for p in pats) / secs, unit="cells/s", cores=1,
                           kind="reference", seconds=secs,
                           sample="first %d bytes of the bench text (as a file), all %d patterns, k=%d, oracle/_ref/apm_sequential" % (sample, len(pats), k))
        finally:
            os.unlink(path)
    import helpers as H                       # the oracle = checker (allowed here: cpu_baseline leg)
    if out is None:
        t0 = time.time()
        for p in pats:
            H.oracle().oracle_count_range(text, len(text), p, len(p), k, 0, sample)
        secs = time.time() - t0
        out = dict(value=cells / secs, unit="cells/s", cores=1, kind="port", seconds=secs,
                   sample="first %d bytes of the bench text, all %d patterns, k=%d, oracle/liboracle.so literal DP" % (sample, len(pats), k))
    # checker: GPU counts on the same slice must equal the oracle's
    want = H.oracle_counts(text, pats, k, banded=True, j_end=sample)
    got = gpu_slice_counts_fn(text, sample)
    out["slice_counts_match_gpu"] = bool(got == want)
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    apm = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    sharding = importlib.import_module(PKG + ".sharding")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: this engine has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend=args.dist_backend)

    cfg = wl.CONFIGS[args.config]
    k, lens, seed = cfg["k"], cfg["lens"], wl.seed_of(cfg["cid"])
    cfg_gpus = 8 if args.config in ("cfg4", "cfg5") else 1
    per_gpu = args.bytes_per_gpu or cfg["n"] // cfg_gpus
    n_total = per_gpu * world                                   # weak scaling
    pats, planted = wl.make_patterns(n_total, lens, k, seed)
    P = len(pats)
    m_max = max(lens)

    stream = torch.cuda.Stream(device=dev)                      # one explicit HIP stream for everything
    torch.cuda.set_stream(stream)
    ctx = apm.ApmContext(device=dev_index)
    ctx.set_stream(stream.cuda_stream)                          # the library launches on torch's stream
    ctx.set_patterns(pats, k)
    ctx.set_kernel(args.kernel)
    kernel_names = sorted({apm.KERNEL_NAMES[ctx.pattern_kernel(i)] for i in range(P)})

    ob, oe, lo, hi = sharding.rank_shard(n_total, k, m_max, rank, world)
    text = torch.empty(hi - lo + 16, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(text.data_ptr(), lo, hi - lo, seed)   # inputs resident in HBM
    # ring of count vectors: the all-reduce of step i (RCCL's own stream) overlaps the scans of the next steps
    RING = 4
    ring = [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(RING)]
    pending = [None] * RING
    counts = ring[0]
    torch.cuda.synchronize()

    def step(i):
        b = i % RING
        if pending[b] is not None:
            pending[b].wait()                                   # buffer free again (stream-level wait)
            pending[b] = None
        c = ring[b]
        c.zero_()
        ctx.count_shard_device(text.data_ptr(), lo, hi - lo, n_total, ob, oe, c.data_ptr())
        if world > 1:
            if args.dist_backend == "nccl":
                pending[b] = dist.all_reduce(c, op=dist.ReduceOp.SUM, async_op=True)  # P x int64 over xGMI
            else:
                h = c.cpu()
                sharding.allreduce_counts(h)
                c.copy_(h)
        return c

    def drain():
        for b in range(RING):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    ctx.set_timing(False)       # no event records inside the timed region
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        counts = step(i)
    drain()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_counts = counts.cpu().tolist()

    # per-launch kernel duration with HIP events on the launch stream (the library brackets
    # its scan kernels with hipEventRecord on the same stream), averaged over `steps` launches
    ctx.set_timing(True)
    kms = []
    scratch = torch.zeros(P, dtype=torch.int64, device=dev)
    for _ in range(args.steps):
        scratch.zero_()
        ctx.count_shard_device(text.data_ptr(), lo, hi - lo, n_total, ob, oe, scratch.data_ptr())
        tm = ctx.timing()
        kms.append(tm["main_kernel_ms"])
    kernel_ms = sum(kms) / len(kms)
    shard_bytes = tm["text_bytes"]
    n_launch = tm["n_launches"]

    counts_ok = all(c >= (1 if d <= k else 0) for c, (_, d) in zip(final_counts, planted))
    if k == 0 and min(lens) >= 24:
        counts_ok = counts_ok and final_counts == wl.expected_counts_k0(n_total, pats, planted, seed)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    cells = wl.algorithmic_cells(n_total, lens, k)
    ms_per_step = elapsed * 1e3 / args.steps
    value = cells / (elapsed / args.steps)
    achieved_gbs = shard_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic = None   # PMC-measured HBM bytes per launch (separate rocprofv3 --pmc passes, profiles/traffic.json)
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            ent = json.load(f).get("%s:%s" % (args.config, "+".join(kernel_names)))
        if ent and world == 1 and not args.bytes_per_gpu:
            traffic = ent["traffic_bytes"]
    except Exception:
        traffic = None
    line = {
        "metric": "window-DP-cells/sec",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.config, cfg["desc"]), "text_bytes_total": n_total,
                   "text_bytes_per_gpu": per_gpu, "patterns": P, "pattern_len": sorted(set(lens)), "k": k,
                   "kernel": "+".join(kernel_names), "partition": "text-sharded x%d, halo m_max-1, RCCL all-reduce of counts" % world},
        "note": "value = algorithmic window-DP cells (sum_p (n-k)*m_p^2) per second with bit-exact counts; "
                "config.kernel names the kernel that produced it ('banded' = exact shortcut: pigeonhole key filter + "
                "banded verification, it does NOT evaluate every DP cell); raw full-DP throughput is under variants",
        "positions_x_patterns_per_s": float(max(0, n_total - k)) * P / (elapsed / args.steps),
        "counts": final_counts,
        "counts_exact_vs_planted": bool(counts_ok),
        "event_ms_per_step": ev_ms / args.steps,
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "+".join(kernel_names), "kernel_ms_avg": kernel_ms, "launches_per_step": n_launch,
                     "algorithmic_bytes_per_launch": shard_bytes,
                     "note": "1 HBM byte per text position per launch (SURVEY 8d); the full DP is integer-VALU bound, see variants[].valu_frac"},
    }

    # full-DP kernel variants, reported under their own label (cells really evaluated)
    if not args.no_variants and world == 1:
        variants = {}
        for name, ops_per_cell in (("wavefront", None), ("bitpar", None)):
            if max(lens) > {"wavefront": 256, "bitpar": 128}[name]:
                continue
            ctx.set_kernel(name)
            reps = 3
            ms = []
            for _ in range(reps + 1):
                scratch.zero_()
                ctx.count_shard_device(text.data_ptr(), lo, hi - lo, n_total, ob, oe, scratch.data_ptr())
                ms.append(ctx.timing()["main_kernel_ms"])
            ms = ms[1:]
            kms_v = sum(ms) / len(ms)
            ok = scratch.cpu().tolist() == final_counts
            variants[name] = {"cells_evaluated_per_s": cells / (kms_v * 1e-3), "kernel_ms": kms_v,
                              "counts_equal_headline": bool(ok),
                              "hbm_frac": shard_bytes / (kms_v * 1e-3) / 1e9 / HBM_PEAK_GBS}
        ctx.set_kernel(args.kernel)
        line["variants"] = variants

    if not args.no_cpu_baseline and world == 1:
        def gpu_slice(text_bytes, sample):
            t = torch.frombuffer(bytearray(text_bytes), dtype=torch.uint8).to(dev)
            c = torch.zeros(P, dtype=torch.int64, device=dev)
            ctx.count_shard_device(t.data_ptr(), 0, len(text_bytes), n_total, 0, sample, c.data_ptr())
            torch.cuda.synchronize()
            return c.cpu().tolist()
        line["cpu_baseline"] = cpu_baseline(apm, wl, pats, k, seed, gpu_slice)

    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
