"""Copy the summaries of a profiles/run_profiles.sh run (gpurun_out/prof_<tag>/<cfg>/) into profiles/<round>/ under a
prefix, and recompute profiles/traffic.json for the bench's dominant kernel of each workload (measurement
bookkeeping, run in the build container).
    python tools/refresh_profiles.py <tag> <round> <prefix> [--traffic]
HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (MI355X_MICROARCH.md, HBM: gfx950 reports half of a
wide coalesced read; separate --pmc passes)."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
G, P = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)


def parse_pmc(path):
    """{kernel name: {counter: mean}}"""
    out, cur = {}, None
    for line in open(path):
        if not line.startswith("   "):
            cur = line.split(" | ")[0].strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=(\d+) mean=(\S+)", line)
            out[cur][m.group(1)] = (float(m.group(3)), int(m.group(2)))
    return out


traffic_path = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(traffic_path))
for cfg in sorted(os.listdir(G)):
    d = os.path.join(G, cfg)
    for fn in ("kernel_stats.csv", "pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt", "pmc_SQ.txt", "bench.json"):
        if os.path.exists(os.path.join(d, fn)):
            shutil.copyfile(os.path.join(d, fn), os.path.join(P, "%s_%s_%s" % (prefix, cfg, fn)))
    if "--traffic" not in sys.argv:
        continue
    bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
    f, w = parse_pmc(os.path.join(d, "pmc_FETCH_SIZE.txt")), parse_pmc(os.path.join(d, "pmc_WRITE_SIZE.txt"))
    per_kernel = {}
    for name in f:
        if "apm_" not in name or "synth" in name:
            continue
        fs, n = f[name]["FETCH_SIZE"]
        ws = w.get(name, {}).get("WRITE_SIZE", (0.0, n))[0]
        per_kernel[name] = dict(launches_in_run=n, fetch_size_kb=fs, write_size_kb=ws, bytes_per_launch=int(round(2 * fs * 1024 + ws * 1024)))
    # the step's dominant kernel = the one with the largest average duration in the kernel trace
    import csv
    rows = [r for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))) if "apm_" in r["Name"] and "synth" not in r["Name"]]
    dom = max(rows, key=lambda r: float(r["AverageNs"]))["Name"]
    dom_key = next((k for k in per_kernel if dom.startswith(k) or k.startswith(dom[:60])), None)
    steps = max(v["launches_in_run"] for v in per_kernel.values())      # passes over the shard in the run = launches of a kernel every step has
    total = sum(v["bytes_per_launch"] * v["launches_in_run"] for v in per_kernel.values()) / steps
    traffic["%s:%s" % (cfg, bench["config"]["kernel"])] = dict(
        round=rnd, kernel=dom, traffic_bytes=per_kernel[dom_key]["bytes_per_launch"] if dom_key else None,
        step_traffic_bytes=int(total), algorithmic_bytes=bench["roofline"].get("algorithmic_bytes_per_step", bench["roofline"].get("algorithmic_bytes_per_launch")), per_kernel=per_kernel)
    print(cfg, dom[:50], per_kernel.get(dom_key, {}).get("bytes_per_launch"), "step", int(total))
json.dump(traffic, open(traffic_path, "w"), indent=1)
