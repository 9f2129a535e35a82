"""Copy the summaries of the last gpurun measurement into profiles/<round>/ and recompute
profiles/traffic.json (measurement bookkeeping, run in the build container)."""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)
pairs = [("bench_cfg2.json", "bench_cfg2_full.json"), ("prof_cfg2/trace_kernel_stats.csv", "bench_cfg2_kernel_stats.csv"),
         ("pmc_fetch/pmc_counter_collection.csv", "bench_cfg2_pmc_FETCH_SIZE.csv"),
         ("pmc_write/pmc_counter_collection.csv", "bench_cfg2_pmc_WRITE_SIZE.csv"),
         ("bench_cfg3.json", "bench_cfg3_1GiB.json"), ("bench_cfg4.json", "bench_cfg4_pergpu_1GiB.json"),
         ("bench_cfg5.json", "bench_cfg5_pergpu_1GiB.json"),
         ("trace_cfg3/t_kernel_stats.csv", "tool_cfg3_1GiB_kernel_stats.csv"),
         ("trace_cfg4/t_kernel_stats.csv", "tool_cfg4_1GiB_kernel_stats.csv"),
         ("trace_cfg5/t_kernel_stats.csv", "tool_cfg5_1GiB_kernel_stats.csv")]
for src, dst in pairs:
    if os.path.exists(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
        print("copied", src, "->", dst)

def mean_counter(path, counter, kernel_prefix):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
         if r["Counter_Name"] == counter and kernel_prefix in r["Kernel_Name"]]
    return sum(v) / len(v) if v else None

kern = "apm_stream_kernel<0, 16, 16>"
f = mean_counter(os.path.join(P, "bench_cfg2_pmc_FETCH_SIZE.csv"), "FETCH_SIZE", kern)
w = mean_counter(os.path.join(P, "bench_cfg2_pmc_WRITE_SIZE.csv"), "WRITE_SIZE", kern)
tpath = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tpath))
t["cfg2:banded"] = {"fetch_size_kb": f, "write_size_kb": w, "traffic_bytes": int(round(2 * f * 1024 + w * 1024)),
                    "round": rnd, "kernel": "apm_stream_kernel<0,16,16>"}
json.dump(t, open(tpath, "w"), indent=1)
print(t["cfg2:banded"])
