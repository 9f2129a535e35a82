"""Print the per-workload summary of a bench.py JSON line (measurement aid)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(name, v):
    r = v["roofline"]
    dk = r["dominant_kernel"]
    print("%s  %.4f ms/step  step frac %.3f (kernels %.4f ms)  dom %s %.4f ms alone %.3f  eq_bitpar %s" % (name, v["ms_per_step"], r["frac"], r["step_kernel_ms"], dk["kernel"], dk["ms_avg"], dk["frac_of_peak_alone"], v.get("counts_equal_bitpar")))
    print("     ", [(x["kernel"], round(x["ms_avg"], 4)) for x in r["launches"]])
    if "sieve" in v:
        print("     ", {k: (round(x, 5) if x < 10 else int(x)) for k, x in v["sieve"].items()})
show(d["config"]["workload"][:4], d)
for k, v in d.get("per_config", {}).items():
    show(k, v)
