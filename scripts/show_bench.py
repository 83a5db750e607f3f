"""Digest of a bench.py JSON line: usage show_bench.py file"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4))
if "config3" in d:
    print("config3", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["config3"].items() if k in ("value", "ms_per_step", "individuals_per_rank")})
for k in ("cold_structure_evals_per_s", "threaded_b1_evals_per_s"):
    if k in d:
        print(k, round(d[k]))
if "sampler_branch" in d:
    print("sampler", round(d["sampler_branch"]["value"]))
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "avg_launch_us", "traffic")})
for k in r.get("kernels", []):
    print("  ", {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in k.items() if kk in ("kernel", "launches", "avg_launch_us", "achieved", "frac", "states_per_launch")})
print("diff vs oracle", d.get("max_abs_diff_vs_cpu_oracle"), "cpu", d.get("cpu_baseline", {}).get("value"))
