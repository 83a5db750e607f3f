"""Condense scripts/profile_rows.sh output: per row the kernel statistics of rocprofv3 --kernel-trace --stats next to the
bench's own HIP-event launch times, and the HBM traffic per launch of the gate-pass kernel's two instantiations
(2 x FETCH_SIZE + WRITE_SIZE, KiB: the gfx950 correction of MI355X_MICROARCH.md).  Writes <dir>/traffic.json in the shape
bench.py's load_traffic() reads from profiles/r03_traffic.json."""
import csv
import glob
import json
import sys

root, rows = sys.argv[1], sys.argv[2:]
traffic = {}
for row in rows:
    print(f"== {row}: rocprofv3 --kernel-trace --stats -- python3 bench.py --only {row} ==")
    stats = {}
    for path in glob.glob(f"{root}/{row}_trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            stats[r["Name"]] = r
            print(f"  {r['Name'][:74]:74s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:10.1f} total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} pct={r['Percentage']}")
    bench = None
    for line in open(f"{root}/{row}_trace.log"):
        if line.startswith("{"):
            bench = json.loads(line)["deep"][row]
    counters = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob(f"{root}/{row}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if "pass_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    kind = "0" if ", true>" in r["Kernel_Name"] else "1"
                    counters.setdefault(kind, {}).setdefault(c, []).append(float(r["Counter_Value"]))
    kernels = {}
    for kind, cs in counters.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            kernels[kind] = {"FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                             "dispatches_sampled": len(cs["FETCH_SIZE"])}
    traffic[row] = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --only {row}`; reads = 2 x "
                              "FETCH_SIZE (gfx950 correction), writes = WRITE_SIZE, KiB -> bytes; mean over the row's launches",
                    "kernels": kernels}
    if bench:
        print(f"  bench (same process): {bench['value']:.0f} evals/s on one stream")
        for k in bench["kernels_one_stream"]:
            kind = "0" if ", true>" in k["kernel"] else "1"
            hbm = kernels.get(kind, {}).get("hbm_bytes_per_launch")
            print(f"    {k['kernel'][:44]:44s} launches={k['launches']:5d} states/launch={k['states_per_launch']:5.1f} HIP-event avg_us={k['avg_launch_us']:8.1f} "
                  f"alg_MB={k['algorithmic_bytes_per_launch'] / 1e6:8.1f} frac_hbm={k['frac_hbm_algorithmic']:.3f} frac_fp={k.get('frac_fp64', k.get('frac_fp32', 0.0)):.3f} "
                  f"PMC traffic_MB={(hbm or 0) / 1e6:8.1f}" + (f" -> {hbm / (k['avg_launch_us'] * 1e-6) / 1e12:.2f} TB/s moved" if hbm else ""))
with open(f"{root}/traffic.json", "w") as f:
    json.dump(traffic, f, indent=1)
print("\n(traffic.json written: merge into profiles/r03_traffic.json)")
