"""Single-gate sweeps at n = 24 / 26 / 27 with non-temporal state stores (and loads): builds variant libraries, runs each."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
VARIANTS = {"base": (), "nt_store": ("QSV_NT=1",), "nt_store_load": ("QSV_NT=2",)}
def lib_of(name): return ROOT / "queasars_amd" / f"libqsv_abl_{name}.so"
if sys.argv[1] == "build":
    from queasars_amd import _build
    for name, d in VARIANTS.items():
        if d: print(_build.build(force=True, defines=d, lib_path=lib_of(name)))
elif sys.argv[1] == "one":
    from queasars_amd.circuit_evaluation import StatevectorDevice
    for n in (24, 26, 27):
        dev = StatevectorDevice(n, group=1)
        rates = []
        for target in (0, 7, 12, n - 3, n - 1):
            for control in (-1, (target + n // 2) % n):
                ms = dev.bench_gate(target, control, reps=20)
                rates.append(32.0 * (1 << n) / ms / 1e6)
        print(f"  n={n}: mean {sum(rates)/len(rates):7.1f} GB/s  min {min(rates):7.1f}  max {max(rates):7.1f}")
        dev.close()
else:
    for name in VARIANTS:
        env = dict(os.environ)
        if name != "base": env["QSV_LIBRARY"] = str(lib_of(name))
        print(name, flush=True)
        subprocess.run([sys.executable, __file__, "one"], env=env)
