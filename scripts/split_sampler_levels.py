"""Which level of the split sampler is off: marginals of the samples over side A / side B / (side, top bits of the other)."""
import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import helpers
from queasars_amd import _lib
from queasars_amd.ir import QSV_OP_DTYPE
from test_gpu_configs import _one_circuit_per_key_count, _sampler_device

n, layers, seed, want_k = (int(v) for v in sys.argv[1:5])
shots = 400_000
chosen = _one_circuit_per_key_count(n, layers, 48, seed)
c, p = chosen[want_k]
ops = c.packed(); cap = 4 * len(ops) + 64
a, b = np.zeros(cap, dtype=QSV_OP_DTYPE), np.zeros(cap, dtype=QSV_OP_DTYPE)
na, nb, mask = C.c_int(0), C.c_int(0), C.c_uint64(0)
k = _lib.load().qsv_split_describe(n, len(ops), _lib.as_ptr(ops), 12, C.byref(mask), _lib.as_ptr(a), cap, C.byref(na), _lib.as_ptr(b), cap, C.byref(nb))
mask_a = int(mask.value); mask_b = ((1 << n) - 1) & ~mask_a
print("keys", k, "mask A", bin(mask_a), "bits A", bin(mask_a).count("1"))
dev = _sampler_device(n, True)
dev.set_operator(helpers.random_ising_operator(n, seed=n))
states, _ = dev.sample_batch([c], [p], shots, seed=17)
st = states[0].astype(np.int64)
probs = np.abs(helpers.oracle_state(c, p)) ** 2
index = np.arange(1 << n)

def extract(v, m):
    out = np.zeros_like(v); pos = 0
    for q in range(n):
        if m >> q & 1:
            out |= ((v >> q) & 1) << pos; pos += 1
    return out

for name, m in (("A", mask_a), ("B", mask_b)):
    bits = bin(m).count("1")
    key_all = extract(index, m)
    exact = np.bincount(key_all, weights=probs, minlength=1 << bits)
    got = np.bincount(extract(st, m), minlength=1 << bits)
    mean = shots * exact; big = mean >= 5
    z = (got[big] - mean[big]) / np.sqrt(mean[big])
    print(f"marginal over side {name}: bins {big.sum()}, chi2/dof {float((z*z).sum())/big.sum():.3f}, max|z| {np.abs(z).max():.2f}")
    # ... and with the top bits (block number) of the other side
    other = mask_b if name == "A" else mask_a
    ob = bin(other).count("1")
    if ob > 6:
        key2 = key_all | ((extract(index, other) >> 6) << bits)
        exact2 = np.bincount(key2, weights=probs, minlength=1 << (bits + ob - 6))
        got2 = np.bincount(extract(st, m) | ((extract(st, other) >> 6) << bits), minlength=1 << (bits + ob - 6))
        mean = shots * exact2; big = mean >= 5
        z = (got2[big] - mean[big]) / np.sqrt(mean[big])
        print(f"  with the block of the other side: bins {big.sum()}, chi2/dof {float((z*z).sum())/big.sum():.3f}, max|z| {np.abs(z).max():.2f}")
