"""Split sampler against exact probabilities: worst bins and a pooled chi-square (diagnostic for the GPU test)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import helpers
from test_gpu_configs import _one_circuit_per_key_count, _sampler_device

n, layers, seed = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (17, 4, 3)))
shots = 400_000
chosen = _one_circuit_per_key_count(n, layers, 48, seed)
dev = _sampler_device(n, True)
dev.set_operator(helpers.random_ising_operator(n, seed=n))
for k, (c, p) in sorted(chosen.items()):
    states, _ = dev.sample_batch([c], [p], shots, seed=17)
    probs = np.abs(helpers.oracle_state(c, p)) ** 2
    counts = np.bincount(states[0].astype(np.int64), minlength=1 << n)
    mean = shots * probs
    big = mean >= 5
    z = (counts[big] - mean[big]) / np.sqrt(mean[big] * (1 - probs[big]))
    small_c, small_m = counts[~big].sum(), mean[~big].sum()
    worst_small = np.argmax(np.where(~big, counts - mean, -1))
    chi = float((z ** 2).sum())
    print(f"K={k}: bins>=5: {big.sum()}, max|z|={np.abs(z).max():.2f}, chi2/dof={chi / max(1, big.sum()):.4f}; "
          f"small bins: count {small_c} vs mean {small_m:.1f}; worst small bin {worst_small}: count {counts[worst_small]} mean {mean[worst_small]:.3f}; "
          f"zero-prob hits {counts[probs == 0].sum()}", flush=True)
