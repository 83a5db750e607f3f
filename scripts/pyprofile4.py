"""cProfile of one end-to-end EVQE solve of the notebook's JSSP instance (estimator branch): where the host time goes."""
import cProfile, pstats, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
import config4
import jssp_instances as inst
from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
branch = sys.argv[1] if len(sys.argv) > 1 else "estimator"
config4.solve(enc, branch, 0, 8)  # warm
pr = cProfile.Profile()
pr.enable()
out = config4.solve(enc, branch, 1, 8)
pr.disable()
print(out["seconds"], out["circuit_evaluations"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
