import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import heat
heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 64, 64, 64, 1.0, 0.0, 20.0, 0.0025, 3)
heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 64, 64, 64, 1.0, 0.0, 20.0, 0.0025, 6)
