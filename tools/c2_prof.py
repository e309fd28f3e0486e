"""configs 2 and 5 end to end for rocprofv3 --kernel-trace --stats.  The persistent 2-D kernel is launched as an ordinary kernel here (tunable
persist2d_launch = 1): after a cooperative launch the profiler's exit handler faults inside libhsa-runtime64 (profiles/r03_rocprofv3_exit_abort.log)."""
import os
import sys, time
from pathlib import Path
os.environ.setdefault("PDEGPU_PERSIST2D_LAUNCH", "1")
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import heat, elasticity, _lib
f = heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 10)
print("2d iters", f.meta["cg_iterations"])
f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, 256, 64, 64, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
print("el iters", f.meta["cg_iterations"])
_lib.close_contexts()          # nothing of this library is alive when the profiler's exit handlers run
import os
if os.environ.get("PDEGPU_DUMP_MAPS"):          # which libraries own the frames of an exit-time fault (profiles/r03_rocprofv3_exit_abort.log)
    with open(os.environ["PDEGPU_DUMP_MAPS"], "w") as fh:
        fh.write("".join(l for l in open("/proc/self/maps") if "r-xp" in l))
