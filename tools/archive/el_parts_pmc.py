"""PMC input for the elasticity apply at 512 x 128 x 128: marching part only / boundary pass only / both, and the z-chunk length
(el_fast_waves) of the marching part.  Run under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE; variants are told apart by Grid_Size
(printed here as workgroups per variant)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
n = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 128, 128)
g = L.make_grid(3, n, (1.0, 0.25, 0.25))
for part, waves in ((0, 4096), (1, 4096), (2, 4096), (1, 2048), (1, 1024), (1, 8192)):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_debug_part", part))
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_fast_waves", waves))
    ms = C.c_double()
    L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 6, C.byref(ms)))
    print(n, "part", part, "waves", waves, round(ms.value * 1e3, 1), "us", flush=True)
