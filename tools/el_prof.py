import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
for n in ((256, 64, 64), (512, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    ms = C.c_double()
    L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 10, C.byref(ms)))
    print(n, ms.value * 1e3, "us")
