"""rocprofv3 input: the elasticity block apply at sizes beyond BASELINE's (one launch shape per size, told apart by Grid_Size in the counter file).
usage (under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE): python3 tools/el_prof_sizes.py"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
for n in ((512, 128, 128), (640, 160, 160), (256, 256, 256), (320, 320, 320)):
    g = L.make_grid(3, n, (1.0, n[1] / n[0], n[2] / n[0]))
    ms = C.c_double()
    L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 6, C.byref(ms)))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    print(n, round(ms.value * 1e3, 1), "us; algorithmic MB:", round(48 * nodes / 1e6, 1), file=sys.stderr)
