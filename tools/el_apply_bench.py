"""Scratch: elasticity operator, vector block stencil vs MFMA element-by-element, BASELINE config 5 grid."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
import time
from pdegpu import elasticity
for gm in (1, 2, 3, 4, 8):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_grid_mult", gm))
    g = L.make_grid(3, (256, 64, 64), (1.0, 0.25, 0.25))
    el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
    ms = C.c_double()
    L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
    t0 = time.perf_counter()
    f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, 256, 64, 64, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
    dt = time.perf_counter() - t0
    print(f"el_grid_mult={gm}: apply {ms.value*1e3:.1f} us; config-5 solve {dt:.3f} s, {f.meta['cg_iterations']} its, {dt/f.meta['cg_iterations']*1e6:.1f} us/it", flush=True)
for n in ():
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    for mf in (0, 1, 0, 1):
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, mf, 20, C.byref(ms)))
        print(f"{n} {'MFMA EBE' if mf else 'vector stencil'}: {ms.value*1e3:.1f} us/apply, {48*nodes/ms.value/1e6:.0f} GB/s alg, "
              f"{(1152 if mf else 270)*nodes/ms.value/1e9:.2f} TFLOP/s", flush=True)
