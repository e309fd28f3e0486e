"""Scratch: elasticity apply, thread-per-node vs z-marching kernel, config 5 and a larger grid; config-5 solve time."""
import ctypes as C, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
from pdegpu import elasticity
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
def tun(**kw):
    for k, v in kw.items():
        L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
for n in ((256, 64, 64), (512, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    for cfg in (dict(el_fast=0), dict(el_fast=1, el_fast_ry=1, el_fast_waves=1024), dict(el_fast=1, el_fast_ry=1, el_fast_waves=2048),
                dict(el_fast=1, el_fast_ry=1, el_fast_waves=4096), dict(el_fast=1, el_fast_ry=1, el_fast_waves=8192),
                dict(el_fast=1, el_fast_ry=2, el_fast_waves=2048), dict(el_fast=1, el_fast_ry=2, el_fast_waves=4096)):
        tun(**cfg)
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
        print(f"{n} {cfg}: {ms.value*1e3:.1f} us/apply, {48*nodes/ms.value/1e6:.0f} GB/s alg", flush=True)
for cfg in (dict(el_fast=0), dict(el_fast=1, el_fast_ry=1, el_fast_waves=2048), dict(el_fast=1, el_fast_ry=2, el_fast_waves=4096)):
    tun(**cfg)
    for rep in range(2):
        t0 = time.perf_counter()
        f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, 256, 64, 64, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
        dt = time.perf_counter() - t0
    print(f"{cfg}: config-5 solve {dt:.3f} s, {f.meta['cg_iterations']} its, {dt/f.meta['cg_iterations']*1e6:.1f} us/it, max vm {f.values.max():.6e}", flush=True)
