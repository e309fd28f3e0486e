"""Round-2 experiment 13 (one GPU): config 5 (256 x 64 x 64 cantilever, static elasticity PCG) — device time per CG iteration against the
grid size of the flat vector kernels (el_grid_mult workgroups per CU), rows per wave of the apply and the face folding."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
n = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 64, 64)
g = L.make_grid(3, n, (1.0, 0.25, 0.25))
N = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
field = np.empty(N)
def run(**tun):
    for k, v in tun.items():
        L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
    el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
    el.body[2] = -76518.0
    el.rtol, el.max_iter = 1e-30, 600           # fixed work: 600 iterations
    el.quantity = L.QUANTITY_STRESS
    st = L.Stats()
    lib.pdegpu_elasticity_solve(ctx, g, el, L.dptr(field), None, st)      # ends with 'not converged' by design: the statistics are filled
    d = st.as_dict()
    return d["cg_ms"] * 1e3 / max(1, d["cg_iterations"])
variants = [dict(el_grid_mult=4, el_fast_yzface=3, el_fast_nt=nt) for nt in (-1, 0, 1)] + [dict(el_grid_mult=4, el_fast_yzface=0, el_fast_nt=0)]
res = {}
for rnd in range(3):
    for k, v in enumerate(variants):
        res.setdefault(k, []).append(run(**v))
for k, v in enumerate(variants):
    print(n, v, round(float(np.median(res[k])), 2), "us per iteration", flush=True)
