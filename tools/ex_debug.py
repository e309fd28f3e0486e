import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
for n in ((260, 6, 11), (512, 4, 10), (130, 9, 5), (384, 3, 9)):
    g = L.make_grid(3, n, (1.0, 1.0, 1.0))
    N = (n[0]+1)*(n[1]+1)*(n[2]+1)
    x = np.random.default_rng(1).uniform(-1, 1, N)
    ys = []
    for ex in (0, 1):
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"fast_ex", ex))
        y = np.empty(N)
        L.check(ctx, lib.pdegpu_heat_apply(ctx, g, 0.0, 1.0, L.dptr(x), L.dptr(y), None, 1))
        ys.append(y.reshape(n[2]+1, n[1]+1, n[0]+1))
    d = np.abs(ys[0] - ys[1])
    bad = np.argwhere(d > 1e-12 * np.abs(ys[0]).max())
    print(n, "mismatches", len(bad), "i values", sorted(set(bad[:, 2].tolist()))[:20], "k", sorted(set(bad[:, 0].tolist()))[:12], "j", sorted(set(bad[:, 1].tolist()))[:8])
