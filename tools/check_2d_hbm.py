"""2-D grids whose vectors are HBM-resident (flat view of the marching kernels under the fitted chunk policy): bare apply against the class-table kernel, and a\nshort transient solve.  usage: python tools/check_2d_hbm.py"""
import sys, ctypes as C, numpy as np
sys.path[:0]=['/root/repo','/root/repo/pde-solver_amd']
from pdegpu import _lib as L
lib=L.load_library(); ctx=L.context(0)
for n in ((3000,3000),(4096,2200)):
    g=L.make_grid(2,n,(1.0,0.7))
    N=(n[0]+1)*(n[1]+1)
    rng=np.random.default_rng(1); x=rng.standard_normal(N)
    out=[]
    for fast in (1,0):
        y=np.empty(N); load=np.empty(N)
        L.check(ctx, lib.pdegpu_heat_apply(ctx,g,0.3,1.0,L.dptr(x),L.dptr(y),L.dptr(load),fast))
        out.append(y)
    a = out[0].reshape(n[1] + 1, n[0] + 1)[1:-1, 1:-1]
    b = out[1].reshape(n[1] + 1, n[0] + 1)[1:-1, 1:-1]          # (the fast path applies the masked operator: compare the free nodes)
    print(n, "rel diff fast vs class-table on the free nodes:", np.abs(a - b).max() / np.abs(b).max())
# and a 2-D transient solve through the tool path at an HBM-resident size: two steps, residual history sane
from pdegpu import heat
f=heat._solve_heat_2d_raw(1.0,1.0,3000,3000,1.0,0.0,20.0,1e-4,2)
print("2-D 3000^2 two steps:", f.meta.get("solver_path"), f.meta.get("cg_iterations"), float(f.values[-1].max()))
