"""Host-side load vector of `project(Expression(..., degree=2), V)` (fenics_mcp_server.py:283-284,
289-290, 414-415, 420-421, 678-679, 684-685).

DOLFIN treats an `Expression(degree=2)` as a P2 function: on every cell it is interpolated at the
vertices and edge midpoints, and `project` integrates b_i = int I2(f) phi_i exactly before solving the
consistent-mass system M u = b.  This module builds b on the structured simplex split (1 interval /
2 triangles / 6 Kuhn tets per grid cell) with slice arithmetic; the mass solve itself runs on the GPU
(pdegpu_mass_solve).  It is one-off O(N) set-up work, not part of the timed solve loop."""
from __future__ import annotations

import itertools
import math

import numpy as np

# local vertices of the simplices of one grid cell, vertex id v = i + 2j + 4k
_SIMPLICES = {
    1: [(0, 1)],
    2: [(0, 1, 3), (0, 2, 3)],                                                    # RectangleMesh diagonal "right"
    3: [(0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7)],   # BoxMesh (Kuhn)
}


def trig_product(kind: str, amplitude: float, wavenumber: float, dim: int):
    """A * prod_d cos(k x_d)  or  A * prod_d sin(k x_d)  (:283, 289, 414, 420, 678, 684)."""
    f = np.cos if kind == "cosine" else np.sin

    def fn(*coords):
        out = amplitude
        for c in coords[:dim]:
            out = out * f(wavenumber * c)
        return out
    return fn


def _simplex_integral(d, *powers):
    """int over a unit-volume d-simplex of prod lambda_a^p_a = d! prod p_a! / (sum p_a + d)!"""
    num = math.factorial(d)
    for p in powers:
        num *= math.factorial(p)
    return num / math.factorial(sum(powers) + d)


def p2_expression_load(dim, n, Ls, fn, element_mode=0, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    """b[N] (x fastest) with b_i = sum_T int_T I2(f) lambda_i, I2 = P2 interpolant of fn on T; the mesh is
    [origin, origin + Ls] (the BoxMesh fallback of the cylinder starts at (0, -R, -R))."""
    n = [int(n[r]) if r < dim else 0 for r in range(3)]
    h = [float(Ls[r]) / n[r] if r < dim else 0.0 for r in range(3)]
    shape = tuple(n[r] + 1 for r in (2, 1, 0))                                    # [z][y][x]
    b = np.zeros(shape)
    cell_axes = [float(origin[r]) + np.arange(n[r]) * h[r] if r < dim else np.zeros(1) for r in range(3)]
    Z0, Y0, X0 = np.meshgrid(cell_axes[2], cell_axes[1], cell_axes[0], indexing="ij")   # cell origins
    vol = np.prod([h[r] for r in range(dim)]) / math.factorial(dim)
    w_vs = 2 * _simplex_integral(dim, 3) - _simplex_integral(dim, 2)              # vertex basis, same vertex
    w_vo = 2 * _simplex_integral(dim, 2, 1) - _simplex_integral(dim, 1, 1)        # vertex basis, other vertex
    w_ei = 4 * _simplex_integral(dim, 2, 1)                                       # edge basis, edge contains i
    w_eo = 4 * _simplex_integral(dim, 1, 1, 1) if dim >= 2 else 0.0               # edge basis, edge opposite

    def pos(v):                                                                   # coordinates of local vertex v
        return (X0 + (v & 1) * h[0], Y0 + ((v >> 1) & 1) * h[1], Z0 + ((v >> 2) & 1) * h[2])

    for simplex in _SIMPLICES[dim]:
        fv = [fn(*pos(v)) for v in simplex]
        edges = list(itertools.combinations(range(len(simplex)), 2))
        fe = []
        for a, c in edges:
            pa, pc = pos(simplex[a]), pos(simplex[c])
            fe.append(fn(*(0.5 * (pa[r] + pc[r]) for r in range(3))))
        for i, v in enumerate(simplex):
            acc = np.zeros_like(X0)
            for a in range(len(simplex)):
                acc = acc + fv[a] * (w_vs if a == i else w_vo)
            for e, (a, c) in enumerate(edges):
                acc = acc + fe[e] * (w_ei if i in (a, c) else w_eo)
            bx, by, bz = v & 1, (v >> 1) & 1, (v >> 2) & 1
            b[bz:bz + max(n[2], 1), by:by + max(n[1], 1), bx:bx + n[0]] += acc * vol
    return np.ascontiguousarray(b.ravel())
