"""CPU ORACLE (test infrastructure, NOT product code) — the Q1 (multilinear) element on the structured interval / rectangle / box grid.

BASELINE.json's configs and `north_star` speak of a "Q1 grid" and a "7-/27-point stencil"; the reference itself is P1 on DOLFIN's simplex split
(/root/reference/fenics_mcp_server.py:230, 370, 535 — `FunctionSpace(mesh, "P", 1)`), so this module restates NO reference arithmetic: it is the
textbook tensor-product element on the same grid, run through the SAME solver loops the reference uses (`oracle.fem_p1._heat_core`: steady
`kappa K u = f m` / backward Euler `(M + dt kappa K) u = M u_n + dt f m`, Dirichlet rows fixed, re-factorised every step — :259-318, :391-447,
:653-716), so that the build's Q1 element mode (PDEGPU_ELEM_Q1: `macro_q1`, the 27-point marching kernel `k_fast<27, ...>`, the truncated
27-point classes on natural faces) has a checker that is not the build's own class tables.  Only tests/ may import it.

PARITY STATUS: nothing in the reference pins Q1 (it has no Q1 path).  The module is pinned on the CPU (tests/test_oracle_q1.py) by two
independent assemblies that must agree to rounding —
  (a) `assemble_gauss`: element by element, 2-point Gauss-Legendre quadrature per axis (exact for products of multilinear functions on a box),
  (b) `assemble_kron`:  Kronecker products of the 1-D P1 matrices, K = Kx (x) My (x) Mz + Mx (x) Ky (x) Mz + Mx (x) My (x) Kz, M = Mx (x) My (x) Mz
      (the tensor-product identity of the Q1 space; 1-D matrices (1/h)[[1,-1],[-1,1]] and (h/6)[[2,1],[1,2]]),
by the closed-form interior stencils of SURVEY.md App. A.3 (3-D stiffness x 1/h: centre 8/3, faces 0, edges -1/6, corners -1/12; mass x 216/h^3:
64, 16, 4, 1), by symmetry / constants in the null space of K / M 1 = load / sum(load) = volume, and by O(h^2) convergence of its solutions to
those of the P1 oracle.  Node order: x fastest, like every array of the C-ABI.
"""
from __future__ import annotations

import itertools

import numpy as np
import scipy.sparse as sp

from . import fem_p1 as p1

_GP = (0.5 - 0.28867513459481288225, 0.5 + 0.28867513459481288225)      # 2-point Gauss-Legendre on [0, 1], weights 1/2


def grid_coords(n, lengths):
    """node coordinates [N, d], x fastest: x_r = L_r * i_r / n_r (the mesh generators' arithmetic, fem_p1.box_mesh)"""
    d = len(n)
    axes = [lengths[r] * np.arange(n[r] + 1) / n[r] for r in range(d)]
    mesh = np.meshgrid(*axes[::-1], indexing="ij")                         # slowest axis first
    return np.stack([m.ravel() for m in mesh[::-1]], axis=1)


def element_matrices_gauss(h):
    """stiffness / mass / load of ONE multilinear element with edge lengths h[0..d-1]; local vertex v = i + 2 j + 4 k"""
    d = len(h)
    nv = 1 << d
    Ke, Me, le = np.zeros((nv, nv)), np.zeros((nv, nv)), np.zeros(nv)
    vol = float(np.prod(h))
    for qp in itertools.product(_GP, repeat=d):                            # qp[r] = abscissa on axis r
        N = np.ones(nv)
        G = np.ones((nv, d))
        for v in range(nv):
            for r in range(d):
                f = qp[r] if (v >> r) & 1 else 1.0 - qp[r]
                N[v] *= f
                for e in range(d):
                    G[v, e] *= (((1.0 if (v >> r) & 1 else -1.0) / h[r]) if e == r else f)
        w = vol / (1 << d)
        Ke += w * G @ G.T
        Me += w * np.outer(N, N)
        le += w * N
    return Ke, Me, le


def _cell_nodes(n):
    """[ncell, 2^d] global node index of every local vertex of every cell"""
    d = len(n)
    nn = [n[r] + 1 for r in range(d)]
    stride = [int(np.prod(nn[:r])) for r in range(d)]
    idx = np.meshgrid(*[np.arange(n[r]) for r in range(d)][::-1], indexing="ij")[::-1]
    v0 = sum(idx[r].ravel() * stride[r] for r in range(d))
    return np.stack([v0 + sum(((v >> r) & 1) * stride[r] for r in range(d)) for v in range(1 << d)], axis=1)


def assemble_gauss(n, lengths):
    """K, M (csr) and the load vector by element-wise Gauss quadrature + scatter (natural boundary on every face)"""
    d = len(n)
    h = [lengths[r] / n[r] for r in range(d)]
    Ke, Me, le = element_matrices_gauss(h)
    cells = _cell_nodes(n)
    N = int(np.prod([n[r] + 1 for r in range(d)]))
    nc = cells.shape[0]
    K = p1._coo(cells, np.broadcast_to(Ke, (nc,) + Ke.shape), N)
    M = p1._coo(cells, np.broadcast_to(Me, (nc,) + Me.shape), N)
    load = np.zeros(N)
    np.add.at(load, cells.ravel(), np.tile(le, nc))
    return K, M, load


def _one_d(n, L):
    h = L / n
    main_k = np.full(n + 1, 2.0 / h)
    main_k[[0, -1]] = 1.0 / h
    K = sp.diags([np.full(n, -1.0 / h), main_k, np.full(n, -1.0 / h)], [-1, 0, 1], format="csr")
    main_m = np.full(n + 1, 4.0 * h / 6.0)
    main_m[[0, -1]] = 2.0 * h / 6.0
    M = sp.diags([np.full(n, h / 6.0), main_m, np.full(n, h / 6.0)], [-1, 0, 1], format="csr")
    load = np.full(n + 1, h)
    load[[0, -1]] = h / 2.0
    return K, M, load


def assemble_kron(n, lengths):
    """the same matrices as Kronecker products of the 1-D P1 matrices (x fastest <=> the x factor is the innermost one)"""
    d = len(n)
    one = [_one_d(n[r], lengths[r]) for r in range(d)]

    def kron(factors):                                                     # factors[r] belongs to axis r
        out = factors[0]
        for r in range(1, d):
            out = sp.kron(factors[r], out, format="csr")
        return out

    K = None
    for r in range(d):
        term = kron([one[e][0] if e == r else one[e][1] for e in range(d)])
        K = term if K is None else K + term
    M = kron([one[e][1] for e in range(d)])
    load = one[0][2]
    for r in range(1, d):
        load = np.kron(one[r][2], load)
    return K.tocsr(), M.tocsr(), load


def _solve(n, lengths, dir_mask, dir_vals, kappa, T_initial, dt, num_steps, steady, source_type, source_value, initial_type):
    if initial_type not in ("constant", "zero"):
        raise NotImplementedError("Q1 oracle: constant / zero initial conditions (the projected trigonometric ones are a P1-split construction)")
    mats = assemble_gauss(n, lengths)
    return p1._heat_core(None, None, dir_mask, dir_vals, kappa, T_initial, dt, num_steps, steady, source_type, source_value, initial_type, 1.0, 1.0,
                         matrices=mats)


def solve_heat_2d(Lx, Ly, nx, ny, diffusivity, T_boundary, T_initial, dt, num_steps, steady=False, source_type="none", source_value=0.0,
                  initial_type="constant") -> p1.OracleField:
    """`_solve_heat_2d_raw` (:345-468) with the bilinear element in place of P1 on the triangle split"""
    n = (nx + 1) * (ny + 1)
    ix, iy = np.arange(n) % (nx + 1), np.arange(n) // (nx + 1)
    dir_mask = (ix == 0) | (ix == nx) | (iy == 0) | (iy == ny)
    vals, times = _solve((nx, ny), (Lx, Ly), dir_mask, np.full(n, float(T_boundary)), diffusivity, T_initial, dt, num_steps, steady, source_type,
                         source_value, initial_type)
    c = grid_coords((nx, ny), (Lx, Ly))
    return p1.OracleField(np.concatenate([c, np.zeros((n, 1))], axis=1), vals, times, 2, {"element": "q1"})


def solve_heat_3d(Lx, Ly, Lz, nx, ny, nz, diffusivity, T_boundary, T_initial, dt, num_steps, steady=False, source_type="none", source_value=0.0,
                  initial_type="constant", T_left=None, T_right=None, T_side=None) -> p1.OracleField:
    """`_solve_heat_3d_raw`, box branch (:475-762), with the trilinear element; the Dirichlet set is the reference's (fem_p1.box_dirichlet)"""
    dir_mask, dir_vals = p1.box_dirichlet(nx, ny, nz, T_boundary, T_left, T_right, T_side)
    vals, times = _solve((nx, ny, nz), (Lx, Ly, Lz), dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps, steady, source_type, source_value,
                         initial_type)
    return p1.OracleField(grid_coords((nx, ny, nz), (Lx, Ly, Lz)), vals, times, 3, {"element": "q1"})
