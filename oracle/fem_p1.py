"""CPU ORACLE (test infrastructure, NOT product code) — P1 finite elements on DOLFIN's simplex
split of an interval / rectangle / box, assembled to a sparse matrix and solved with sparse LU.

This file restates, with numpy/scipy, the algorithm the reference executes through
FEniCS/DOLFIN 2019.1.0 on its box-geometry heat / static-elasticity path
(/root/reference/fenics_mcp_server.py:204-762 and :1470-1892).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path
(pde-solver_amd/) never does.

PARITY STATUS: **parity unpinned by the reference.**  The arithmetic lives in a third-party
dependency that is absent from /root/reference and from this image (fenics-dolfin / ffc / fiat /
ufl / dijitso = 2019.1.0 from conda-forge, /root/reference/Dockerfile:18-24,
requirements.txt:21-27; `import dolfin` -> ModuleNotFoundError, an ordinary missing dependency).
The reference ships no tests, golden vectors or stored outputs for this path.
The restatement is therefore pinned by what CAN be pinned offline (tests/test_oracle_*.py):
analytic nodally-exact solutions, the closed-form P1 stencils / macro-element matrices, patch
tests, rigid-body null spaces, O(h^2) convergence, and symmetric-positive-definiteness.

What is restated (published algorithm of the named dependency + the reference's call sites):
  * meshes: IntervalMesh / RectangleMesh(diagonal="right") / BoxMesh (6 Kuhn tets sharing the
    v0-v7 diagonal) — DOLFIN 2019.1.0 mesh generators; call sites fenics_mcp_server.py:229,369,533,
    1516,1648,1803.
  * FunctionSpace(mesh,"P",1) / VectorFunctionSpace: one dof (d dofs) per vertex (:230,370,535,
    1517,1649-1650,1804-1805).  Our dof order is the vertex order (x fastest); the reference
    exposes its order only through tabulate_dof_coordinates, so any consistent order is valid.
  * assemble(): exact integration of the constant-coefficient P1 forms.
  * DirichletBC + solve(a == L, u, bcs): identity rows for constrained dofs, then sparse LU,
    *re-assembled and re-factorised every time step* (:309-318, 438-447, 707-716).  We solve the
    mathematically identical symmetric-elimination system (free rows, lifted right-hand side).
  * project(expr, V): consistent-mass solve (:284,290,415,421,679,685,1542,1546,1714,1862);
    Expression(..., degree=2) is interpolated to P2 on each cell before integration.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ----------------------------------------------------------------------------------------------
# result container mirroring fenics_mcp_server.py:168-183 (numpy-backed)
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleField:
    coords: np.ndarray          # [N,3]
    values: np.ndarray          # [Nt,N]
    times: np.ndarray           # [Nt]
    dim: int
    meta: Dict[str, Any] = field(default_factory=dict)
    extra: Dict[str, Any] = field(default_factory=dict)   # oracle-only diagnostics (e.g. displacement)


# ----------------------------------------------------------------------------------------------
# meshes
# ----------------------------------------------------------------------------------------------
def interval_mesh(nx: int, a: float, b: float):
    """IntervalMesh(nx, a, b) — fenics_mcp_server.py:229."""
    x = a + (b - a) * np.arange(nx + 1) / nx
    cells = np.stack([np.arange(nx), np.arange(1, nx + 1)], axis=1)
    return x.reshape(-1, 1), cells


def rectangle_mesh(Lx: float, Ly: float, nx: int, ny: int):
    """RectangleMesh(Point(0,0), Point(Lx,Ly), nx, ny), diagonal="right" — :369.
    cell (ix,iy): v0 lower-left, v1=v0+1, v2=v0+(nx+1), v3=v2+1 -> (v0,v1,v3), (v0,v2,v3)."""
    xs = Lx * np.arange(nx + 1) / nx
    ys = Ly * np.arange(ny + 1) / ny
    X, Y = np.meshgrid(xs, ys, indexing="xy")            # y slow, x fast
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (iy * (nx + 1) + ix).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.concatenate([np.stack([v0, v1, v3], 1), np.stack([v0, v2, v3], 1)], axis=0)
    return coords, cells


# the six Kuhn tets of a cube, local vertex v = i + 2j + 4k, all sharing the diagonal v0-v7
KUHN_TETS = np.array([(0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7)])


def box_mesh(Lx: float, Ly: float, Lz: float, nx: int, ny: int, nz: int):
    """BoxMesh(Point(0,0,0), Point(Lx,Ly,Lz), nx, ny, nz) — :533, :1803."""
    xs = Lx * np.arange(nx + 1) / nx
    ys = Ly * np.arange(ny + 1) / ny
    zs = Lz * np.arange(nz + 1) / nz
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")     # z slowest, x fastest
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    px, pxy = nx + 1, (nx + 1) * (ny + 1)
    v0 = (iz * pxy + iy * px + ix).ravel()
    loc = np.stack([v0 + (l & 1) + ((l >> 1) & 1) * px + ((l >> 2) & 1) * pxy for l in range(8)], axis=1)
    cells = np.concatenate([loc[:, t] for t in KUHN_TETS], axis=0)
    return coords, cells


# ----------------------------------------------------------------------------------------------
# P1 element geometry
# ----------------------------------------------------------------------------------------------
def simplex_geometry(coords: np.ndarray, cells: np.ndarray):
    """|T| and the constant gradients of the barycentric basis, grads[c, a, :] = grad(lambda_a)."""
    d = coords.shape[1]
    P = coords[cells]                                     # [nc, d+1, d]
    J = (P[:, 1:, :] - P[:, :1, :]).transpose(0, 2, 1)    # columns = edge vectors
    detJ = np.linalg.det(J)
    vol = np.abs(detJ) / math.factorial(d)
    Jinv = np.linalg.inv(J)                               # rows = grad(lambda_1..d)
    grads = np.empty((cells.shape[0], d + 1, d))
    grads[:, 1:, :] = Jinv
    grads[:, 0, :] = -Jinv.sum(axis=1)
    return vol, grads


def _coo(cells, blocks, n):
    nc, m = cells.shape
    rows = np.repeat(cells, m, axis=1).ravel()
    cols = np.tile(cells, (1, m)).ravel()
    return sp.coo_matrix((blocks.reshape(nc, -1).ravel(), (rows, cols)), shape=(n, n)).tocsr()


def assemble_scalar(coords: np.ndarray, cells: np.ndarray):
    """K_ij = int grad phi_i . grad phi_j, M_ij = int phi_i phi_j (consistent), m_i = int phi_i."""
    n, d = coords.shape
    vol, grads = simplex_geometry(coords, cells)
    Ke = vol[:, None, None] * np.einsum("cad,cbd->cab", grads, grads)
    m = d + 1
    Mref = (np.ones((m, m)) + np.eye(m)) / ((d + 1) * (d + 2))
    Me = vol[:, None, None] * Mref[None]
    K = _coo(cells, Ke, n)
    M = _coo(cells, Me, n)
    load = np.zeros(n)
    np.add.at(load, cells.ravel(), np.repeat(vol / (d + 1), m))
    return K, M, load


# ----------------------------------------------------------------------------------------------
# "solve(a == L, u, bcs)": Dirichlet rows fixed, sparse LU (re-factorised by the caller per step)
# ----------------------------------------------------------------------------------------------
def solve_dirichlet(A: sp.csr_matrix, b: np.ndarray, dir_mask: np.ndarray, dir_vals: np.ndarray) -> np.ndarray:
    free = ~dir_mask
    u = np.where(dir_mask, dir_vals, 0.0)
    if free.any():
        Aff = A[free][:, free].tocsc()
        rhs = b[free] - A[free][:, dir_mask] @ u[dir_mask]
        u[free] = spla.splu(Aff).solve(rhs)
    return u


def _project_p2_expression(coords, cells, fn, M):
    """project(Expression(..., degree=2), V) (:284,415,679): the expression is interpolated into P2
    on every cell (vertices + edge midpoints), b_i = int I2(f) phi_i is integrated exactly, then the
    consistent-mass system M u = b is solved with LU (no boundary conditions)."""
    n, d = coords.shape
    vol, _ = simplex_geometry(coords, cells)
    m = d + 1
    P = coords[cells]                                     # [nc, m, d]
    fv = fn(P.reshape(-1, d)).reshape(-1, m)              # vertex values
    edges = list(itertools.combinations(range(m), 2))
    fe = np.stack([fn(0.5 * (P[:, a] + P[:, b])) for a, b in edges], axis=1)
    # exact integrals on a simplex (Dirichlet formula): int l_a^p l_b^q l_c^r = |T| d! p!q!r!/(p+q+r+d)!
    def I(*pw):
        num = math.factorial(d)
        for p in pw:
            num *= math.factorial(p)
        return num / math.factorial(sum(pw) + d)
    b = np.zeros(n)
    for i in range(m):
        acc = np.zeros(cells.shape[0])
        # vertex P2 basis N_a = l_a (2 l_a - 1)
        for a in range(m):
            w = (2 * I(3) - I(2)) if a == i else (2 * I(2, 1) - I(1, 1))
            acc += fv[:, a] * w
        # edge P2 basis N_ab = 4 l_a l_b
        for e, (a, bb) in enumerate(edges):
            w = 4 * I(2, 1) if i in (a, bb) else 4 * I(1, 1, 1)
            acc += fe[:, e] * w
        np.add.at(b, cells[:, i], acc * vol)
    return spla.splu(M.tocsc()).solve(b)


def _initial_condition(coords, cells, M, initial_type, T_initial, A, k):
    """fenics_mcp_server.py:276-292 / :408-423 / :672-687."""
    n, d = coords.shape
    if initial_type == "zero":
        return np.zeros(n)
    if initial_type in ("cosine", "sine"):
        f = np.cos if initial_type == "cosine" else np.sin
        fn = lambda X: A * np.prod(f(k * X), axis=1)      # noqa: E731  product over the d coordinates
        return _project_p2_expression(coords, cells, fn, M)
    return np.full(n, float(T_initial))


def _heat_core(coords, cells, dir_mask, dir_vals, kappa, T_initial, dt, num_steps, steady,
               source_type, source_value, initial_type, initial_amplitude, initial_wavenumber,
               refactor_each_step=True, matrices=None):
    """Shared body of the three Cartesian heat solvers (:259-318, :391-447, :653-716).  matrices = (K, M, load) of another element on the
    same nodes (oracle.fem_q1) runs the identical loops on them; constant / zero initial conditions only in that case."""
    K, M, load = assemble_scalar(coords, cells) if matrices is None else matrices
    f = float(source_value) if source_type == "constant" else 0.0
    if steady:
        u = solve_dirichlet(kappa * K, f * load, dir_mask, dir_vals)
        return u[None, :], np.array([0.0])
    if matrices is None:
        u_n = _initial_condition(coords, cells, M, initial_type, T_initial, initial_amplitude, initial_wavenumber)
    else:
        u_n = np.zeros(load.shape[0]) if initial_type == "zero" else np.full(load.shape[0], float(T_initial))
    u_n = np.where(dir_mask, dir_vals, u_n)               # bc.apply(u_n.vector())
    snaps, times = [u_n.copy()], [0.0]
    A = (M + dt * kappa * K).tocsr()
    free = ~dir_mask
    lu = None
    for n in range(num_steps):
        b = M @ u_n + dt * f * load
        if refactor_each_step or lu is None:              # the reference re-assembles + re-factorises
            Aff = A[free][:, free].tocsc()
            AfD = A[free][:, dir_mask]
            lu = spla.splu(Aff)
        u = np.where(dir_mask, dir_vals, 0.0)
        u[free] = lu.solve(b[free] - AfD @ u[dir_mask])
        snaps.append(u.copy())
        times.append((n + 1) * dt)
        u_n = u
    return np.array(snaps), np.array(times)


# ----------------------------------------------------------------------------------------------
# heat solvers — same signatures as the reference's raw solvers
# ----------------------------------------------------------------------------------------------
def solve_heat_1d(length, nx, diffusivity, T_left, T_right, T_initial, dt, num_steps, steady=False,
                  source_type="none", source_value=0.0, initial_type="constant",
                  initial_amplitude=1.0, initial_wavenumber=1.0, refactor_each_step=True) -> OracleField:
    """_solve_heat_1d_raw, fenics_mcp_server.py:204-338.  Output sorted by x (:252-254)."""
    coords, cells = interval_mesh(nx, 0.0, length)
    n = nx + 1
    dir_mask = np.zeros(n, bool)
    dir_vals = np.zeros(n)
    dir_mask[0], dir_vals[0] = True, T_left               # bcs = [bc_left, bc_right] (:239-241)
    dir_mask[-1], dir_vals[-1] = True, T_right
    vals, times = _heat_core(coords, cells, dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps, steady,
                             source_type, source_value, initial_type, initial_amplitude, initial_wavenumber,
                             refactor_each_step)
    c3 = np.zeros((n, 3))
    c3[:, 0] = coords[:, 0]
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian",
            "length": length, "source_type": source_type, "source_value": source_value, "steady": steady}
    return OracleField(c3, vals, times, 1, meta)


def solve_heat_2d(Lx, Ly, nx, ny, diffusivity, T_boundary, T_initial, dt, num_steps, steady=False,
                  source_type="none", source_value=0.0, initial_type="constant",
                  initial_amplitude=1.0, initial_wavenumber=1.0, refactor_each_step=True) -> OracleField:
    """_solve_heat_2d_raw, fenics_mcp_server.py:345-468."""
    coords, cells = rectangle_mesh(Lx, Ly, nx, ny)
    n = coords.shape[0]
    ix = np.arange(n) % (nx + 1)
    iy = np.arange(n) // (nx + 1)
    dir_mask = (ix == 0) | (ix == nx) | (iy == 0) | (iy == ny)     # on_boundary (:373-376)
    dir_vals = np.full(n, float(T_boundary))
    vals, times = _heat_core(coords, cells, dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps, steady,
                             source_type, source_value, initial_type, initial_amplitude, initial_wavenumber,
                             refactor_each_step)
    c3 = np.zeros((n, 3))
    c3[:, :2] = coords
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian",
            "Lx": Lx, "Ly": Ly, "source_type": source_type, "source_value": source_value, "steady": steady}
    return OracleField(c3, vals, times, 2, meta)


def box_dirichlet(nx, ny, nz, T_boundary=0.0, T_left=None, T_right=None, T_side=None):
    """Dirichlet set of the box branch, fenics_mcp_server.py:576-628, with exact index tests in place
    of near().  Directional mode: left = x==0, right = x==Lx, side = boundary nodes with 0<x<Lx;
    faces whose value is None stay natural.  bcs are applied in list order (left, right, side).
    [3P] DirichletBC's default method "topological" marks FACETS — a boundary facet is taken when the predicate
    holds at all of its vertices and at its midpoint — and constrains the dofs of the taken facets.  left_face /
    right_face / `boundary` take whole faces (= their node sets).  `other_faces` (:614-617) takes a side facet
    only if no vertex of it lies on x = 0 or x = Lx: impossible for nx <= 2 (T_side then constrains nothing);
    for nx >= 3 every side node with 0 < ix < nx is a vertex of such a facet, so the node set below is exact
    (enumerated against the facet rule in tests/test_oracle_known_answers.py)."""
    n = (nx + 1) * (ny + 1) * (nz + 1)
    idx = np.arange(n)
    ix = idx % (nx + 1)
    iy = (idx // (nx + 1)) % (ny + 1)
    iz = idx // ((nx + 1) * (ny + 1))
    onb = (ix == 0) | (ix == nx) | (iy == 0) | (iy == ny) | (iz == 0) | (iz == nz)
    mask = np.zeros(n, bool)
    vals = np.zeros(n)
    if T_left is None and T_right is None and T_side is None:
        mask[:] = onb
        vals[:] = float(T_boundary)
        return mask, vals
    if T_left is not None:
        s = ix == 0
        mask |= s
        vals[s] = T_left
    if T_right is not None:
        s = ix == nx
        mask |= s
        vals[s] = T_right
    if T_side is not None and nx >= 3:
        s = onb & (ix != 0) & (ix != nx)
        mask |= s
        vals[s] = T_side
    return mask, vals


def core_cells(coords, cells, core_radius):
    """cells of the composite core (:537-550): SubDomain.mark on cells takes a cell when inside() holds at all of its vertices
    and at its midpoint; inside = sqrt(y^2 + z^2) < core_radius"""
    P = coords[cells]
    rv = np.sqrt(P[:, :, 1] ** 2 + P[:, :, 2] ** 2)
    mid = P.mean(axis=1)
    rm = np.sqrt(mid[:, 1] ** 2 + mid[:, 2] ** 2)
    return (rv < core_radius).all(axis=1) & (rm < core_radius)


def solve_heat_3d(Lx, Ly, Lz, nx, ny, nz, diffusivity, T_boundary, T_initial, dt, num_steps, steady=False,
                  source_type="none", source_value=0.0, initial_type="constant",
                  initial_amplitude=1.0, initial_wavenumber=1.0, geometry_type="box",
                  T_left=None, T_right=None, T_side=None, refactor_each_step=True,
                  core_radius=None, core_diffusivity=None) -> OracleField:
    """_solve_heat_3d_raw, box branch, fenics_mcp_server.py:475-762 (box = :533,574,606-628,647,653-716); with
    core_radius / core_diffusivity the DG0 diffusivity of :537-572 on the plain box (distance from the x-axis, i.e. from the
    box edge y = z = 0)."""
    if geometry_type != "box":
        raise NotImplementedError("oracle restates the box branch only (SURVEY.md §8a row a3)")
    coords, cells = box_mesh(Lx, Ly, Lz, nx, ny, nz)
    dir_mask, dir_vals = box_dirichlet(nx, ny, nz, T_boundary, T_left, T_right, T_side)
    composite = core_radius is not None and core_diffusivity is not None
    if composite:
        kcell = np.where(core_cells(coords, cells, core_radius), core_diffusivity / diffusivity, 1.0)
        u0 = None
        if not steady:
            _, M, _ = assemble_scalar(coords, cells)
            u0 = _initial_condition(coords, cells, M, initial_type, T_initial, initial_amplitude, initial_wavenumber)
        vals, times = _weighted_heat_core(coords, cells, "one", dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps, steady,
                                          source_type, source_value, kcell, u0)
    else:
        vals, times = _heat_core(coords, cells, dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps, steady,
                                 source_type, source_value, initial_type, initial_amplitude, initial_wavenumber,
                                 refactor_each_step)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian",
            "Lx": Lx, "Ly": Ly, "Lz": Lz, "geometry_type": geometry_type, "source_type": source_type,
            "source_value": source_value, "steady": steady}
    if T_left is not None or T_right is not None or T_side is not None:
        for k, v in (("T_left", T_left), ("T_right", T_right), ("T_side", T_side)):
            if v is not None:
                meta[k] = v
    else:
        meta["T_boundary"] = T_boundary
    if composite:                                            # :747-753
        meta.update({"core_radius": core_radius, "core_diffusivity": core_diffusivity, "base_diffusivity": diffusivity})
    else:
        meta["diffusivity"] = diffusivity
    return OracleField(coords, vals, times, 3, meta)


# ----------------------------------------------------------------------------------------------
# static linear elasticity (:1470-1892)
# ----------------------------------------------------------------------------------------------
def lame(E, nu, plane_stress=False):
    """:1658-1665 (2-D) and :1813-1814 (3-D)."""
    mu = E / (2.0 * (1.0 + nu))
    lam = E * nu / (1.0 - nu ** 2) if plane_stress else E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    return lam, mu


def assemble_elasticity(coords, cells, lam, mu):
    """a = int sigma(u):eps(v), sigma = lam tr(eps) I + 2 mu eps; dof = d*vertex + component."""
    n, d = coords.shape
    vol, g = simplex_geometry(coords, cells)
    m = d + 1
    # K[(a,i),(b,j)] = |T| ( lam g_ai g_bj + mu g_aj g_bi + mu delta_ij g_a.g_b )
    gg = np.einsum("cad,cbd->cab", g, g)
    Ke = (lam * np.einsum("cai,cbj->caibj", g, g) + mu * np.einsum("caj,cbi->caibj", g, g)
          + mu * np.einsum("cab,ij->caibj", gg, np.eye(d)))
    Ke = Ke * vol[:, None, None, None, None]
    dofs = (cells[:, :, None] * d + np.arange(d)[None, None, :]).reshape(cells.shape[0], m * d)
    return _coo(dofs, Ke.reshape(cells.shape[0], m * d, m * d), n * d)


def cell_strain(coords, cells, u):
    """eps(u) = sym(grad u), constant per simplex.  u: [N,d] -> [nc,d,d]."""
    _, g = simplex_geometry(coords, cells)
    G = np.einsum("cai,cad->cid", u[cells], g)            # grad u [c, i, d]
    return 0.5 * (G + G.transpose(0, 2, 1))


def von_mises(eps, lam, mu, quantity):
    """:1694-1711 / :1844-1859 — deviator uses tr/3 with the d x d identity also when d == 2."""
    d = eps.shape[1]
    I = np.eye(d)
    tr_e = np.trace(eps, axis1=1, axis2=2)
    if quantity == "strain":
        dev = eps - (tr_e / 3.0)[:, None, None] * I
        return np.sqrt(2.0 / 3.0 * np.einsum("cij,cij->c", dev, dev))
    sig = lam * tr_e[:, None, None] * I + 2.0 * mu * eps
    tr_s = np.trace(sig, axis1=1, axis2=2)
    dev = sig - (tr_s / 3.0)[:, None, None] * I
    return np.sqrt(3.0 / 2.0 * np.einsum("cij,cij->c", dev, dev))


def project_cellwise(coords, cells, cell_vals, M):
    """project(piecewise-constant expr, P1) (:1714, :1862): M s = r, r_i = sum_T val_T |T|/(d+1)."""
    n, d = coords.shape
    vol, _ = simplex_geometry(coords, cells)
    r = np.zeros(n)
    np.add.at(r, cells.ravel(), np.repeat(cell_vals * vol / (d + 1), d + 1))
    return spla.splu(M.tocsc()).solve(r), r


def solve_elasticity_1d(L, nx, E, area, body_force, quantity="stress") -> OracleField:
    """_solve_elasticity_1d_static, fenics_mcp_server.py:1470-1587."""
    coords, cells = interval_mesh(nx, 0.0, L)
    n = nx + 1
    K, M, load = assemble_scalar(coords, cells)
    dir_mask = np.zeros(n, bool)
    dir_mask[0] = True
    u = solve_dirichlet(E * area * K, body_force * load, dir_mask, np.zeros(n))
    du = (u[1:] - u[:-1]) / (L / nx)                      # grad(u_sol)[0], constant per cell
    eps, _ = project_cellwise(coords, cells, du, M)       # :1542
    sig, _ = project_cellwise(coords, cells, E * du, M)   # :1546
    strain = quantity == "strain"
    c3 = np.zeros((n, 3))
    c3[:, 0] = coords[:, 0]
    meta = {"name": "axial_strain" if strain else "axial_stress", "unit": "-" if strain else "Pa",
            "pde": "elasticity_1d", "L": L, "E": E, "area": area, "body_force": body_force, "quantity": quantity}
    return OracleField(c3, (eps if strain else sig)[None, :], np.array([0.0]), 1, meta, {"displacement": u})


def _elasticity_nd(coords, cells, ix_of_node, lam, mu, body, quantity):
    n, d = coords.shape
    K = assemble_elasticity(coords, cells, lam, mu)
    _, Ms, load = assemble_scalar(coords, cells)
    b = (load[:, None] * np.asarray(body, float)[None, :]).ravel()
    dir_mask = np.repeat(ix_of_node == 0, d)              # clamp x = 0 (:1681-1684, :1831-1834)
    u = solve_dirichlet(K, b, dir_mask, np.zeros(n * d)).reshape(n, d)
    vm = von_mises(cell_strain(coords, cells, u), lam, mu, quantity)
    s, r = project_cellwise(coords, cells, vm, Ms)
    return s, u, vm, r


def solve_elasticity_2d(Lx, Ly, nx, ny, E, nu, body_fx=0.0, body_fy=0.0, quantity="stress",
                        plane_stress=True) -> OracleField:
    """_solve_elasticity_2d_static, fenics_mcp_server.py:1593-1743."""
    coords, cells = rectangle_mesh(Lx, Ly, nx, ny)
    n = coords.shape[0]
    lam, mu = lame(E, nu, plane_stress)
    s, u, vm, r = _elasticity_nd(coords, cells, np.arange(n) % (nx + 1), lam, mu, (body_fx, body_fy), quantity)
    strain = quantity == "strain"
    c3 = np.zeros((n, 3))
    c3[:, :2] = coords
    meta = {"name": "von_mises_strain" if strain else "von_mises_stress", "unit": "-" if strain else "Pa",
            "pde": "elasticity_2d", "Lx": Lx, "Ly": Ly, "E": E, "nu": nu, "body_fx": body_fx, "body_fy": body_fy,
            "quantity": quantity, "plane_stress": plane_stress}
    return OracleField(c3, s[None, :], np.array([0.0]), 2, meta, {"displacement": u, "cell_vm": vm, "proj_rhs": r})


def solve_elasticity_3d(Lx, Ly, Lz, nx, ny, nz, E, nu, body_fx=0.0, body_fy=0.0, body_fz=0.0,
                        quantity="stress") -> OracleField:
    """_solve_elasticity_3d_static, fenics_mcp_server.py:1749-1892."""
    coords, cells = box_mesh(Lx, Ly, Lz, nx, ny, nz)
    n = coords.shape[0]
    lam, mu = lame(E, nu, False)
    s, u, vm, r = _elasticity_nd(coords, cells, np.arange(n) % (nx + 1), lam, mu,
                                 (body_fx, body_fy, body_fz), quantity)
    strain = quantity == "strain"
    meta = {"name": "von_mises_strain" if strain else "von_mises_stress", "unit": "-" if strain else "Pa",
            "pde": "elasticity_3d", "Lx": Lx, "Ly": Ly, "Lz": Lz, "E": E, "nu": nu, "body_fx": body_fx,
            "body_fy": body_fy, "body_fz": body_fz, "quantity": quantity}
    return OracleField(coords, s[None, :], np.array([0.0]), 3, meta, {"displacement": u, "cell_vm": vm, "proj_rhs": r})


# ----------------------------------------------------------------------------------------------
# curvilinear heat tools (:769-1464): P1 on a box mesh in COORDINATE space with one scalar weight w
# multiplying both forms,  a = w u v dx + dt kappa w grad(u).grad(v) dx,  L = w u_n v dx + dt w f v dx.
# w is an Expression of degree 1 or 2, i.e. it enters the forms as its P1 / P2 interpolant on every cell
# and the products are integrated exactly (the form compiler picks the quadrature degree from the
# polynomial degrees).   kind: "r" = x0 (degree 1; :802, 1100), "r2" = x0^2 (degree 2; :965),
# "r2sin" = x0^2 sin(x1) (degree 2; :1230, 1371).
# ----------------------------------------------------------------------------------------------
WEIGHTS = {
    "r": (1, lambda X: X[:, 0]),
    "r2": (2, lambda X: X[:, 0] ** 2),
    "r2sin": (2, lambda X: X[:, 0] ** 2 * np.sin(X[:, 1])),
    "rcyl": (2, lambda X: np.sqrt(X[:, 1] ** 2 + X[:, 2] ** 2)),   # Expression("sqrt(x[1]*x[1] + x[2]*x[2])", degree=2) (:645)
    "one": (1, lambda X: np.ones(X.shape[0])),                      # r_expr = Constant(1.0) of the plain box (:647) — with a DG0 core
}


def _mono(d, idxs):
    """int over a unit-volume d-simplex of prod_{i in idxs} lambda_i = d! prod p! / (sum p + d)!"""
    counts = {}
    for i in idxs:
        counts[i] = counts.get(i, 0) + 1
    num = math.factorial(d)
    for p in counts.values():
        num *= math.factorial(p)
    return num / math.factorial(len(idxs) + d)


def _weight_terms(coords, cells, kind):
    """the interpolated weight of every cell as a list of (coefficient[nc], barycentric index tuple):
    degree 1: sum_v w_v l_v;  degree 2: sum_v w_v (2 l_v^2 - l_v) + sum_e w_e 4 l_a l_b"""
    degree, fn = WEIGHTS[kind]
    d = coords.shape[1]
    m = d + 1
    P = coords[cells]
    Xv = P.reshape(-1, d)
    if d < 2:                                   # the weight functions index x1 only for "r2sin" (d >= 2)
        Xv2 = np.concatenate([Xv, np.zeros((Xv.shape[0], 1))], axis=1)
    else:
        Xv2 = Xv
    wv = fn(Xv2).reshape(-1, m)
    terms = []
    if degree == 1:
        for v in range(m):
            terms.append((wv[:, v], (v,)))
        return terms
    for v in range(m):
        terms.append((2.0 * wv[:, v], (v, v)))
        terms.append((-wv[:, v], (v,)))
    for a, b in itertools.combinations(range(m), 2):
        mid = 0.5 * (P[:, a] + P[:, b])
        if d < 2:
            mid = np.concatenate([mid, np.zeros((mid.shape[0], 1))], axis=1)
        terms.append((4.0 * fn(mid), (a, b)))
    return terms


def assemble_weighted(coords, cells, kind, kcell=None):
    """K_w, M_w, load_w with the interpolated weight integrated exactly; kcell = per-cell factor of the stiffness
    form (DG0 diffusivity of the composite core, :554-572)"""
    n, d = coords.shape
    m = d + 1
    vol, grads = simplex_geometry(coords, cells)
    terms = _weight_terms(coords, cells, kind)
    w_int = sum(c * _mono(d, idx) for c, idx in terms) * vol                      # int_T w
    if kcell is not None:
        w_int = w_int * kcell
    Ke = w_int[:, None, None] * np.einsum("cad,cbd->cab", grads, grads)
    Me = np.zeros((cells.shape[0], m, m))
    le = np.zeros((cells.shape[0], m))
    for i in range(m):
        le[:, i] = sum(c * _mono(d, idx + (i,)) for c, idx in terms) * vol
        for j in range(m):
            Me[:, i, j] = sum(c * _mono(d, idx + (i, j)) for c, idx in terms) * vol
    K = _coo(cells, Ke, n)
    M = _coo(cells, Me, n)
    load = np.zeros(n)
    np.add.at(load, cells.ravel(), le.ravel())
    return K, M, load


def _weighted_heat_core(coords, cells, kind, dir_mask, dir_vals, kappa, T_initial, dt, num_steps, steady, source_type,
                        source_value, kcell=None, u0=None):
    K, M, load = assemble_weighted(coords, cells, kind, kcell)
    f = float(source_value) if source_type == "constant" else 0.0
    if steady:
        return solve_dirichlet(kappa * K, f * load, dir_mask, dir_vals)[None, :], np.array([0.0])
    # constant IC (:872-875), or the field `project(init_expr, V)` of the 3-D tool (:672-687; u0 given by the caller)
    u_n = np.where(dir_mask, dir_vals, float(T_initial) if u0 is None else u0)
    snaps, times = [u_n.copy()], [0.0]
    A = (M + dt * kappa * K).tocsr()
    for n in range(num_steps):
        u_n = solve_dirichlet(A, M @ u_n + dt * f * load, dir_mask, dir_vals)     # re-factorised every step
        snaps.append(u_n.copy())
        times.append((n + 1) * dt)
    return np.array(snaps), np.array(times)


def _radial_1d(kind, r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, steady, source_type,
               source_value, coord_sys, geom_names):
    coords, cells = interval_mesh(nr, r_inner, r_outer)
    n = nr + 1
    dir_mask = np.zeros(n, bool)
    dir_vals = np.zeros(n)
    if r_inner > 1e-10:                                                           # :811-815, :968-972
        dir_mask[0], dir_vals[0] = True, T_inner
    dir_mask[-1], dir_vals[-1] = True, T_outer
    vals, times = _weighted_heat_core(coords, cells, kind, dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps,
                                      steady, source_type, source_value)
    c3 = np.zeros((n, 3))
    c3[:, 0] = coords[:, 0]
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": coord_sys,
            "geometry_type": geom_names[0] if r_inner < 1e-10 else geom_names[1], "r_inner": r_inner, "r_outer": r_outer,
            "source_type": source_type, "source_value": source_value, "steady": steady}
    return OracleField(c3, vals, times, 1, meta)


def solve_heat_1d_cylindrical(r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, steady=False,
                              source_type="none", source_value=0.0, initial_type="constant", initial_amplitude=1.0):
    """_solve_heat_1d_cylindrical_raw, fenics_mcp_server.py:769-922 (w = r, degree 1)"""
    return _radial_1d("r", r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, steady, source_type,
                      source_value, "cylindrical", ("cylinder", "annulus"))


def solve_heat_1d_spherical(r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, steady=False,
                            source_type="none", source_value=0.0, initial_type="constant", initial_amplitude=1.0):
    """_solve_heat_1d_spherical_raw, :925-1060 (w = r^2, degree 2)"""
    return _radial_1d("r2", r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, steady, source_type,
                      source_value, "spherical", ("sphere", "spherical_shell"))


def shifted_rectangle_mesh(x0, x1, y0, y1, nx, ny):
    c, cells = rectangle_mesh(x1 - x0, y1 - y0, nx, ny)
    return c + np.array([x0, y0]), cells


def solve_heat_2d_cylindrical(r_inner, r_outer, z_length, nr, nz, diffusivity, T_boundary, T_initial, dt, num_steps,
                              steady=False, source_type="none", source_value=0.0, initial_type="constant",
                              initial_amplitude=1.0):
    """_solve_heat_2d_cylindrical_raw, :1063-1192: RectangleMesh((r_inner,0),(r_outer,z_length)), w = r; coords (r,0,z)"""
    coords, cells = shifted_rectangle_mesh(r_inner, r_outer, 0.0, z_length, nr, nz)
    n = coords.shape[0]
    ix, iy = np.arange(n) % (nr + 1), np.arange(n) // (nr + 1)
    dir_mask = (ix == 0) | (ix == nr) | (iy == 0) | (iy == nz)
    vals, times = _weighted_heat_core(coords, cells, "r", dir_mask, np.full(n, float(T_boundary)), diffusivity, T_initial, dt,
                                      num_steps, steady, source_type, source_value)
    c3 = np.stack([coords[:, 0], np.zeros(n), coords[:, 1]], axis=1)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cylindrical",
            "geometry_type": "cylinder" if r_inner < 1e-10 else "annular_cylinder", "r_inner": r_inner, "r_outer": r_outer,
            "z_length": z_length, "source_type": source_type, "source_value": source_value, "steady": steady}
    return OracleField(c3, vals, times, 2, meta)


def solve_heat_2d_spherical(r_inner, r_outer, nr, ntheta, diffusivity, T_boundary, T_initial, dt, num_steps, steady=False,
                            source_type="none", source_value=0.0, initial_type="constant", initial_amplitude=1.0):
    """_solve_heat_2d_spherical_raw, :1195-1330: RectangleMesh((r_inner,0),(r_outer,pi)), w = r^2 sin(theta);
    coords (r sin t, 0, r cos t)"""
    coords, cells = shifted_rectangle_mesh(r_inner, r_outer, 0.0, np.pi, nr, ntheta)
    n = coords.shape[0]
    ix, iy = np.arange(n) % (nr + 1), np.arange(n) // (nr + 1)
    dir_mask = (ix == 0) | (ix == nr) | (iy == 0) | (iy == ntheta)
    vals, times = _weighted_heat_core(coords, cells, "r2sin", dir_mask, np.full(n, float(T_boundary)), diffusivity, T_initial,
                                      dt, num_steps, steady, source_type, source_value)
    r, t = coords[:, 0], coords[:, 1]
    c3 = np.stack([r * np.sin(t), np.zeros(n), r * np.cos(t)], axis=1)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "spherical",
            "geometry_type": "sphere" if r_inner < 1e-10 else "spherical_shell", "r_inner": r_inner, "r_outer": r_outer,
            "source_type": source_type, "source_value": source_value, "steady": steady}
    return OracleField(c3, vals, times, 2, meta)


def solve_heat_3d_spherical(r_inner, r_outer, nr, ntheta, nphi, diffusivity, T_boundary, T_initial, dt, num_steps,
                            steady=False, source_type="none", source_value=0.0, initial_type="constant",
                            initial_amplitude=1.0):
    """_solve_heat_3d_spherical_raw, :1333-1464: BoxMesh((r_inner,0,0),(r_outer,pi,2pi)), w = r^2 sin(theta)"""
    coords, cells = box_mesh(r_outer - r_inner, np.pi, 2.0 * np.pi, nr, ntheta, nphi)
    coords = coords + np.array([r_inner, 0.0, 0.0])
    dir_mask, _ = box_dirichlet(nr, ntheta, nphi, T_boundary=T_boundary)
    n = coords.shape[0]
    vals, times = _weighted_heat_core(coords, cells, "r2sin", dir_mask, np.full(n, float(T_boundary)), diffusivity, T_initial,
                                      dt, num_steps, steady, source_type, source_value)
    r, t, ph = coords[:, 0], coords[:, 1], coords[:, 2]
    c3 = np.stack([r * np.sin(t) * np.cos(ph), r * np.sin(t) * np.sin(ph), r * np.cos(t)], axis=1)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "spherical",
            "geometry_type": "sphere" if r_inner < 1e-10 else "spherical_shell", "r_inner": r_inner, "r_outer": r_outer,
            "source_type": source_type, "source_value": source_value, "steady": steady}
    return OracleField(c3, vals, times, 3, meta)


def solve_heat_3d_cylinder_fallback(Lx, nx, ny, nz, diffusivity, T_boundary, T_initial, dt, num_steps, cylinder_radius,
                                    steady=False, source_type="none", source_value=0.0, T_left=None, T_right=None, T_side=None,
                                    core_radius=None, core_diffusivity=None, Ly=1.0, Lz=1.0, initial_type="constant",
                                    initial_amplitude=1.0, initial_wavenumber=1.0):
    """_solve_heat_3d_raw with geometry_type="cylinder" when mshr is not installed (it is in neither the reference's
    Dockerfile nor its requirements.txt, so this IS what the reference runs): fenics_mcp_server.py:523-529 BoxMesh
    (0,-R,-R)-(Lx,R,R) with int(ny*2R) x int(nz*2R) cells across, :642-645 weight sqrt(y^2+z^2) as a degree-2
    Expression, :537-572 DG0 diffusivity (a cell is core when all its vertices and its midpoint lie within
    core_radius of the axis — SubDomain.mark), :582-604 directional BCs.  On this box mesh no boundary facet has all
    its vertices at distance exactly R from the axis, so the `T_side` condition (:590-594, topological DirichletBC)
    constrains nothing; without directional values every boundary node gets T_boundary (:626-628).
    Sine / cosine initial conditions (:672-687) are `project(Expression(degree=2), V)` with the plain dx measure (project does not
    know about the weight), evaluated at the shifted mesh coordinates."""
    R = float(cylinder_radius)
    ny2, nz2 = int(ny * R * 2), int(nz * R * 2)
    if ny2 < 1 or nz2 < 1:
        raise ValueError("BoxMesh needs at least one cell per axis (int(ny*2R), int(nz*2R))")
    coords, cells = box_mesh(Lx, 2 * R, 2 * R, nx, ny2, nz2)
    coords = coords + np.array([0.0, -R, -R])
    n = coords.shape[0]
    directional = T_left is not None or T_right is not None or T_side is not None
    if directional:
        dir_mask, dir_vals = box_dirichlet(nx, ny2, nz2, T_left=T_left, T_right=T_right, T_side=None)
    else:
        dir_mask, dir_vals = box_dirichlet(nx, ny2, nz2, T_boundary=T_boundary)
    kcell = None
    composite = core_radius is not None and core_diffusivity is not None
    if composite:
        kcell = np.where(core_cells(coords, cells, core_radius), core_diffusivity / diffusivity, 1.0)
    u0 = None
    if not steady:
        if initial_type in ("cosine", "sine"):
            _, M, _ = assemble_scalar(coords, cells)
            u0 = _initial_condition(coords, cells, M, initial_type, T_initial, initial_amplitude, initial_wavenumber)
        elif initial_type == "zero":
            u0 = np.zeros(n)
    vals, times = _weighted_heat_core(coords, cells, "rcyl", dir_mask, dir_vals, diffusivity, T_initial, dt, num_steps, steady,
                                      source_type, source_value, kcell, u0)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cylindrical", "Lx": Lx, "Ly": R * 2,
            "Lz": R * 2, "geometry_type": "cylinder", "source_type": source_type, "source_value": source_value, "steady": steady,
            "cylinder_radius": R}
    if directional:
        for k, v in (("T_left", T_left), ("T_right", T_right), ("T_side", T_side)):
            if v is not None:
                meta[k] = v
    else:
        meta["T_boundary"] = T_boundary
    if composite:
        meta.update({"core_radius": core_radius, "core_diffusivity": core_diffusivity, "base_diffusivity": diffusivity})
    else:
        meta["diffusivity"] = diffusivity
    return OracleField(coords, vals, times, 3, meta)
