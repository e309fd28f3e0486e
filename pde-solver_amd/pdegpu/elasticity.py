"""Host-side mirror of the reference's static elasticity solvers — same names, arguments and outputs as
`_solve_elasticity_{1,2,3}d_static` (fenics_mcp_server.py:1470-1587, 1593-1743, 1749-1892); the DOLFIN block
(VectorFunctionSpace, clamp at x=0, solve, von Mises, project) is one C-ABI call: pdegpu_elasticity_solve."""
from __future__ import annotations

import os

import numpy as np

from . import _lib as L
from .fields import TimeSeriesField, grid_coords


def run_elasticity(dim, n, Ls, *, E, nu=0.3, area=1.0, body=(0.0, 0.0, 0.0), plane_stress=True, quantity="stress",
                   rtol=None, max_iter=None, want_displacement=False, device=None):
    lib = L.load_library()
    ctx = L.context(device)
    grid = L.make_grid(dim, n, Ls, L.ELEM_P1SPLIT)
    N = int(lib.pdegpu_grid_num_nodes(grid))
    if N < 0:
        raise L.PdeGpuError(L.EINVAL, f"invalid grid dim={dim} n={tuple(n)} (every axis needs >= 1 cell)")
    el = L.Elasticity()
    el.E, el.nu, el.area = float(E), float(nu), float(area)
    for r in range(3):
        el.body[r] = float(body[r]) if r < len(body) else 0.0
    el.plane_stress = 1 if plane_stress else 0
    el.quantity = L.QUANTITY_STRAIN if quantity == "strain" else L.QUANTITY_STRESS     # anything else -> stress (:1553, :1707)
    el.rtol = float(rtol) if rtol is not None else float(os.environ.get("PDEGPU_RTOL", 1e-10))
    el.max_iter = int(max_iter) if max_iter is not None else int(os.environ.get("PDEGPU_MAX_ITER", 400000))
    field = np.empty(N)
    disp = np.empty((N, dim)) if want_displacement else None
    st = L.Stats()
    L.check(ctx, lib.pdegpu_elasticity_solve(ctx, grid, el, L.dptr(field), L.dptr(disp) if disp is not None else None, st))
    return field, disp, st.as_dict()


def _extra(st):
    return {"backend": "pdegpu-hip-gfx950", "element_mode": "p1split", "cg_iterations": int(st["cg_iterations"]),
            "solver_path": st.get("path", "general"), "solve_ms": float(st.get("solve_ms", 0.0))}


def _solve_elasticity_1d_static(L_: float, nx: int, E: float, area: float, body_force: float,
                                quantity: str = "stress", **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:1470-1587 (the first parameter is named `L` there)."""
    field, disp, st = run_elasticity(1, (nx,), (L_,), E=E, area=area, body=(body_force,), quantity=quantity, **opts)
    strain = quantity == "strain"
    meta = {"name": "axial_strain" if strain else "axial_stress", "unit": "-" if strain else "Pa",
            "pde": "elasticity_1d", "L": L_, "E": E, "area": area, "body_force": body_force, "quantity": quantity}
    meta.update(_extra(st))
    f = TimeSeriesField(coords=grid_coords(1, (nx,), (L_,)), values=field[None, :], times=np.array([0.0]), dim=1, meta=meta)
    f.displacement = disp
    return f


def _solve_elasticity_2d_static(Lx: float, Ly: float, nx: int, ny: int, E: float, nu: float, body_fx: float = 0.0,
                                body_fy: float = 0.0, quantity: str = "stress", plane_stress: bool = True,
                                **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:1593-1743."""
    field, disp, st = run_elasticity(2, (nx, ny), (Lx, Ly), E=E, nu=nu, body=(body_fx, body_fy),
                                     plane_stress=plane_stress, quantity=quantity, **opts)
    strain = quantity == "strain"
    meta = {"name": "von_mises_strain" if strain else "von_mises_stress", "unit": "-" if strain else "Pa",
            "pde": "elasticity_2d", "Lx": Lx, "Ly": Ly, "E": E, "nu": nu, "body_fx": body_fx, "body_fy": body_fy,
            "quantity": quantity, "plane_stress": plane_stress}
    meta.update(_extra(st))
    f = TimeSeriesField(coords=grid_coords(2, (nx, ny), (Lx, Ly)), values=field[None, :], times=np.array([0.0]), dim=2,
                        meta=meta)
    f.displacement = disp
    return f


def _solve_elasticity_3d_static(Lx: float, Ly: float, Lz: float, nx: int, ny: int, nz: int, E: float, nu: float,
                                body_fx: float = 0.0, body_fy: float = 0.0, body_fz: float = 0.0,
                                quantity: str = "stress", **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:1749-1892."""
    field, disp, st = run_elasticity(3, (nx, ny, nz), (Lx, Ly, Lz), E=E, nu=nu, body=(body_fx, body_fy, body_fz),
                                     quantity=quantity, **opts)
    strain = quantity == "strain"
    meta = {"name": "von_mises_strain" if strain else "von_mises_stress", "unit": "-" if strain else "Pa",
            "pde": "elasticity_3d", "Lx": Lx, "Ly": Ly, "Lz": Lz, "E": E, "nu": nu, "body_fx": body_fx,
            "body_fy": body_fy, "body_fz": body_fz, "quantity": quantity}
    meta.update(_extra(st))
    f = TimeSeriesField(coords=grid_coords(3, (nx, ny, nz), (Lx, Ly, Lz)), values=field[None, :], times=np.array([0.0]),
                        dim=3, meta=meta)
    f.displacement = disp
    return f
