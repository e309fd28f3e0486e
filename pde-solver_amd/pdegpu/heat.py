"""Host-side mirror of the reference's Cartesian heat solvers — same function names, argument meaning
and outputs as `_solve_heat_{1,2,3}d_raw` (fenics_mcp_server.py:204-338, 345-468, 475-762), but the
mesh / FunctionSpace / DirichletBC / `solve(a == L, u, bcs)` block is replaced by one call through the
C-ABI into the HIP library (include/pdegpu.h: pdegpu_heat_solve)."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import _lib as L
from .fields import TimeSeriesField, grid_coords
from .project import p2_expression_load, trig_product


def _env_float(name, default):
    v = os.environ.get(name)
    return float(v) if v else default


def _env_int(name, default):
    v = os.environ.get(name)
    return int(v) if v else default


def run_heat(dim, n, Ls, *, kappa, dt, num_steps, steady, source_type, source_value, initial_type,
             T_initial, initial_amplitude, initial_wavenumber, bc_mode, bc_value, bc_present,
             snapshot_stride=None, rtol=None, max_iter=None, fast_path=True, element_mode=None, device=None):
    """Shared body of the three solvers: fills pdegpu_grid / pdegpu_heat, calls pdegpu_heat_solve and
    returns (values [n_snap,N], times [n_snap], stats dict)."""
    lib = L.load_library()
    ctx = L.context(device)
    if element_mode is None:
        element_mode = L.element_mode_from_env()
    grid = L.make_grid(dim, n, Ls, element_mode)
    h = L.Heat()
    h.kappa = float(kappa)
    h.dt = float(dt)
    h.num_steps = int(num_steps)
    h.steady = 1 if steady else 0
    h.source = float(source_value) if source_type == "constant" else 0.0       # :244-247, :379-382, :631-634
    h.bc_mode = bc_mode
    for r in range(3):
        h.bc_value[r] = float(bc_value[r])
        h.bc_present[r] = int(bc_present[r])
    h.rtol = float(rtol if rtol is not None else _env_float("PDEGPU_RTOL", 1e-10))
    h.max_iter = int(max_iter if max_iter is not None else _env_int("PDEGPU_MAX_ITER", 200000))
    h.fast_path = 1 if fast_path else 0
    N = int(lib.pdegpu_grid_num_nodes(grid))
    if N < 0:
        raise L.PdeGpuError(L.EINVAL, f"invalid grid dim={dim} n={tuple(n)} (every axis needs >= 1 cell)")
    # output volume: the stride actually used (argument > PDEGPU_SNAPSHOT_STRIDE > 1, widened when the states would not fit the host
    # budget) travels back in the stats dict and is what meta records
    stride, stride_meta = L.snapshot_policy(int(num_steps), N, snapshot_stride if snapshot_stride is not None else
                                            _env_int("PDEGPU_SNAPSHOT_STRIDE", 1), bool(steady))
    h.snapshot_stride = stride

    u0 = None
    if steady:
        h.ic_type = L.IC_CONSTANT
    elif initial_type == "zero":                                              # :276-277
        h.ic_type = L.IC_ZERO
    elif initial_type in ("cosine", "sine"):                                  # :278-290: project(Expression(degree=2), V)
        rhs = p2_expression_load(dim, n, Ls, trig_product(initial_type, initial_amplitude, initial_wavenumber, dim),
                                 element_mode)
        u0 = np.empty(N)
        st = L.Stats()
        L.check(ctx, lib.pdegpu_mass_solve(ctx, grid, L.dptr(rhs), L.dptr(u0), 1e-13, 20000, st))
        h.ic_type = L.IC_NODAL
    else:                                                                     # "constant" or anything else (:291-292)
        h.ic_type = L.IC_CONSTANT
        h.ic_value = float(T_initial)

    n_snap = lib.pdegpu_heat_num_snapshots(h)
    values = np.empty((n_snap, N))
    times = np.empty(n_snap)
    stats = L.Stats()
    L.check(ctx, lib.pdegpu_heat_solve(ctx, grid, h, L.dptr(u0) if u0 is not None else None, L.dptr(values),
                                       L.dptr(times), stats))
    st = stats.as_dict()
    st["snapshot_meta"] = stride_meta
    return values, times, st


def _extra_meta(stats, element_mode):
    """Build-side additions go into meta only."""
    m = {"backend": "pdegpu-hip-gfx950", "element_mode": "q1" if element_mode == L.ELEM_Q1 else "p1split",
         "cg_iterations": int(stats["cg_iterations"]), "solver_path": stats["path"], "solve_ms": float(stats["solve_ms"])}
    m.update(stats.get("snapshot_meta", {"snapshot_stride": 1}))
    return m


def _solve_heat_1d_raw(length: float, nx: int, diffusivity: float, T_left: float, T_right: float, T_initial: float,
                       dt: float, num_steps: int, steady: bool = False, source_type: str = "none",
                       source_value: float = 0.0, initial_type: str = "constant", initial_amplitude: float = 1.0,
                       initial_wavenumber: float = 1.0, **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:204-338.  BCs T_left at x=0, T_right at x=L (:233-241); output ascending in x
    (:252-254) — our node order already is."""
    em = opts.get("element_mode", L.element_mode_from_env())
    values, times, st = run_heat(1, (nx,), (length,), kappa=diffusivity, dt=dt, num_steps=num_steps, steady=steady,
                                 source_type=source_type, source_value=source_value, initial_type=initial_type,
                                 T_initial=T_initial, initial_amplitude=initial_amplitude,
                                 initial_wavenumber=initial_wavenumber, bc_mode=L.BC_DIRECTIONAL,
                                 bc_value=(T_left, T_right, 0.0), bc_present=(1, 1, 0), **opts)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian", "length": length,
            "source_type": source_type, "source_value": source_value, "steady": steady}
    meta.update(_extra_meta(st, em))
    return TimeSeriesField(coords=grid_coords(1, (nx,), (length,)), values=values, times=times, dim=1, meta=meta)


def _solve_heat_2d_raw(Lx: float, Ly: float, nx: int, ny: int, diffusivity: float, T_boundary: float,
                       T_initial: float, dt: float, num_steps: int, steady: bool = False, source_type: str = "none",
                       source_value: float = 0.0, initial_type: str = "constant", initial_amplitude: float = 1.0,
                       initial_wavenumber: float = 1.0, **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:345-468.  One constant Dirichlet value on the whole boundary (:373-376)."""
    em = opts.get("element_mode", L.element_mode_from_env())
    values, times, st = run_heat(2, (nx, ny), (Lx, Ly), kappa=diffusivity, dt=dt, num_steps=num_steps, steady=steady,
                                 source_type=source_type, source_value=source_value, initial_type=initial_type,
                                 T_initial=T_initial, initial_amplitude=initial_amplitude,
                                 initial_wavenumber=initial_wavenumber, bc_mode=L.BC_UNIFORM,
                                 bc_value=(T_boundary, 0.0, 0.0), bc_present=(1, 0, 0), **opts)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian", "Lx": Lx, "Ly": Ly,
            "source_type": source_type, "source_value": source_value, "steady": steady}
    meta.update(_extra_meta(st, em))
    return TimeSeriesField(coords=grid_coords(2, (nx, ny), (Lx, Ly)), values=values, times=times, dim=2, meta=meta)


def _solve_heat_3d_raw(Lx: float, Ly: float, Lz: float, nx: int, ny: int, nz: int, diffusivity: float,
                       T_boundary: float, T_initial: float, dt: float, num_steps: int, steady: bool = False,
                       source_type: str = "none", source_value: float = 0.0, initial_type: str = "constant",
                       initial_amplitude: float = 1.0, initial_wavenumber: float = 1.0, geometry_type: str = "box",
                       cylinder_radius: Optional[float] = None, T_left: Optional[float] = None,
                       T_right: Optional[float] = None, T_side: Optional[float] = None,
                       core_radius: Optional[float] = None, core_diffusivity: Optional[float] = None,
                       **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:475-762: box branch (:533, 574, 606-628, 647, 653-716), its composite core (:537-572) and the
    BoxMesh fallback of the cylinder (:523-529) with every initial condition; only a true mshr cylinder mesh (:512-521,
    unstructured; mshr is in neither the reference's Dockerfile nor its requirements) is outside the path."""
    if geometry_type == "cylinder" and cylinder_radius is not None:
        # :512-530 — the reference meshes a true cylinder with mshr when it is installed; it is in neither its Dockerfile
        # nor its requirements.txt, so what actually runs is the BoxMesh fallback with the sqrt(y^2+z^2) weight
        from .curvilinear import solve_heat_3d_cylinder_fallback
        return solve_heat_3d_cylinder_fallback(Lx, nx, ny, nz, diffusivity, T_boundary, 0.0 if initial_type == "zero" else T_initial,
                                               dt, num_steps, cylinder_radius, steady=steady, source_type=source_type,
                                               source_value=source_value, T_left=T_left, T_right=T_right, T_side=T_side,
                                               core_radius=core_radius, core_diffusivity=core_diffusivity,
                                               initial_type=initial_type, initial_amplitude=initial_amplitude,
                                               initial_wavenumber=initial_wavenumber, **opts)
    if core_radius is not None and core_diffusivity is not None:                        # :537-572 on the plain box
        from .curvilinear import solve_heat_3d_box_core
        return solve_heat_3d_box_core(Lx, Ly, Lz, nx, ny, nz, diffusivity, T_boundary, T_initial, dt, num_steps, core_radius,
                                      core_diffusivity, steady=steady, source_type=source_type, source_value=source_value,
                                      initial_type=initial_type, initial_amplitude=initial_amplitude,
                                      initial_wavenumber=initial_wavenumber, T_left=T_left, T_right=T_right, T_side=T_side, **opts)
    directional = T_left is not None or T_right is not None or T_side is not None      # :578
    if directional:                                                                     # :606-623
        bc_mode = L.BC_DIRECTIONAL
        bc_value = tuple(0.0 if v is None else float(v) for v in (T_left, T_right, T_side))
        bc_present = tuple(0 if v is None else 1 for v in (T_left, T_right, T_side))
    else:                                                                               # :626-628
        bc_mode, bc_value, bc_present = L.BC_UNIFORM, (T_boundary, 0.0, 0.0), (1, 0, 0)
    em = opts.get("element_mode", L.element_mode_from_env())
    values, times, st = run_heat(3, (nx, ny, nz), (Lx, Ly, Lz), kappa=diffusivity, dt=dt, num_steps=num_steps,
                                 steady=steady, source_type=source_type, source_value=source_value,
                                 initial_type=initial_type, T_initial=T_initial, initial_amplitude=initial_amplitude,
                                 initial_wavenumber=initial_wavenumber, bc_mode=bc_mode, bc_value=bc_value,
                                 bc_present=bc_present, **opts)
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian",          # :721-733
            "Lx": Lx, "Ly": Ly, "Lz": Lz, "geometry_type": geometry_type, "source_type": source_type,
            "source_value": source_value, "steady": steady}
    if directional:                                                                     # :738-746
        for k, v in (("T_left", T_left), ("T_right", T_right), ("T_side", T_side)):
            if v is not None:
                meta[k] = v
    else:
        meta["T_boundary"] = T_boundary
    meta["diffusivity"] = diffusivity                                                   # :753
    meta.update(_extra_meta(st, em))
    return TimeSeriesField(coords=grid_coords(3, (nx, ny, nz), (Lx, Ly, Lz)), values=values, times=times, dim=3,
                           meta=meta)
