"""Host-side mirror of the reference's curvilinear heat solvers — same names, arguments and outputs as
`_solve_heat_1d_cylindrical_raw` (fenics_mcp_server.py:769-922), `_solve_heat_1d_spherical_raw` (:925-1060),
`_solve_heat_2d_cylindrical_raw` (:1063-1192), `_solve_heat_2d_spherical_raw` (:1195-1330) and
`_solve_heat_3d_spherical_raw` (:1333-1464).  Each is P1 on a box mesh in coordinate space with one weight
(r, r^2 or r^2 sin(theta)) multiplying both forms; the DOLFIN block is one C-ABI call, pdegpu_heat_solve_weighted."""
from __future__ import annotations

import math
import os

import numpy as np

from . import _lib as L
from .fields import GridCoords, TimeSeriesField


def run_weighted(dim, n, lengths, origin, kind, *, kappa, dt, num_steps, steady, source_type, source_value, T_initial,
                 bc_mode, bc_value, bc_present, core_radius=0.0, core_ratio=1.0, rtol=None, max_iter=None,
                 snapshot_stride=None, device=None, u0=None):
    lib = L.load_library()
    ctx = L.context(device)
    grid = L.make_grid(dim, n, lengths, L.ELEM_P1SPLIT)
    N = int(lib.pdegpu_grid_num_nodes(grid))
    if N < 0:
        raise L.PdeGpuError(L.EINVAL, f"invalid grid dim={dim} n={tuple(n)} lengths={tuple(lengths)}")
    w = L.Weight()
    w.kind = kind
    for r in range(3):
        w.origin[r] = float(origin[r]) if r < len(origin) else 0.0
    w.core_radius, w.core_ratio = float(core_radius), float(core_ratio)
    h = L.Heat()
    h.kappa, h.dt, h.num_steps, h.steady = float(kappa), float(dt), int(num_steps), 1 if steady else 0
    h.source = float(source_value) if source_type == "constant" else 0.0
    h.ic_type, h.ic_value = L.IC_CONSTANT, float(T_initial)          # every initial_type assigns the constant (:872-875)
    h.bc_mode = bc_mode
    for r in range(3):
        h.bc_value[r], h.bc_present[r] = float(bc_value[r]), int(bc_present[r])
    stride, stride_meta = L.snapshot_policy(int(num_steps), N, int(snapshot_stride if snapshot_stride is not None else
                                                                   os.environ.get("PDEGPU_SNAPSHOT_STRIDE", 1)), bool(steady))
    h.snapshot_stride = stride
    h.rtol = float(rtol if rtol is not None else os.environ.get("PDEGPU_RTOL", 1e-10))
    h.max_iter = int(max_iter if max_iter is not None else os.environ.get("PDEGPU_MAX_ITER", 400000))
    h.fast_path = 0
    n_snap = lib.pdegpu_heat_num_snapshots(h)
    values, times = np.empty((n_snap, N)), np.empty(n_snap)
    st = L.Stats()
    if u0 is not None and not steady:                                 # projected initial field (3-D tool, :672-687)
        h.ic_type = L.IC_NODAL
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
    L.check(ctx, lib.pdegpu_heat_solve_weighted_ic(ctx, grid, w, h, L.dptr(u0) if (u0 is not None and not steady) else None,
                                                   L.dptr(values), L.dptr(times), st))
    d = st.as_dict()
    d["snapshot_meta"] = stride_meta
    return values, times, d


def _axes(n, origin, lengths):
    return [origin[r] + lengths[r] * np.arange(n[r] + 1) / n[r] for r in range(len(n))]


def _meta(coord_sys, geom, r_inner, r_outer, source_type, source_value, steady, st, **extra):
    m = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": coord_sys, "geometry_type": geom,
         "r_inner": r_inner, "r_outer": r_outer}
    m.update(extra)
    m.update({"source_type": source_type, "source_value": source_value, "steady": steady})
    m.update({"backend": "pdegpu-hip-gfx950", "element_mode": "p1split", "cg_iterations": int(st["cg_iterations"]),
              "solver_path": st.get("path", "weighted"), "solve_ms": float(st.get("solve_ms", 0.0))})
    m.update(st.get("snapshot_meta", {"snapshot_stride": 1}))
    return m


def _radial_1d(kind, coord_sys, names, r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, steady,
               source_type, source_value, **opts):
    inner = 1 if r_inner > 1e-10 else 0                               # inner BC only for r_inner > 0 (:811, :968)
    values, times, st = run_weighted(1, (nr,), (r_outer - r_inner,), (r_inner,), kind, kappa=diffusivity, dt=dt,
                                     num_steps=num_steps, steady=steady, source_type=source_type, source_value=source_value,
                                     T_initial=T_initial, bc_mode=L.BC_DIRECTIONAL, bc_value=(T_inner, T_outer, 0.0),
                                     bc_present=(inner, 1, 0), **opts)
    r = _axes((nr,), (r_inner,), (r_outer - r_inner,))[0]
    coords = np.stack([r, np.zeros_like(r), np.zeros_like(r)], axis=1)
    meta = _meta(coord_sys, names[0] if r_inner < 1e-10 else names[1], r_inner, r_outer, source_type, source_value, steady, st)
    return TimeSeriesField(coords=coords, values=values, times=times, dim=1, meta=meta)


def _solve_heat_1d_cylindrical_raw(r_inner: float, r_outer: float, nr: int, diffusivity: float, T_inner: float, T_outer: float,
                                   T_initial: float, dt: float, num_steps: int, steady: bool = False, source_type: str = "none",
                                   source_value: float = 0.0, initial_type: str = "constant", initial_amplitude: float = 1.0,
                                   **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:769-922: IntervalMesh(nr, r_inner, r_outer), weight r"""
    return _radial_1d(L.WEIGHT_R, "cylindrical", ("cylinder", "annulus"), r_inner, r_outer, nr, diffusivity, T_inner, T_outer,
                      T_initial, dt, num_steps, steady, source_type, source_value, **opts)


def _solve_heat_1d_spherical_raw(r_inner: float, r_outer: float, nr: int, diffusivity: float, T_inner: float, T_outer: float,
                                 T_initial: float, dt: float, num_steps: int, steady: bool = False, source_type: str = "none",
                                 source_value: float = 0.0, initial_type: str = "constant", initial_amplitude: float = 1.0,
                                 **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:925-1060: weight r^2"""
    return _radial_1d(L.WEIGHT_R2, "spherical", ("sphere", "spherical_shell"), r_inner, r_outer, nr, diffusivity, T_inner,
                      T_outer, T_initial, dt, num_steps, steady, source_type, source_value, **opts)


def _uniform(dim, n, lengths, origin, kind, diffusivity, T_boundary, T_initial, dt, num_steps, steady, source_type, source_value,
             **opts):
    return run_weighted(dim, n, lengths, origin, kind, kappa=diffusivity, dt=dt, num_steps=num_steps, steady=steady,
                        source_type=source_type, source_value=source_value, T_initial=T_initial, bc_mode=L.BC_UNIFORM,
                        bc_value=(T_boundary, 0.0, 0.0), bc_present=(1, 0, 0), **opts)


def _solve_heat_2d_cylindrical_raw(r_inner: float, r_outer: float, z_length: float, nr: int, nz: int, diffusivity: float,
                                   T_boundary: float, T_initial: float, dt: float, num_steps: int, steady: bool = False,
                                   source_type: str = "none", source_value: float = 0.0, initial_type: str = "constant",
                                   initial_amplitude: float = 1.0, **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:1063-1192: RectangleMesh((r_inner,0),(r_outer,z_length)), weight r, coords (r, 0, z)"""
    values, times, st = _uniform(2, (nr, nz), (r_outer - r_inner, z_length), (r_inner, 0.0), L.WEIGHT_R, diffusivity, T_boundary,
                                 T_initial, dt, num_steps, steady, source_type, source_value, **opts)
    r, z = _axes((nr, nz), (r_inner, 0.0), (r_outer - r_inner, z_length))
    Z, R = np.meshgrid(z, r, indexing="ij")
    coords = np.stack([R.ravel(), np.zeros(R.size), Z.ravel()], axis=1)
    meta = _meta("cylindrical", "cylinder" if r_inner < 1e-10 else "annular_cylinder", r_inner, r_outer, source_type,
                 source_value, steady, st, z_length=z_length)
    return TimeSeriesField(coords=coords, values=values, times=times, dim=2, meta=meta)


def _solve_heat_2d_spherical_raw(r_inner: float, r_outer: float, nr: int, ntheta: int, diffusivity: float, T_boundary: float,
                                 T_initial: float, dt: float, num_steps: int, steady: bool = False, source_type: str = "none",
                                 source_value: float = 0.0, initial_type: str = "constant", initial_amplitude: float = 1.0,
                                 **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:1195-1330: RectangleMesh((r_inner,0),(r_outer,pi)), weight r^2 sin(theta); coords
    (r sin t, 0, r cos t)"""
    values, times, st = _uniform(2, (nr, ntheta), (r_outer - r_inner, math.pi), (r_inner, 0.0), L.WEIGHT_R2SIN, diffusivity,
                                 T_boundary, T_initial, dt, num_steps, steady, source_type, source_value, **opts)
    r, t = _axes((nr, ntheta), (r_inner, 0.0), (r_outer - r_inner, math.pi))
    T, R = np.meshgrid(t, r, indexing="ij")
    R, T = R.ravel(), T.ravel()
    coords = np.stack([R * np.sin(T), np.zeros(R.size), R * np.cos(T)], axis=1)
    meta = _meta("spherical", "sphere" if r_inner < 1e-10 else "spherical_shell", r_inner, r_outer, source_type, source_value,
                 steady, st)
    return TimeSeriesField(coords=coords, values=values, times=times, dim=2, meta=meta)


def _solve_heat_3d_spherical_raw(r_inner: float, r_outer: float, nr: int, ntheta: int, nphi: int, diffusivity: float,
                                 T_boundary: float, T_initial: float, dt: float, num_steps: int, steady: bool = False,
                                 source_type: str = "none", source_value: float = 0.0, initial_type: str = "constant",
                                 initial_amplitude: float = 1.0, **opts) -> TimeSeriesField:
    """fenics_mcp_server.py:1333-1464: BoxMesh((r_inner,0,0),(r_outer,pi,2pi)), weight r^2 sin(theta)"""
    values, times, st = _uniform(3, (nr, ntheta, nphi), (r_outer - r_inner, math.pi, 2.0 * math.pi), (r_inner, 0.0, 0.0),
                                 L.WEIGHT_R2SIN, diffusivity, T_boundary, T_initial, dt, num_steps, steady, source_type,
                                 source_value, **opts)
    r, t, ph = _axes((nr, ntheta, nphi), (r_inner, 0.0, 0.0), (r_outer - r_inner, math.pi, 2.0 * math.pi))
    PH, T, R = np.meshgrid(ph, t, r, indexing="ij")
    R, T, PH = R.ravel(), T.ravel(), PH.ravel()
    coords = np.stack([R * np.sin(T) * np.cos(PH), R * np.sin(T) * np.sin(PH), R * np.cos(T)], axis=1)
    meta = _meta("spherical", "sphere" if r_inner < 1e-10 else "spherical_shell", r_inner, r_outer, source_type, source_value,
                 steady, st)
    return TimeSeriesField(coords=coords, values=values, times=times, dim=3, meta=meta)


def projected_trig_field(n, lengths, origin, initial_type, amplitude, wavenumber, device=None):
    """`project(Expression(A cos|sin(k x) cos|sin(k y) cos|sin(k z), degree=2), V)` on the BoxMesh [origin, origin + lengths]
    (fenics_mcp_server.py:672-687): P2-interpolated load on the host, consistent-mass solve on the GPU.  project() uses the plain dx
    measure whatever the weight of the heat forms is."""
    from .project import p2_expression_load, trig_product
    lib = L.load_library()
    ctx = L.context(device)
    grid = L.make_grid(3, n, lengths, L.ELEM_P1SPLIT)
    rhs = p2_expression_load(3, n, lengths, trig_product(initial_type, amplitude, wavenumber, 3), L.ELEM_P1SPLIT, origin=origin)
    u0 = np.empty(rhs.shape[0])
    st = L.Stats()
    L.check(ctx, lib.pdegpu_mass_solve(ctx, grid, L.dptr(rhs), L.dptr(u0), 1e-13, 20000, st))
    return u0


def solve_heat_3d_box_core(Lx, Ly, Lz, nx, ny, nz, diffusivity, T_boundary, T_initial, dt, num_steps, core_radius, core_diffusivity,
                           steady=False, source_type="none", source_value=0.0, initial_type="constant", initial_amplitude=1.0,
                           initial_wavenumber=1.0, T_left=None, T_right=None, T_side=None, **opts) -> TimeSeriesField:
    """`_solve_heat_3d_raw` on the plain box WITH a composite core (fenics_mcp_server.py:537-572): DG0 diffusivity, a simplex is
    core when all its vertices and its midpoint lie within core_radius of the x-axis — on BoxMesh((0,0,0),(Lx,Ly,Lz)) that is the
    quarter cylinder around the edge y = z = 0.  Unit weight (:647), every boundary-condition and initial-condition variant of the
    box branch."""
    directional = T_left is not None or T_right is not None or T_side is not None
    if directional:
        bc_mode = L.BC_DIRECTIONAL
        bc_value = tuple(0.0 if v is None else float(v) for v in (T_left, T_right, T_side))
        bc_present = tuple(0 if v is None else 1 for v in (T_left, T_right, T_side))
    else:
        bc_mode, bc_value, bc_present = L.BC_UNIFORM, (T_boundary, 0.0, 0.0), (1, 0, 0)
    u0 = None
    if not steady and initial_type in ("cosine", "sine"):
        u0 = projected_trig_field((nx, ny, nz), (Lx, Ly, Lz), (0.0, 0.0, 0.0), initial_type, initial_amplitude, initial_wavenumber,
                                  opts.get("device"))
    values, times, st = run_weighted(3, (nx, ny, nz), (Lx, Ly, Lz), (0.0, 0.0, 0.0), L.WEIGHT_ONE, kappa=diffusivity, dt=dt,
                                     num_steps=num_steps, steady=steady, source_type=source_type, source_value=source_value,
                                     T_initial=0.0 if initial_type == "zero" else T_initial, bc_mode=bc_mode, bc_value=bc_value,
                                     bc_present=bc_present, core_radius=core_radius, core_ratio=core_diffusivity / diffusivity,
                                     u0=u0, **opts)
    coords = GridCoords(3, (nx, ny, nz), (Lx, Ly, Lz))
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cartesian", "Lx": Lx, "Ly": Ly, "Lz": Lz,
            "geometry_type": "box", "source_type": source_type, "source_value": source_value, "steady": steady}   # :721-733
    if directional:
        for k, v in (("T_left", T_left), ("T_right", T_right), ("T_side", T_side)):
            if v is not None:
                meta[k] = v
    else:
        meta["T_boundary"] = T_boundary
    meta.update({"core_radius": core_radius, "core_diffusivity": core_diffusivity, "base_diffusivity": diffusivity})   # :747-750
    meta.update({"backend": "pdegpu-hip-gfx950", "element_mode": "p1split", "cg_iterations": int(st["cg_iterations"]),
                 "solver_path": st.get("path", "general"), "solve_ms": float(st.get("solve_ms", 0.0))})
    meta.update(st.get("snapshot_meta", {"snapshot_stride": 1}))
    return TimeSeriesField(coords=coords, values=values, times=times, dim=3, meta=meta)


def solve_heat_3d_cylinder_fallback(Lx, nx, ny, nz, diffusivity, T_boundary, T_initial, dt, num_steps, cylinder_radius,
                                    steady=False, source_type="none", source_value=0.0, T_left=None, T_right=None, T_side=None,
                                    core_radius=None, core_diffusivity=None, initial_type="constant", initial_amplitude=1.0,
                                    initial_wavenumber=1.0, **opts) -> TimeSeriesField:
    """`_solve_heat_3d_raw(geometry_type="cylinder")` as the reference runs it without mshr (which is in neither its
    Dockerfile nor its requirements): fenics_mcp_server.py:523-529 BoxMesh (0,-R,-R)-(Lx,R,R) with int(ny*2R) x
    int(nz*2R) cells across, :645 weight sqrt(y^2+z^2) (degree 2), :537-572 per-cell core diffusivity, :582-604
    directional conditions (on this mesh the `T_side` predicate matches no boundary facet), else T_boundary on the
    whole box boundary (:626-628).  Sine / cosine initial conditions are projected on the shifted mesh (:672-687)."""
    R = float(cylinder_radius)
    ny2, nz2 = int(ny * R * 2), int(nz * R * 2)
    if ny2 < 1 or nz2 < 1:
        raise L.PdeGpuError(L.EINVAL, f"BoxMesh fallback needs int(ny*2R) >= 1 and int(nz*2R) >= 1 (got {ny2}, {nz2})")
    directional = T_left is not None or T_right is not None or T_side is not None
    if directional:
        bc_mode = L.BC_DIRECTIONAL
        bc_value = (0.0 if T_left is None else T_left, 0.0 if T_right is None else T_right, 0.0)
        bc_present = (0 if T_left is None else 1, 0 if T_right is None else 1, 0)
    else:
        bc_mode, bc_value, bc_present = L.BC_UNIFORM, (T_boundary, 0.0, 0.0), (1, 0, 0)
    composite = core_radius is not None and core_diffusivity is not None
    u0 = None
    if not steady and initial_type in ("cosine", "sine"):                          # :672-687 on the shifted BoxMesh
        u0 = projected_trig_field((nx, ny2, nz2), (Lx, 2 * R, 2 * R), (0.0, -R, -R), initial_type, initial_amplitude,
                                  initial_wavenumber, opts.get("device"))
    values, times, st = run_weighted(3, (nx, ny2, nz2), (Lx, 2 * R, 2 * R), (0.0, -R, -R), L.WEIGHT_RCYL, kappa=diffusivity, dt=dt,
                                     num_steps=num_steps, steady=steady, source_type=source_type, source_value=source_value,
                                     T_initial=T_initial, bc_mode=bc_mode, bc_value=bc_value, bc_present=bc_present,
                                     core_radius=core_radius if composite else 0.0,
                                     core_ratio=core_diffusivity / diffusivity if composite else 1.0, u0=u0, **opts)
    coords = GridCoords(3, (nx, ny2, nz2), (Lx, 2 * R, 2 * R), origin=(0.0, -R, -R))
    meta = {"name": "temperature", "unit": "°C", "pde": "heat", "coordinate_system": "cylindrical", "Lx": Lx, "Ly": R * 2,
            "Lz": R * 2, "geometry_type": "cylinder", "source_type": source_type, "source_value": source_value, "steady": steady,
            "cylinder_radius": R}                                                   # :721-736
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
    meta.update({"backend": "pdegpu-hip-gfx950", "element_mode": "p1split", "cg_iterations": int(st["cg_iterations"]),
                 "solver_path": st.get("path", "general"), "solve_ms": float(st.get("solve_ms", 0.0)), "mesh": "BoxMesh fallback (no mshr)"})
    return TimeSeriesField(coords=coords, values=values, times=times, dim=3, meta=meta)
