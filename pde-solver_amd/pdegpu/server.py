"""Tool boundary: the six solver tools of the reference's MCP server with their names, keyword arguments and
defaults kept verbatim (fenics_mcp_server.py:1902-1918, 1977-1994, 2122-2148, 2523-2531, 2590-2602, 2680-2694),
the pickle hand-off (`data_dir/<prefix>_<uuid8>.pkl`, :1962-1968, 2029-2035, 2201-2207, 2576-2582, 2666-2672,
2749-2755) and `SolveResult(data_file, dim, meta)`.  Bulk arrays leave the process only through the pickle file,
as in the reference (:41-52); stdout is never written (it is the JSON-RPC channel, :222-225)."""
from __future__ import annotations

import pickle
import uuid
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np

from . import curvilinear as _cv
from . import elasticity as _el
from . import heat as _heat
from . import params as _params
from .fields import PlotResult, SolveResult, TimeSeriesField


def _store(field: TimeSeriesField, data_dir: str, prefix: str) -> SolveResult:
    data_path = Path(data_dir)
    data_path.mkdir(parents=True, exist_ok=True)
    filepath = data_path / f"{prefix}_{uuid.uuid4().hex[:8]}.pkl"
    with open(filepath, "wb") as f:
        pickle.dump(field, f, protocol=pickle.HIGHEST_PROTOCOL)
    return SolveResult(data_file=str(filepath), dim=field.dim, meta=field.meta)


def solve_heat_1D(length: float = 2.0, nx: int = 50, diffusivity: float = 1.0, T_left: float = 20.0,
                  T_right: float = 0.0, T_initial: float = 0.0, dt: float = 0.01, num_steps: int = 50,
                  data_dir: str = "data", steady: bool = False, source_type: str = "none", source_value: float = 0.0,
                  initial_type: str = "constant", initial_amplitude: float = 1.0,
                  initial_wavenumber: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:1902-1974"""
    field = _heat._solve_heat_1d_raw(length=length, nx=nx, diffusivity=diffusivity, T_left=T_left, T_right=T_right,
                                     T_initial=T_initial, dt=dt, num_steps=num_steps, steady=steady,
                                     source_type=source_type, source_value=source_value, initial_type=initial_type,
                                     initial_amplitude=initial_amplitude, initial_wavenumber=initial_wavenumber)
    return _store(field, data_dir, "heat_1d")


def solve_heat_2D(Lx: float = 1.0, Ly: float = 1.0, nx: int = 30, ny: int = 30, diffusivity: float = 1.0,
                  T_boundary: float = 0.0, T_initial: float = 20.0, dt: float = 0.01, num_steps: int = 50,
                  data_dir: str = "data", steady: bool = False, source_type: str = "none", source_value: float = 0.0,
                  initial_type: str = "constant", initial_amplitude: float = 1.0,
                  initial_wavenumber: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:1977-2041"""
    field = _heat._solve_heat_2d_raw(Lx=Lx, Ly=Ly, nx=nx, ny=ny, diffusivity=diffusivity, T_boundary=T_boundary,
                                     T_initial=T_initial, dt=dt, num_steps=num_steps, steady=steady,
                                     source_type=source_type, source_value=source_value, initial_type=initial_type,
                                     initial_amplitude=initial_amplitude, initial_wavenumber=initial_wavenumber)
    return _store(field, data_dir, "heat_2d")


def solve_heat_3D(Lx: float = 1.0, Ly: float = 1.0, Lz: float = 1.0, nx: int = 10, ny: int = 10, nz: int = 10,
                  diffusivity: float = 1.0, T_boundary: float = 0.0, T_initial: float = 20.0, dt: float = 0.01,
                  num_steps: int = 20, data_dir: str = "data", steady: bool = False, source_type: str = "none",
                  source_value: float = 0.0, initial_type: str = "constant", initial_amplitude: float = 1.0,
                  initial_wavenumber: float = 1.0, geometry_type: str = "box", cylinder_radius: Optional[float] = None,
                  T_left: Optional[float] = None, T_right: Optional[float] = None, T_side: Optional[float] = None,
                  core_radius: Optional[float] = None, core_diffusivity: Optional[float] = None) -> SolveResult:
    """fenics_mcp_server.py:2122-2213 (box geometry; cylinder / composite core raise NotImplementedError)"""
    field = _heat._solve_heat_3d_raw(Lx=Lx, Ly=Ly, Lz=Lz, nx=nx, ny=ny, nz=nz, diffusivity=diffusivity,
                                     T_boundary=T_boundary, T_initial=T_initial, dt=dt, num_steps=num_steps,
                                     steady=steady, source_type=source_type, source_value=source_value,
                                     initial_type=initial_type, initial_amplitude=initial_amplitude,
                                     initial_wavenumber=initial_wavenumber, geometry_type=geometry_type,
                                     cylinder_radius=cylinder_radius, T_left=T_left, T_right=T_right, T_side=T_side,
                                     core_radius=core_radius, core_diffusivity=core_diffusivity)
    return _store(field, data_dir, "heat_3d")


def solve_heat_3D_spherical(r_inner: float = 0.1, r_outer: float = 1.0, nr: int = 20, ntheta: int = 20, nphi: int = 20,
                            diffusivity: float = 1.0, T_boundary: float = 20.0, T_initial: float = 20.0, dt: float = 0.01,
                            num_steps: int = 50, data_dir: str = "data", steady: bool = False, source_type: str = "none",
                            source_value: float = 0.0, initial_type: str = "constant",
                            initial_amplitude: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:2044-2119"""
    field = _cv._solve_heat_3d_spherical_raw(r_inner=r_inner, r_outer=r_outer, nr=nr, ntheta=ntheta, nphi=nphi,
                                             diffusivity=diffusivity, T_boundary=T_boundary, T_initial=T_initial, dt=dt,
                                             num_steps=num_steps, steady=steady, source_type=source_type,
                                             source_value=source_value, initial_type=initial_type,
                                             initial_amplitude=initial_amplitude)
    return _store(field, data_dir, "heat_3d_spherical")


def _radial_tool(fn, prefix, r_inner, r_outer, nr, diffusivity, T_inner, T_outer, T_initial, dt, num_steps, data_dir, steady,
                 source_type, source_value, initial_type, initial_amplitude):
    field = fn(r_inner=r_inner, r_outer=r_outer, nr=nr, diffusivity=diffusivity, T_inner=T_inner, T_outer=T_outer,
               T_initial=T_initial, dt=dt, num_steps=num_steps, steady=steady, source_type=source_type,
               source_value=source_value, initial_type=initial_type, initial_amplitude=initial_amplitude)
    return _store(field, data_dir, prefix)


def solve_heat_1D_cylindrical(r_inner: float = 0.1, r_outer: float = 1.0, nr: int = 50, diffusivity: float = 1.0,
                              T_inner: float = 100.0, T_outer: float = 20.0, T_initial: float = 20.0, dt: float = 0.01,
                              num_steps: int = 50, data_dir: str = "data", steady: bool = False, source_type: str = "none",
                              source_value: float = 0.0, initial_type: str = "constant",
                              initial_amplitude: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:2220-2292"""
    return _radial_tool(_cv._solve_heat_1d_cylindrical_raw, "heat_1d_cylindrical", r_inner, r_outer, nr, diffusivity, T_inner,
                        T_outer, T_initial, dt, num_steps, data_dir, steady, source_type, source_value, initial_type,
                        initial_amplitude)


def solve_heat_1D_spherical(r_inner: float = 0.1, r_outer: float = 1.0, nr: int = 50, diffusivity: float = 1.0,
                            T_inner: float = 100.0, T_outer: float = 20.0, T_initial: float = 20.0, dt: float = 0.01,
                            num_steps: int = 50, data_dir: str = "data", steady: bool = False, source_type: str = "none",
                            source_value: float = 0.0, initial_type: str = "constant",
                            initial_amplitude: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:2295-2367"""
    return _radial_tool(_cv._solve_heat_1d_spherical_raw, "heat_1d_spherical", r_inner, r_outer, nr, diffusivity, T_inner,
                        T_outer, T_initial, dt, num_steps, data_dir, steady, source_type, source_value, initial_type,
                        initial_amplitude)


def solve_heat_2D_cylindrical(r_inner: float = 0.1, r_outer: float = 1.0, z_length: float = 2.0, nr: int = 30, nz: int = 30,
                              diffusivity: float = 1.0, T_boundary: float = 20.0, T_initial: float = 20.0, dt: float = 0.01,
                              num_steps: int = 50, data_dir: str = "data", steady: bool = False, source_type: str = "none",
                              source_value: float = 0.0, initial_type: str = "constant",
                              initial_amplitude: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:2370-2445"""
    field = _cv._solve_heat_2d_cylindrical_raw(r_inner=r_inner, r_outer=r_outer, z_length=z_length, nr=nr, nz=nz,
                                               diffusivity=diffusivity, T_boundary=T_boundary, T_initial=T_initial, dt=dt,
                                               num_steps=num_steps, steady=steady, source_type=source_type,
                                               source_value=source_value, initial_type=initial_type,
                                               initial_amplitude=initial_amplitude)
    return _store(field, data_dir, "heat_2d_cylindrical")


def solve_heat_2D_spherical(r_inner: float = 0.1, r_outer: float = 1.0, nr: int = 30, ntheta: int = 30, diffusivity: float = 1.0,
                            T_boundary: float = 20.0, T_initial: float = 20.0, dt: float = 0.01, num_steps: int = 50,
                            data_dir: str = "data", steady: bool = False, source_type: str = "none", source_value: float = 0.0,
                            initial_type: str = "constant", initial_amplitude: float = 1.0) -> SolveResult:
    """fenics_mcp_server.py:2448-2520"""
    field = _cv._solve_heat_2d_spherical_raw(r_inner=r_inner, r_outer=r_outer, nr=nr, ntheta=ntheta, diffusivity=diffusivity,
                                             T_boundary=T_boundary, T_initial=T_initial, dt=dt, num_steps=num_steps,
                                             steady=steady, source_type=source_type, source_value=source_value,
                                             initial_type=initial_type, initial_amplitude=initial_amplitude)
    return _store(field, data_dir, "heat_2d_spherical")


def solve_elasticity_1D_static(L: float = 1.0, nx: int = 50, E: float = 210e9, area: float = 1.0,
                               body_force: float = 0.0, quantity: str = "stress", data_dir: str = "data") -> SolveResult:
    """fenics_mcp_server.py:2523-2588"""
    field = _el._solve_elasticity_1d_static(L, nx=nx, E=E, area=area, body_force=body_force, quantity=quantity)
    return _store(field, data_dir, f"elasticity_1d_{quantity}")


def solve_elasticity_2D_static(Lx: float = 1.0, Ly: float = 1.0, nx: int = 30, ny: int = 30, E: float = 210e9,
                               nu: float = 0.3, body_fx: float = 0.0, body_fy: float = 0.0, quantity: str = "stress",
                               plane_stress: bool = True, data_dir: str = "data") -> SolveResult:
    """fenics_mcp_server.py:2590-2678"""
    field = _el._solve_elasticity_2d_static(Lx=Lx, Ly=Ly, nx=nx, ny=ny, E=E, nu=nu, body_fx=body_fx, body_fy=body_fy,
                                            quantity=quantity, plane_stress=plane_stress)
    return _store(field, data_dir, f"elasticity_2d_{quantity}")


def solve_elasticity_3D_static(Lx: float = 1.0, Ly: float = 1.0, Lz: float = 1.0, nx: int = 10, ny: int = 10,
                               nz: int = 10, E: float = 210e9, nu: float = 0.3, body_fx: float = 0.0,
                               body_fy: float = 0.0, body_fz: float = 0.0, quantity: str = "stress",
                               data_dir: str = "data") -> SolveResult:
    """fenics_mcp_server.py:2680-2761"""
    field = _el._solve_elasticity_3d_static(Lx=Lx, Ly=Ly, Lz=Lz, nx=nx, ny=ny, nz=nz, E=E, nu=nu, body_fx=body_fx,
                                            body_fy=body_fy, body_fz=body_fz, quantity=quantity)
    return _store(field, data_dir, f"elasticity_3d_{quantity}")


def plot_time_series_field_from_file(data_file: str, field_name: Optional[str] = None, unit: Optional[str] = None,
                                     output_dir: str = "plots", filename: Optional[str] = None) -> PlotResult:
    """Signature of fenics_mcp_server.py:2765-2771.  The dispatcher fails hard when this tool is missing
    (dispatcher_agent.py:248-257, 301-311), so the boundary offers it; the reference's plotly animation
    (:2764-4551) is a consumer of the solve path and is not rebuilt — this writes a small self-contained HTML
    summary (shape, value range per stored time, meta) and validates the same shapes the reference's plotter
    validates (:3457-3466)."""
    with open(data_file, "rb") as f:
        field = pickle.load(f)
    coords = np.asarray(field.coords, dtype=float)
    values = np.asarray(field.values, dtype=float)
    times = np.asarray(field.times, dtype=float)
    if coords.ndim != 2 or coords.shape[1] != 3:
        raise ValueError(f"coords must be [N,3], got {coords.shape}")
    if values.ndim != 2 or values.shape[1] != coords.shape[0]:
        raise ValueError(f"values must be [Nt,N] with N={coords.shape[0]}, got {values.shape}")
    if times.shape != (values.shape[0],):
        raise ValueError(f"times must be [Nt] with Nt={values.shape[0]}, got {times.shape}")
    name = field_name or field.meta.get("name", "field")
    unit = unit or field.meta.get("unit", "")
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    stem = filename or (Path(data_file).stem + ".html")
    if not stem.endswith(".html"):
        stem += ".html"
    rows = "".join(f"<tr><td>{t:.6g}</td><td>{v.min():.6g}</td><td>{v.max():.6g}</td><td>{v.mean():.6g}</td></tr>"
                   for t, v in zip(times[:: max(1, len(times) // 50)], values[:: max(1, len(times) // 50)]))
    meta_rows = "".join(f"<tr><td>{k}</td><td>{v}</td></tr>" for k, v in field.meta.items())
    html = (f"<html><head><meta charset='utf-8'><title>{name}</title></head><body><h2>{name} [{unit}] — {field.dim}-D, "
            f"{coords.shape[0]} nodes, {len(times)} stored times</h2><table border=1><tr><th>t</th><th>min</th><th>max</th>"
            f"<th>mean</th></tr>{rows}</table><h3>meta</h3><table border=1>{meta_rows}</table></body></html>")
    path = out / stem
    path.write_text(html, encoding="utf-8")
    return PlotResult(html_path=str(path))


TOOLS = {f.__name__: f for f in (solve_heat_1D, solve_heat_2D, solve_heat_3D, solve_heat_1D_cylindrical,
                                 solve_heat_1D_spherical, solve_heat_2D_cylindrical, solve_heat_2D_spherical,
                                 solve_heat_3D_spherical, solve_elasticity_1D_static, solve_elasticity_2D_static,
                                 solve_elasticity_3D_static, plot_time_series_field_from_file)}


def solve(pde_params, data_dir: Optional[str] = None) -> Dict[str, Any]:
    """pde_schema.PDEParameters (dataclass or its to_dict()) in -> {solver, solver_args, solver_result} out: the
    params -> kwargs -> tool path of dispatcher_agent.DispatcherAgent.dispatch (:97-231) without the MCP hop.
    Errors come back the way the dispatcher reports them: {"error": str, "solver_args": ...} (:320-324)."""
    name, kwargs = None, None
    try:
        name, kwargs = _params.route(pde_params)
        if data_dir is not None:
            kwargs["data_dir"] = data_dir
        res = TOOLS[name](**kwargs)
        return {"solver": name, "solver_args": kwargs, "solver_result": res, "data_file": res.data_file}
    except Exception as e:  # noqa: BLE001 — same catch-all as the reference boundary
        return {"error": str(e), "solver": name, "solver_args": kwargs}
