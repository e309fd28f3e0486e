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
from .fields import GridCoords, PlotResult, SolveResult, TimeSeriesField

PREVIEW_MAX_FRAMES, PREVIEW_MAX_PIXELS = 60, 192


def _grid_shape_of(coords):
    """(NX, NY, NZ) of the structured node order behind `coords`: known for GridCoords, inferred from the repetition pattern otherwise"""
    if isinstance(coords, GridCoords):
        return coords.grid_shape
    return _infer_grid(np.asarray(coords, dtype=float))


def _validate(field):
    """the shape checks of the reference's plotter (:3457-3466); returns (values, times) as arrays"""
    coords = field.coords
    values = np.asarray(field.values, dtype=float)
    times = np.asarray(field.times, dtype=float)
    cshape = tuple(coords.shape) if hasattr(coords, "shape") else np.asarray(coords, dtype=float).shape
    if len(cshape) != 2 or cshape[1] != 3:
        raise ValueError(f"coords must be [N,3], got {cshape}")
    if values.ndim != 2 or values.shape[1] != cshape[0]:
        raise ValueError(f"values must be [Nt,N] with N={cshape[0]}, got {values.shape}")
    if times.shape != (values.shape[0],):
        raise ValueError(f"times must be [Nt] with Nt={values.shape[0]}, got {times.shape}")
    return values, times


def _preview(field: TimeSeriesField):
    """What the plot tool shows, cut from the field while it is in memory: <= 60 stored times, the mid-z plane (the only plane in 1-D / 2-D)
    strided to <= 192 x 192, and min / max / mean of every shown state.  A few hundred KB next to a pickle of up to tens of GB."""
    values, times = _validate(field)
    shape = _grid_shape_of(field.coords)
    tsel = np.unique(np.linspace(0, len(times) - 1, min(len(times), PREVIEW_MAX_FRAMES)).round().astype(int))
    stats = np.array([[values[k].min(), values[k].max(), values[k].mean()] for k in tsel])
    import json
    out = {"tsel": tsel, "times": times[tsel], "stats": stats, "n_times": np.array(len(times)), "n_nodes": np.array(values.shape[1]),
           "meta_json": np.array(json.dumps(field.meta, default=str)), "dim": np.array(int(field.dim)),
           "vmin": np.array(float(values.min()) if values.size <= 1 << 27 else float(stats[:, 0].min())),
           "vmax": np.array(float(values.max()) if values.size <= 1 << 27 else float(stats[:, 1].max()))}
    if shape is not None and shape[0] * shape[1] * shape[2] == values.shape[1]:
        nx, ny, nz = shape
        sx, sy = max(1, -(-nx // PREVIEW_MAX_PIXELS)), max(1, -(-ny // PREVIEW_MAX_PIXELS))
        out["frames"] = np.stack([values[k].reshape(nz, ny, nx)[nz // 2, ::sy, ::sx] for k in tsel])
        out["shape"] = np.array(shape)
    return out


def _store(field: TimeSeriesField, data_dir: str, prefix: str) -> SolveResult:
    data_path = Path(data_dir)
    data_path.mkdir(parents=True, exist_ok=True)
    filepath = data_path / f"{prefix}_{uuid.uuid4().hex[:8]}.pkl"
    with open(filepath, "wb") as f:
        pickle.dump(field, f, protocol=pickle.HIGHEST_PROTOCOL)
    # the plot tool's view as a sidecar, so that plotting a 513^3 field does not un-pickle gigabytes to show 192 x 192 pixels of them
    # (the pickle stays the complete hand-off of the reference, :1962-1968; the sidecar is optional and rebuilt from it when missing)
    try:
        pv = _preview(field)
        if pv is not None:
            np.savez(str(filepath) + ".preview.npz", **pv)
    except Exception:  # noqa: BLE001 — a preview must never fail a solve
        pass
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
    """fenics_mcp_server.py:2122-2213: the box, its composite core (:537-572) and the BoxMesh fallback the reference runs for
    geometry_type="cylinder" (:523-529); only a true mshr cylinder mesh is not offered"""
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


def _infer_grid(coords: np.ndarray):
    """(NX, NY, NZ) of a lexicographic structured grid (x fastest), or None.  Works on curvilinear outputs too because
    only the repetition pattern of the node order is used, not the coordinate values of a Cartesian box."""
    n = coords.shape[0]
    x = coords[:, 0]
    back = np.nonzero(np.diff(x) < 0)[0]
    nx = int(back[0]) + 1 if back.size else n
    if n % nx:
        return None
    rows = n // nx
    if rows == 1:
        return nx, 1, 1
    y = coords[::nx, 1]
    back = np.nonzero(np.diff(y) < 0)[0]
    ny = int(back[0]) + 1 if back.size else rows
    if rows % ny:
        return None
    return nx, ny, rows // ny


_PLOT_JS = """
const D=JSON.parse(document.getElementById('d').textContent),cv=document.getElementById('c'),g=cv.getContext('2d'),
sl=document.getElementById('s'),lb=document.getElementById('l');
const raw=Uint8Array.from(atob(D.b64),c=>c.charCodeAt(0));
function col(v){const t=v/255;return [Math.round(255*Math.min(1,Math.max(0,1.5-Math.abs(4*t-3)))),
Math.round(255*Math.min(1,Math.max(0,1.5-Math.abs(4*t-2)))),Math.round(255*Math.min(1,Math.max(0,1.5-Math.abs(4*t-1))))];}
function draw(k){lb.textContent='t = '+D.times[k].toPrecision(5)+'   ['+D.vmin.toPrecision(5)+', '+D.vmax.toPrecision(5)+'] '+D.unit;
const o=k*D.w*D.h;if(D.h==1){g.fillStyle='#fff';g.fillRect(0,0,cv.width,cv.height);g.strokeStyle='#c00';g.beginPath();
for(let i=0;i<D.w;i++){const X=i/(D.w-1)*(cv.width-1),Y=(1-raw[o+i]/255)*(cv.height-1);i?g.lineTo(X,Y):g.moveTo(X,Y);}g.stroke();return;}
const im=g.createImageData(D.w,D.h);for(let j=0;j<D.h;j++)for(let i=0;i<D.w;i++){const c=col(raw[o+(D.h-1-j)*D.w+i]),q=4*(j*D.w+i);
im.data[q]=c[0];im.data[q+1]=c[1];im.data[q+2]=c[2];im.data[q+3]=255;}
createImageBitmap(im).then(b=>{g.imageSmoothingEnabled=false;g.drawImage(b,0,0,cv.width,cv.height);});}
sl.max=D.times.length-1;sl.oninput=()=>draw(+sl.value);draw(0);
"""


def plot_time_series_field_from_file(data_file: str, field_name: Optional[str] = None, unit: Optional[str] = None,
                                     output_dir: str = "plots", filename: Optional[str] = None) -> PlotResult:
    """Signature of fenics_mcp_server.py:2765-2771.  The dispatcher fails hard when this tool is missing
    (dispatcher_agent.py:248-257, 301-311), so the boundary offers it.  The reference's plotly animation (:2764-4551)
    scatters every node and re-triangulates with scipy `griddata`, which cannot digest the grids this build solves
    (> 10^5 points); this re-hosting writes a self-contained HTML (no external script): a time slider over the field
    down-sampled by striding the structured node order to <= 192 x 192 pixels and <= 60 frames — a line plot in 1-D, the
    image in 2-D, the mid-z plane in 3-D — plus the per-time min / max / mean table and the meta block.  It validates the
    same shapes the reference's plotter validates (:3457-3466)."""
    import base64
    import json
    side = Path(str(data_file) + ".preview.npz")
    pv, field = None, None
    if side.exists() and side.stat().st_mtime >= Path(data_file).stat().st_mtime:
        try:
            with np.load(side) as z:
                pv = {k: z[k] for k in z.files}
        except Exception:  # noqa: BLE001 — a damaged sidecar: fall back to the pickle
            pv = None
    if pv is None:
        with open(data_file, "rb") as f:
            field = pickle.load(f)
        pv = _preview(field)               # (validates the shapes the reference's plotter validates; raises ValueError)
    meta, dim = json.loads(str(pv["meta_json"])), int(pv["dim"])
    name = field_name or meta.get("name", "field")
    unit = unit or meta.get("unit", "")
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    stem = filename or (Path(data_file).stem + ".html")
    if not stem.endswith(".html"):
        stem += ".html"
    n_nodes, n_times = int(pv["n_nodes"]), int(pv["n_times"])
    rows = "".join(f"<tr><td>{t:.6g}</td><td>{st[0]:.6g}</td><td>{st[1]:.6g}</td><td>{st[2]:.6g}</td></tr>" for t, st in zip(pv["times"], pv["stats"]))
    meta_rows = "".join(f"<tr><td>{k}</td><td>{v}</td></tr>" for k, v in meta.items())
    view = ""
    if "frames" in pv:
        nx, ny, nz = (int(v) for v in pv["shape"])
        v = pv["frames"]
        vmin, vmax = float(pv["vmin"]), float(pv["vmax"])
        q = np.zeros_like(v, dtype=np.uint8) if vmax <= vmin else np.clip((v - vmin) / (vmax - vmin) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        doc = {"w": int(v.shape[2]), "h": int(v.shape[1]), "times": [float(t) for t in pv["times"]], "vmin": vmin, "vmax": vmax,
               "unit": unit, "b64": base64.b64encode(q.tobytes()).decode("ascii")}
        what = "line" if ny == 1 else ("field" if nz == 1 else f"plane k = {nz // 2} of {nz}")
        view = (f"<p>{what}, {nx} x {ny} x {nz} nodes shown at {doc['w']} x {doc['h']}</p><canvas id='c' width='576' height='"
                f"{576 if ny > 1 else 240}' style='border:1px solid #888'></canvas><br><input id='s' type='range' min='0' value='0' "
                f"style='width:576px'><div id='l'></div><script type='application/json' id='d'>{json.dumps(doc)}</script>"
                f"<script>{_PLOT_JS}</script>")
    html = (f"<html><head><meta charset='utf-8'><title>{name}</title></head><body><h2>{name} [{unit}] — {dim}-D, "
            f"{n_nodes} nodes, {n_times} stored times</h2>{view}<table border=1><tr><th>t</th><th>min</th><th>max</th>"
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
