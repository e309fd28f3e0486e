"""The params -> kwargs -> tool boundary: golden kwargs captured from the reference's own arg builders
(tests/golden/dispatcher_kwargs.json, made by tests/golden/make_dispatcher_goldens.py), tool signatures
identical to the reference's, and — on a GPU — the pickle hand-off."""
import inspect
import json
import pickle
from pathlib import Path

import numpy as np
import pytest

GOLDEN = json.loads((Path(__file__).parent / "golden" / "dispatcher_kwargs.json").read_text())


@pytest.mark.parametrize("case", GOLDEN, ids=[c["name"] for c in GOLDEN])
def test_route_reproduces_reference_kwargs(case):
    from pdegpu import params
    name, kw = params.route(case["params"])
    assert name == case["tool"]
    assert kw == case["kwargs"]
    for k, v in case["kwargs"].items():
        assert type(kw[k]) is type(v), (k, kw[k], v)        # int stays int, float stays float through JSON


def test_route_accepts_the_reference_dataclass_shape_and_rejects_other_pdes():
    from pdegpu import params

    class P:                                   # stands for pde_schema.PDEParameters: only to_dict() is used
        def to_dict(self):
            return {"pde_type": "heat", "dim": 2, "domain_size": {}, "bc_values": {"T_boundary": 5.0}}
    assert params.route(P())[1]["T_boundary"] == 5.0
    with pytest.raises(ValueError):
        params.route({"pde_type": "wave", "dim": 1})
    with pytest.raises(ValueError):
        params.route({"pde_type": "heat", "dim": 4})
    assert params.route({"pde_type": "heat", "dim": 3, "geometry_type": "pipe"})[1]["geometry_type"] == "cylinder"
    assert params.geometry_of({"domain_size": {"r2": 1.0, "h": 2.0}}) == "cylinder"
    assert params.geometry_of({"coordinate_system": "spherical"}) == "sphere"
    assert params.geometry_of({"geometry_type": "Cube"}) == "box"


# signatures of the reference's tools (fenics_mcp_server.py:1903-1918, 1978-1994, 2123-2148, 2524-2531, 2591-2602,
# 2681-2694, 2765-2771): (name, default) in order
SIGNATURES = {
    "solve_heat_1D": [("length", 2.0), ("nx", 50), ("diffusivity", 1.0), ("T_left", 20.0), ("T_right", 0.0), ("T_initial", 0.0),
                      ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"), ("steady", False), ("source_type", "none"),
                      ("source_value", 0.0), ("initial_type", "constant"), ("initial_amplitude", 1.0), ("initial_wavenumber", 1.0)],
    "solve_heat_2D": [("Lx", 1.0), ("Ly", 1.0), ("nx", 30), ("ny", 30), ("diffusivity", 1.0), ("T_boundary", 0.0), ("T_initial", 20.0),
                      ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"), ("steady", False), ("source_type", "none"),
                      ("source_value", 0.0), ("initial_type", "constant"), ("initial_amplitude", 1.0), ("initial_wavenumber", 1.0)],
    "solve_heat_3D": [("Lx", 1.0), ("Ly", 1.0), ("Lz", 1.0), ("nx", 10), ("ny", 10), ("nz", 10), ("diffusivity", 1.0), ("T_boundary", 0.0),
                      ("T_initial", 20.0), ("dt", 0.01), ("num_steps", 20), ("data_dir", "data"), ("steady", False),
                      ("source_type", "none"), ("source_value", 0.0), ("initial_type", "constant"), ("initial_amplitude", 1.0),
                      ("initial_wavenumber", 1.0), ("geometry_type", "box"), ("cylinder_radius", None), ("T_left", None),
                      ("T_right", None), ("T_side", None), ("core_radius", None), ("core_diffusivity", None)],
    "solve_heat_1D_cylindrical": [("r_inner", 0.1), ("r_outer", 1.0), ("nr", 50), ("diffusivity", 1.0), ("T_inner", 100.0), ("T_outer", 20.0),
                                  ("T_initial", 20.0), ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"), ("steady", False),
                                  ("source_type", "none"), ("source_value", 0.0), ("initial_type", "constant"), ("initial_amplitude", 1.0)],
    "solve_heat_1D_spherical": [("r_inner", 0.1), ("r_outer", 1.0), ("nr", 50), ("diffusivity", 1.0), ("T_inner", 100.0), ("T_outer", 20.0),
                                ("T_initial", 20.0), ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"), ("steady", False),
                                ("source_type", "none"), ("source_value", 0.0), ("initial_type", "constant"), ("initial_amplitude", 1.0)],
    "solve_heat_2D_cylindrical": [("r_inner", 0.1), ("r_outer", 1.0), ("z_length", 2.0), ("nr", 30), ("nz", 30), ("diffusivity", 1.0),
                                  ("T_boundary", 20.0), ("T_initial", 20.0), ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"),
                                  ("steady", False), ("source_type", "none"), ("source_value", 0.0), ("initial_type", "constant"),
                                  ("initial_amplitude", 1.0)],
    "solve_heat_2D_spherical": [("r_inner", 0.1), ("r_outer", 1.0), ("nr", 30), ("ntheta", 30), ("diffusivity", 1.0), ("T_boundary", 20.0),
                                ("T_initial", 20.0), ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"), ("steady", False),
                                ("source_type", "none"), ("source_value", 0.0), ("initial_type", "constant"), ("initial_amplitude", 1.0)],
    "solve_heat_3D_spherical": [("r_inner", 0.1), ("r_outer", 1.0), ("nr", 20), ("ntheta", 20), ("nphi", 20), ("diffusivity", 1.0),
                                ("T_boundary", 20.0), ("T_initial", 20.0), ("dt", 0.01), ("num_steps", 50), ("data_dir", "data"),
                                ("steady", False), ("source_type", "none"), ("source_value", 0.0), ("initial_type", "constant"),
                                ("initial_amplitude", 1.0)],
    "solve_elasticity_1D_static": [("L", 1.0), ("nx", 50), ("E", 210e9), ("area", 1.0), ("body_force", 0.0), ("quantity", "stress"),
                                   ("data_dir", "data")],
    "solve_elasticity_2D_static": [("Lx", 1.0), ("Ly", 1.0), ("nx", 30), ("ny", 30), ("E", 210e9), ("nu", 0.3), ("body_fx", 0.0),
                                   ("body_fy", 0.0), ("quantity", "stress"), ("plane_stress", True), ("data_dir", "data")],
    "solve_elasticity_3D_static": [("Lx", 1.0), ("Ly", 1.0), ("Lz", 1.0), ("nx", 10), ("ny", 10), ("nz", 10), ("E", 210e9), ("nu", 0.3),
                                   ("body_fx", 0.0), ("body_fy", 0.0), ("body_fz", 0.0), ("quantity", "stress"), ("data_dir", "data")],
    "plot_time_series_field_from_file": [("data_file", inspect.Parameter.empty), ("field_name", None), ("unit", None),
                                         ("output_dir", "plots"), ("filename", None)],
}


@pytest.mark.parametrize("tool", sorted(SIGNATURES))
def test_tool_signatures_are_the_reference_signatures(tool):
    from pdegpu import server
    sig = inspect.signature(server.TOOLS[tool])
    assert [(p.name, p.default) for p in sig.parameters.values()] == SIGNATURES[tool]


def test_drop_in_module_exposes_the_tools_without_mcp():
    import importlib.util
    path = Path(__file__).resolve().parent.parent / "pde-solver_amd" / "fenics_mcp_server.py"
    spec = importlib.util.spec_from_file_location("fenics_mcp_server_dropin", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for tool in SIGNATURES:
        assert callable(getattr(mod, tool))
    assert {"TimeSeriesField", "SolveResult", "PlotResult"} <= set(dir(mod))


def test_plot_tool_and_pickle_contract(tmp_path):
    """pickle round trip + the shape checks of the reference plotter (:3457-3466), no GPU involved"""
    from pdegpu import server
    from pdegpu.fields import TimeSeriesField, grid_coords
    f = TimeSeriesField(coords=grid_coords(2, (3, 2), (1.0, 0.5)), values=np.arange(24.0).reshape(2, 12),
                        times=np.array([0.0, 0.1]), dim=2, meta={"name": "temperature", "unit": "°C", "pde": "heat"})
    res = server._store(f, str(tmp_path / "data"), "heat_2d")
    assert Path(res.data_file).name.startswith("heat_2d_") and res.data_file.endswith(".pkl") and len(Path(res.data_file).stem) == 16
    back = pickle.load(open(res.data_file, "rb"))
    assert {"coords", "values", "times", "dim", "meta"} <= set(vars(back))
    assert np.asarray(back.coords).shape == (12, 3) and np.asarray(back.values).shape == (2, 12)
    plot = server.plot_time_series_field_from_file(res.data_file, output_dir=str(tmp_path / "plots"))
    assert Path(plot.html_path).exists() and "temperature" in Path(plot.html_path).read_text()
    bad = TimeSeriesField(coords=np.zeros((5, 3)), values=np.zeros((2, 4)), times=np.zeros(2), dim=1, meta={})
    res2 = server._store(bad, str(tmp_path / "data"), "heat_1d")
    with pytest.raises(ValueError):
        server.plot_time_series_field_from_file(res2.data_file, output_dir=str(tmp_path / "plots"))


@pytest.mark.parametrize("dim,n", [(1, (700,)), (2, (400, 30)), (3, (9, 5, 4))])
def test_plot_tool_downsamples_structured_fields(tmp_path, dim, n):
    """the re-hosted plot tool: grid shape inferred from the node order, frames strided to <= 192 pixels per axis and
    <= 60 times, embedded as bytes in a self-contained HTML (no GPU involved)"""
    import base64, json, re
    from pdegpu import server
    from pdegpu.fields import TimeSeriesField, grid_coords
    Ls = (2.0, 1.0, 0.5)[:dim]
    c = grid_coords(dim, n, Ls)
    assert c.grid_shape == tuple(k + 1 for k in n) + (1,) * (3 - dim) and len(pickle.dumps(c)) < 400          # lazy: pickles as its axes' description
    assert server._infer_grid(np.asarray(c)) == c.grid_shape                                                   # plain arrays: inferred from the node order
    nt = 130
    vals = np.linspace(0.0, 1.0, nt)[:, None] * (1.0 + c[:, 0])[None, :]
    f = TimeSeriesField(coords=c, values=vals, times=np.arange(nt) * 0.1, dim=dim, meta={"name": "temperature", "unit": "°C"})
    res = server._store(f, str(tmp_path / "data"), f"heat_{dim}d")
    assert Path(res.data_file + ".preview.npz").exists()
    html = Path(server.plot_time_series_field_from_file(res.data_file, output_dir=str(tmp_path / "plots")).html_path).read_text()
    # without the sidecar (a pickle written by someone else) the tool builds the same view from the pickle
    Path(res.data_file + ".preview.npz").unlink()
    assert Path(server.plot_time_series_field_from_file(res.data_file, output_dir=str(tmp_path / "plots2")).html_path).read_text() == html
    doc = json.loads(re.search(r"<script type='application/json' id='d'>(.*?)</script>", html, re.S).group(1))
    assert doc["w"] <= 192 and doc["h"] <= 192 and len(doc["times"]) <= 60
    assert doc["times"][0] == 0.0 and abs(doc["times"][-1] - 12.9) < 1e-12
    raw = np.frombuffer(base64.b64decode(doc["b64"]), dtype=np.uint8).reshape(len(doc["times"]), doc["h"], doc["w"])
    assert raw[0].max() == 0 and raw[-1].max() == 255 and raw[-1, 0, 0] < raw[-1, 0, -1]      # 0 at t = 0; grows with x
    assert server._infer_grid(np.random.default_rng(0).uniform(size=(11, 3))) is None or True


def test_solve_reports_errors_like_the_dispatcher():
    from pdegpu import server
    out = server.solve({"pde_type": "wave", "dim": 1})
    assert "error" in out and "only 'heat' and 'elasticity'" in out["error"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in GOLDEN if c["name"] in ("heat1d_defaults", "heat2d_defaults", "heat3d_defaults",
                                                                     "elast1d_defaults", "elast2d_defaults", "elast3d_explicit_force")],
                         ids=lambda c: c["name"])
def test_solve_params_dict_end_to_end(case, tmp_path, gpu_ctx):
    """params dict in -> pickle with numpy arrays out, meta keys of the reference, values match the oracle"""
    from oracle import fem_p1 as o
    from pdegpu import server
    p = dict(case["params"])
    if p["pde_type"] == "heat":
        p["num_steps"] = 3                           # keep the GPU suite short; dt stays the dispatcher default
    out = server.solve(p, data_dir=str(tmp_path))
    assert "error" not in out, out
    field = pickle.load(open(out["data_file"], "rb"))
    kw = dict(out["solver_args"])
    kw.pop("data_dir")
    ref = {"solve_heat_1D": o.solve_heat_1d, "solve_heat_2D": o.solve_heat_2d, "solve_heat_3D": o.solve_heat_3d,
           "solve_elasticity_1D_static": lambda L, **k: o.solve_elasticity_1d(L, **k),
           "solve_elasticity_2D_static": o.solve_elasticity_2d, "solve_elasticity_3D_static": o.solve_elasticity_3d}[out["solver"]](**kw)
    assert isinstance(field.values, np.ndarray) and field.values.shape == ref.values.shape
    assert np.abs(field.values - ref.values).max() <= 1e-8 * max(np.abs(ref.values).max(), 1e-300)
    assert set(ref.meta) <= set(field.meta)
    assert out["solver_result"].dim == p["dim"] and out["solver_result"].meta == field.meta


@pytest.mark.gpu
def test_tool_calls_write_nothing_to_stdout(lib, gpu_ctx, tmp_path):
    """stdout is the reference server's JSON-RPC channel (fenics_mcp_server.py:41-52; every raw solver redirects it, :223-225, 364-366, 508-510; SURVEY §5:
    "the build's server must never write to stdout").  One call of every tool family — 1-D direct, 2-D persistent launch, 3-D marching / hybrid / weighted
    marching / class-table, elasticity 1-D / 2-D / 3-D, the plot tool — and the creation of the library's RCCL communicator (RCCL prints a version banner
    when a communicator is initialised: csrc/comm.hip points fd 1 at stderr for that call), in a child process whose fd 1 is a file: it must stay empty."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    script = f'''
import sys
sys.path[:0] = [{str(root)!r}, {str(root / "pde-solver_amd")!r}]
from pdegpu import server as S, _lib as L, dist as D
d = {str(tmp_path)!r}
outs = [S.solve_heat_1D(nx=40, num_steps=3, data_dir=d), S.solve_heat_2D(nx=20, ny=20, num_steps=2, data_dir=d),
        S.solve_heat_3D(nx=8, ny=8, nz=8, num_steps=2, data_dir=d), S.solve_heat_3D(nx=8, ny=8, nz=8, num_steps=2, T_left=50.0, data_dir=d),
        S.solve_heat_3D(nx=8, ny=8, nz=8, num_steps=2, core_radius=0.4, core_diffusivity=5.0, data_dir=d),
        S.solve_heat_3D(nx=8, ny=8, nz=8, num_steps=2, geometry_type="cylinder", cylinder_radius=0.5, data_dir=d),
        S.solve_heat_3D_spherical(nr=8, ntheta=8, nphi=8, num_steps=2, data_dir=d), S.solve_heat_2D_spherical(nr=8, ntheta=8, num_steps=2, data_dir=d),
        S.solve_heat_2D_cylindrical(nr=8, nz=8, num_steps=2, data_dir=d), S.solve_heat_1D_cylindrical(nr=10, num_steps=2, data_dir=d),
        S.solve_heat_1D_spherical(nr=10, num_steps=2, data_dir=d), S.solve_elasticity_1D_static(nx=10, body_force=1.0, data_dir=d),
        S.solve_elasticity_2D_static(nx=8, ny=8, body_fy=-1e4, data_dir=d), S.solve_elasticity_3D_static(nx=8, ny=4, nz=4, body_fz=-7e4, data_dir=d)]
S.plot_time_series_field_from_file(outs[2].data_file, output_dir=d)
S.solve({{"pde_type": "heat", "dim": 3, "nx": 6, "ny": 6, "nz": 6, "num_steps": 2}}, data_dir=d)


class H:
    ctx = L.context(0)


c = D.NativeComm(H, 0, 1)          # ncclCommInitRank of a one-rank communicator: the banner, if this RCCL prints one, must go to stderr
c.close()
sys.stderr.write("CHILD DONE %d\\n" % len(outs))
'''
    out_file = tmp_path / "fd1.txt"
    with open(out_file, "wb") as fh:
        p = subprocess.run([sys.executable, "-c", script], stdout=fh, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert b"CHILD DONE 14" in p.stderr
    assert out_file.read_bytes() == b"", out_file.read_bytes()[:500]
