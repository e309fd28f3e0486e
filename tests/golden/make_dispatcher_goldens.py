"""Generates tests/golden/dispatcher_kwargs.json by calling the reference's own pure functions
DispatcherAgent._build_*_args (dispatcher_agent.py:326-773, 979-1144) on a list of PDEParameters inputs.

Runs ONLY in the build container (reads /root/reference; python -B so nothing is written there).  The LLM /
MCP client packages the module imports at top level (langchain_openai, langchain_core, langchain_mcp_adapters:
dispatcher_agent.py:11-13) are not installed; they are never touched by the arg builders, so inert module
objects are registered for those names before the import.  Only the resulting inputs + kwargs (data) are
committed, never reference source."""
import contextlib
import io
import json
import sys
import types
from pathlib import Path

REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))


def _inert(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Dummy:
    def __init__(self, *a, **k):
        pass


_inert("langchain_openai", ChatOpenAI=_Dummy)
_inert("langchain_core")
_inert("langchain_core.messages", HumanMessage=_Dummy, SystemMessage=_Dummy, AIMessage=_Dummy)
_inert("langchain_core.output_parsers", JsonOutputParser=_Dummy, StrOutputParser=_Dummy)
_inert("langchain_mcp_adapters")
_inert("langchain_mcp_adapters.client", MultiServerMCPClient=_Dummy)

from pde_schema import PDEParameters  # noqa: E402
import dispatcher_agent  # noqa: E402

agent = dispatcher_agent.DispatcherAgent.__new__(dispatcher_agent.DispatcherAgent)

CASES = [
    # BASELINE.json configs 1, 2, 3, 5
    dict(name="cfg1_heat1d", params=dict(pde_type="heat", dim=1, domain_size={"length": 2.0}, nx=1000, bc_values={"t_left": 20, "t_right": 0})),
    dict(name="cfg2_heat2d", params=dict(pde_type="heat", dim=2, domain_size={"Lx": 1, "Ly": 1}, nx=1024, ny=1024, bc_values={"t_boundary": 0}, initial_value=20)),
    dict(name="cfg3_heat3d", params=dict(pde_type="heat", dim=3, domain_size={"Lx": 1, "Ly": 1, "Lz": 1}, nx=512, ny=512, nz=512, bc_values={"t_boundary": 0}, initial_value=20)),
    dict(name="cfg5_elast3d", params=dict(pde_type="elasticity", dim=3, domain_size={"Lx": 1, "Ly": 0.25, "Lz": 0.25}, nx=256, ny=64, nz=64, young_modulus=210e9, poisson_ratio=0.3, material_params={"gravity": True})),
    # defaults and quirks
    dict(name="heat1d_defaults", params=dict(pde_type="heat", dim=1)),
    dict(name="heat1d_keys", params=dict(pde_type="heat", dim=1, domain_size={"L": 3.0}, diffusivity=0.5, bc_values={"left": 5.0, "T_right_boundary": 1.0}, total_time=0.4)),
    dict(name="heat1d_nested", params=dict(pde_type="heat", dim=1, domain_size={"domain_size": 4.0}, dt=0.02, initial_type="sine", initial_amplitude=2.0, initial_wavenumber=3.0, source_type="constant", source_value=7.0)),
    dict(name="heat1d_zero_ic_steady", params=dict(pde_type="heat", dim=1, initial_type="zero", initial_value=9.0, steady=True, num_steps=7)),
    dict(name="heat2d_defaults", params=dict(pde_type="heat", dim=2)),
    dict(name="heat2d_quirk_lower_t_boundary", params=dict(pde_type="heat", dim=2, bc_values={"t_boundary": 50})),
    dict(name="heat2d_T_boundary", params=dict(pde_type="heat", dim=2, domain_size={"width": 2.0, "height": 0.5}, bc_values={"T_boundary": 50}, diffusivity=0.1, nx=40)),
    dict(name="heat2d_T_boundary_value", params=dict(pde_type="heat", dim=2, domain_size={"domain_size": 2.0}, bc_values={"T_boundary_value": 12.0}, initial_type="cosine", total_time=1.0)),
    dict(name="heat3d_defaults", params=dict(pde_type="heat", dim=3)),
    dict(name="heat3d_box_dims", params=dict(pde_type="heat", dim=3, geometry_type="cube", domain_size={"Lx": 1.0, "Ly": 0.2, "Lz": 0.2}, nx=20, ny=8, nz=8, bc_values={"T_boundary": 100.0}, initial_value=0.0, diffusivity=2.0)),
    dict(name="heat3d_dt_given", params=dict(pde_type="heat", dim=3, dt=0.005, num_steps=12, initial_type="zero")),
    dict(name="cyl1d_defaults", params=dict(pde_type="heat", dim=1, geometry_type="cylinder")),
    dict(name="cyl1d_full", params=dict(pde_type="heat", dim=1, geometry_type="pipe", geometry_params={"r_inner": 0.2, "cylinder_radius": 0.8}, nx=64, bc_values={"t_inner": 300.0, "T_outer": 25.0}, diffusivity=0.5, dt=0.002, num_steps=7, initial_value=25.0, source_type="constant", source_value=3.0)),
    dict(name="sph1d_boundary_only", params=dict(pde_type="heat", dim=1, geometry_type="ball", geometry_params={"radius": 2.0}, bc_values={"t_boundary": 5.0}, steady=True)),
    dict(name="cyl2d_defaults", params=dict(pde_type="heat", dim=2, coordinate_system="cylindrical")),
    dict(name="cyl2d_full", params=dict(pde_type="heat", dim=2, geometry_type="cylinder", domain_size={"length": 3.0}, geometry_params={"r_outer": 0.5}, nx=12, ny=20, bc_values={"T_boundary": 0.0, "t_boundary": 40.0}, initial_value=0.0)),
    dict(name="sph2d_defaults", params=dict(pde_type="heat", dim=2, geometry_type="sphere")),
    dict(name="sph3d_defaults", params=dict(pde_type="heat", dim=3, geometry_type="spherical")),
    dict(name="sph3d_full", params=dict(pde_type="heat", dim=3, domain_size={"r1": 0.1, "r2": 1.0}, geometry_params={"r_inner": 0.1, "sphere_radius": 1.5}, nx=8, ny=9, nz=10, bc_values={"t_boundary": 0.0}, initial_value=50.0, dt=0.005)),
    dict(name="cyl3d_defaults", params=dict(pde_type="heat", dim=3, geometry_type="cylinder")),
    dict(name="cyl3d_notation_core", params=dict(pde_type="heat", dim=3, geometry_type="tube", domain_size={"r1": 0.1, "r2": 0.4, "h": 1.5}, nx=12, ny=10, nz=10, bc_values={"t_left": 100.0, "T_right": 20.0, "t_side": 30.0}, core_diffusivity=25.0, initial_value=20.0)),
    dict(name="cyl3d_solid_core_default", params=dict(pde_type="heat", dim=3, geometry_type="cylinder", domain_size={"r2": 0.5, "h": 2.0}, core_diffusivity=10.0, bc_values={"T_boundary": 5.0})),
    dict(name="cyl3d_geom_params", params=dict(pde_type="heat", dim=3, geometry_type="pipe", domain_size={"Lx": 3.0}, geometry_params={"radius": 0.25}, core_radius=0.1, core_diffusivity=4.0)),
    dict(name="elast1d_defaults", params=dict(pde_type="elasticity", dim=1)),
    dict(name="elast1d_full", params=dict(pde_type="elasticity", dim=1, domain_size={"length": 2.0}, nx=80, young_modulus=70e9, material_params={"area": 0.01, "body_force": 1000.0, "quantity": "strain"})),
    dict(name="elast2d_defaults", params=dict(pde_type="elasticity", dim=2)),
    dict(name="elast2d_gravity", params=dict(pde_type="elasticity", dim=2, domain_size={"Lx": 2.0, "Ly": 0.5}, nx=40, ny=10, density=2700.0, material_params={"gravity": True, "plane_stress": False, "quantity": "weird"})),
    dict(name="elast3d_defaults", params=dict(pde_type="elasticity", dim=3)),
    dict(name="elast3d_explicit_force", params=dict(pde_type="elasticity", dim=3, domain_size={"width": 1.0, "height": 0.2, "depth": 0.1}, material_params={"body_fx": 5.0, "body_force_z": -3.0, "quantity": "strain"})),
]

BUILDERS = {("heat", 1): "_build_1d_args", ("heat", 2): "_build_2d_args", ("heat", 3): "_build_3d_args",
            ("elasticity", 1): "_build_elasticity_1d_args", ("elasticity", 2): "_build_elasticity_2d_args",
            ("elasticity", 3): "_build_elasticity_3d_args"}

CURVI = {("cylinder", 1): ("solve_heat_1D_cylindrical", "_build_1d_cylindrical_args"),
         ("sphere", 1): ("solve_heat_1D_spherical", "_build_1d_spherical_args"),
         ("cylinder", 2): ("solve_heat_2D_cylindrical", "_build_2d_cylindrical_args"),
         ("sphere", 2): ("solve_heat_2D_spherical", "_build_2d_spherical_args"),
         ("sphere", 3): ("solve_heat_3D_spherical", "_build_3d_spherical_args")}


def geometry_of(p):
    """the geometry normalisation block of DispatcherAgent.dispatch (dispatcher_agent.py:117-146) is inline code of an
    async method that needs the MCP client; the golden `tool` for the curvilinear cases is therefore derived here from
    the geometry the case was written for (each case names it unambiguously)."""
    g = (p.geometry_type or "").lower()
    if g in ("cylinder", "cylindrical", "column", "tube", "pipe", "tunnel"):
        return "cylinder"
    if g in ("sphere", "spherical", "ball"):
        return "sphere"
    if not g:
        if p.coordinate_system:
            return {"cylindrical": "cylinder", "spherical": "sphere"}.get(str(p.coordinate_system).lower(), "box")
        d = p.domain_size or {}
        if ("r1" in d or "r2" in d):
            return "cylinder" if "h" in d else "sphere"
    return "box"


out = []
for case in CASES:
    p = PDEParameters(**case["params"])
    geom = geometry_of(p) if p.pde_type == "heat" else "box"
    with contextlib.redirect_stdout(io.StringIO()):              # the 3-D builder prints debug lines
        if (geom, p.dim) in CURVI:
            tool, builder = CURVI[(geom, p.dim)]
            kw = getattr(agent, builder)(p)
        else:
            fn = getattr(agent, BUILDERS[(p.pde_type, p.dim)])
            kw = fn(p, geom if (p.dim == 3 and geom == "cylinder") else "box") if p.pde_type == "heat" else fn(p)
            tool = f"solve_heat_{p.dim}D" if p.pde_type == "heat" else f"solve_elasticity_{p.dim}D_static"
    out.append({"name": case["name"], "params": case["params"], "tool": tool, "kwargs": kw})

dst = Path(__file__).resolve().parent / "dispatcher_kwargs.json"
dst.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
print(f"wrote {len(out)} cases to {dst}")
