"""pde_schema.PDEParameters (dict or dataclass) -> tool name + tool kwargs.

Restates, as a specification, what the reference's dispatcher does between the parser and the MCP tools
(dispatcher_agent.py:115-197 routing; :336-446, 458-535, 537-773 heat arg builders; :979-1144 elasticity arg
builders) so that a bare params dict can drive the solver without the LLM stack.  Quirks are reproduced
literally and marked; tests/golden/dispatcher_kwargs.json (captured from the reference's own functions) pins it.

Two idioms of the reference matter:
  * `a.get(k1) or a.get(k2) or default`  -> first *truthy* value (0 / 0.0 / "" fall through to the default);
  * `if k in d: v = d[k]`                -> first *present* key (0.0 is kept)."""
from __future__ import annotations

from typing import Any, Dict, Mapping, Tuple

_BOX = {"cube", "cubic", "box", "rectangular"}
_CYL = {"cylinder", "cylindrical", "column", "tube", "pipe", "tunnel"}
_SPH = {"sphere", "spherical", "ball"}


def _as_dict(params) -> Dict[str, Any]:
    if isinstance(params, Mapping):
        return dict(params)
    if hasattr(params, "to_dict"):
        return params.to_dict()                                   # pde_schema.PDEParameters.to_dict (pde_schema.py:74-84)
    return dict(vars(params))


def _truthy(d: Mapping, keys, default):
    for k in keys:
        v = d.get(k)
        if v:
            return v
    return default


def _present(d: Mapping, keys, default):
    for k in keys:
        if k in d:
            return d[k]
    return default


def _val(p, key, default):
    """`params.x if params.x is not None else default`"""
    v = p.get(key)
    return default if v is None else v


def geometry_of(p: Mapping) -> str:
    """dispatcher_agent.py:117-146"""
    g = (p.get("geometry_type") or "").lower()
    if g in _BOX:
        return "box"
    if g in _CYL:
        return "cylinder"
    if g in _SPH:
        return "sphere"
    if not g:
        cs = p.get("coordinate_system")
        if cs:
            cs = str(cs).lower()
            return "cylinder" if cs == "cylindrical" else ("sphere" if cs == "spherical" else "box")
        dom = p.get("domain_size") or {}
        has_r, has_h = ("r1" in dom or "r2" in dom), "h" in dom
        if has_r and has_h:
            return "cylinder"
        if has_r:
            return "sphere"
        return "box"
    return g                                                       # unknown names pass through (:117-126) and end in the box tools


def _time_grid(p, t_char_len, kappa):
    """dt / num_steps defaults (:393-404, 484-498, 631-645): t_char = L^2/(2 kappa); dt = min(t_char/200, 0.01);
    num_steps = max(int(total_time/dt), 100) if total_time else max(int(t_char/dt), 200)."""
    dt, num_steps = p.get("dt"), p.get("num_steps")
    if dt is None or num_steps is None:
        t_char = t_char_len ** 2 / (2 * kappa) if kappa > 0 else 0.1
        if dt is None:
            dt = min(t_char / 200, 0.01)
        if num_steps is None:
            if p.get("total_time"):
                num_steps = max(int(p["total_time"] / dt), 100) if dt > 0 else 200
            else:
                num_steps = max(int(t_char / dt), 200) if dt > 0 else 200
    return dt, num_steps


def _common_heat_tail(p, default_T_initial):
    """source / steady / initial-condition fields shared by the three heat builders (:406-430, 504-520, 651-667)"""
    initial_type = p.get("initial_type") or "constant"
    if initial_type == "constant":
        T_initial = _val(p, "initial_value", default_T_initial)
    elif initial_type == "zero":
        T_initial = 0.0
    else:
        T_initial = _val(p, "initial_value", 0.0)
    return {
        "T_initial": T_initial,
        "data_dir": "data",
        "steady": _val(p, "steady", False),
        "source_type": p.get("source_type") or "none",
        "source_value": _val(p, "source_value", 0.0),
        "initial_type": initial_type,
        "initial_amplitude": _val(p, "initial_amplitude", 1.0),
        "initial_wavenumber": _val(p, "initial_wavenumber", 1.0),
    }


def heat_1d_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:336-446"""
    dom = p.get("domain_size") or {}
    if "length" in dom or "L" in dom or "l" in dom:
        length = _truthy(dom, ("length", "Length", "L", "l"), 2.0)
    elif "domain_size" in dom and isinstance(dom["domain_size"], (int, float)):
        length = float(dom["domain_size"])
    else:
        length = 2.0
    kappa = p.get("diffusivity") or 1.0
    bc = p.get("bc_values") or {}
    out = {
        "length": length,
        "nx": p.get("nx") or 50,
        "diffusivity": kappa,
        "T_left": _present(bc, ("T_left", "t_left", "T_Left", "left", "T_left_boundary", "t_left_boundary"), 20.0),
        "T_right": _present(bc, ("T_right", "t_right", "T_Right", "right", "T_right_boundary", "t_right_boundary"), 0.0),
    }
    out["dt"], out["num_steps"] = _time_grid(p, length, kappa)
    out.update(_common_heat_tail(p, 0.0))
    return out


def _box_lengths(dom, names):
    """`domain.get("Lx") or domain.get("lx") or ... or 1.0` per axis; a single nested number sets every axis"""
    if "domain_size" in dom and isinstance(dom["domain_size"], (int, float)):
        return [float(dom["domain_size"])] * len(names)
    return [_truthy(dom, keys, 1.0) for keys in names]


_LX = ("Lx", "lx", "width", "Width", "W")
_LY = ("Ly", "ly", "height", "Height", "H")
_LZ = ("Lz", "lz", "depth", "Depth", "D")


def heat_2d_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:458-535.  QUIRK (:477): only the keys "T_boundary" / "T_boundary_value" are read, with
    `or`, so the lower-case "t_boundary" the parser is told to emit — and an explicit 0 — both give 0.0."""
    dom = p.get("domain_size") or {}
    Lx, Ly = _box_lengths(dom, (_LX, _LY))
    kappa = p.get("diffusivity") or 1.0
    bc = p.get("bc_values") or {}
    out = {"Lx": Lx, "Ly": Ly, "nx": p.get("nx") or 30, "ny": p.get("ny") or 30, "diffusivity": kappa,
           "T_boundary": _truthy(bc, ("T_boundary", "T_boundary_value"), 0.0)}
    dt, num_steps = _time_grid(p, (Lx + Ly) / 2, kappa)
    out["dt"], out["num_steps"] = dt or 0.01, num_steps or 200
    out.update(_common_heat_tail(p, 20.0))
    return out


def heat_3d_kwargs(p: Mapping, geometry: str = "box") -> Dict[str, Any]:
    """dispatcher_agent.py:537-773: box branch and the cylinder branch (:574-596 lengths, :692-760 radius, directional
    boundary values, composite core)."""
    dom = p.get("domain_size") or {}
    gp = p.get("geometry_params") or {}
    cyl_notation = ("r1" in dom or "r2" in dom) and "h" in dom
    if "domain_size" in dom and isinstance(dom["domain_size"], (int, float)):
        Lx = Ly = Lz = float(dom["domain_size"])
    elif geometry == "cylinder":
        if cyl_notation:
            if dom.get("r2") is None:
                raise ValueError("For cylindrical geometry, domain_size must contain 'r2' (outer radius)")
            if dom.get("h") is None:
                raise ValueError("For cylindrical geometry, domain_size must contain 'h' (height)")
            Lx, Ly, Lz = float(dom["h"]), float(dom["r2"] * 2), float(dom["r2"] * 2)
        else:
            Lx = dom.get("Lx") or 2.0
            r2 = _truthy(gp, ("cylinder_radius", "r2", "r_outer"), 0.5)
            Ly = Lz = float(r2 * 2)
    else:
        Lx, Ly, Lz = _box_lengths(dom, (_LX, _LY, _LZ))
    kappa = p.get("diffusivity") or 1.0
    bc = p.get("bc_values") or {}
    out = {"Lx": Lx, "Ly": Ly, "Lz": Lz, "nx": p.get("nx") or 10, "ny": p.get("ny") or 10, "nz": p.get("nz") or 10,
           "diffusivity": kappa, "T_boundary": _truthy(bc, ("T_boundary", "T_boundary_value"), 0.0)}
    dt, num_steps = _time_grid(p, (Lx + Ly + Lz) / 3, kappa)
    out["dt"], out["num_steps"] = dt or 0.01, num_steps or 200
    out.update(_common_heat_tail(p, 20.0))
    if geometry != "cylinder":
        out["geometry_type"] = "box"      # T_left/T_right/T_side are forwarded for cylinders only (:713-720)
        return out
    out["geometry_type"] = "cylinder"
    r1 = r2 = None
    if cyl_notation:
        r1, r2 = dom.get("r1", 0.0), dom.get("r2")
        out["Lx"] = float(dom["h"])
        out["cylinder_radius"] = float(r2)
    else:
        out["cylinder_radius"] = r2 = _truthy(gp, ("cylinder_radius", "r_outer", "r2", "radius"), 0.5)
    for key in ("left", "right", "side"):                          # :713-720: `bc.get("t_x") or bc.get("T_x")`
        if f"t_{key}" in bc or f"T_{key}" in bc:
            out[f"T_{key}"] = bc.get(f"t_{key}") or bc.get(f"T_{key}")
    if p.get("core_diffusivity") is not None:                      # :722-754
        if p.get("core_radius") is not None:
            out["core_radius"] = float(p["core_radius"])
        elif cyl_notation and r1 is not None:
            out["core_radius"] = float(r1) if r1 > 0.0 else (float(r2) * 0.3 if r2 is not None else 0.1)
        else:
            out["core_radius"] = 0.1
        out["core_diffusivity"] = float(p["core_diffusivity"])
    return out


def _curvilinear_common(p, nr_default):
    """fields shared by the five curvilinear builders (dispatcher_agent.py:775-977): plain `x or default` defaults,
    no characteristic-time logic"""
    return {
        "diffusivity": p.get("diffusivity") or 1.0,
        "T_initial": _val(p, "initial_value", 20.0),
        "dt": p.get("dt") or 0.01,
        "num_steps": p.get("num_steps") or 50,
        "data_dir": "data",
        "steady": _val(p, "steady", False),
        "source_type": p.get("source_type") or "none",
        "source_value": _val(p, "source_value", 0.0),
        "initial_type": p.get("initial_type") or "constant",
        "initial_amplitude": _val(p, "initial_amplitude", 1.0),
    }


def _radii(p, radius_key):
    gp = p.get("geometry_params") or {}
    return gp.get("r_inner") or 0.0, _truthy(gp, ("r_outer", radius_key, "radius"), 1.0)


def _order(d, keys):
    return {k: d[k] for k in keys}


def heat_1d_radial_kwargs(p: Mapping, radius_key: str) -> Dict[str, Any]:
    """dispatcher_agent.py:775-812 (cylindrical, radius_key "cylinder_radius") and :814-851 (spherical, "sphere_radius")"""
    bc = p.get("bc_values") or {}
    r_inner, r_outer = _radii(p, radius_key)
    out = {"r_inner": r_inner, "r_outer": r_outer, "nr": p.get("nx") or 50,
           "T_inner": _truthy(bc, ("t_inner", "T_inner", "t_boundary", "T_boundary"), 100.0),
           "T_outer": _truthy(bc, ("t_outer", "T_outer", "t_boundary", "T_boundary"), 20.0)}
    out.update(_curvilinear_common(p, 50))
    return _order(out, ("r_inner", "r_outer", "nr", "diffusivity", "T_inner", "T_outer", "T_initial", "dt", "num_steps",
                        "data_dir", "steady", "source_type", "source_value", "initial_type", "initial_amplitude"))


def heat_2d_cylindrical_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:853-894"""
    dom = p.get("domain_size") or {}
    bc = p.get("bc_values") or {}
    r_inner, r_outer = _radii(p, "cylinder_radius")
    out = {"r_inner": r_inner, "r_outer": r_outer, "z_length": _truthy(dom, ("Lx", "length", "L"), 2.0),
           "nr": p.get("nx") or 30, "nz": p.get("ny") or 30, "T_boundary": _truthy(bc, ("t_boundary", "T_boundary"), 20.0)}
    out.update(_curvilinear_common(p, 30))
    return _order(out, ("r_inner", "r_outer", "z_length", "nr", "nz", "diffusivity", "T_boundary", "T_initial", "dt", "num_steps",
                        "data_dir", "steady", "source_type", "source_value", "initial_type", "initial_amplitude"))


def heat_2d_spherical_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:896-934"""
    bc = p.get("bc_values") or {}
    r_inner, r_outer = _radii(p, "sphere_radius")
    out = {"r_inner": r_inner, "r_outer": r_outer, "nr": p.get("nx") or 30, "ntheta": p.get("ny") or 30,
           "T_boundary": _truthy(bc, ("t_boundary", "T_boundary"), 20.0)}
    out.update(_curvilinear_common(p, 30))
    return _order(out, ("r_inner", "r_outer", "nr", "ntheta", "diffusivity", "T_boundary", "T_initial", "dt", "num_steps",
                        "data_dir", "steady", "source_type", "source_value", "initial_type", "initial_amplitude"))


def heat_3d_spherical_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:936-977"""
    bc = p.get("bc_values") or {}
    r_inner, r_outer = _radii(p, "sphere_radius")
    out = {"r_inner": r_inner, "r_outer": r_outer, "nr": p.get("nx") or 20, "ntheta": p.get("ny") or 20,
           "nphi": p.get("nz") or 20, "T_boundary": _truthy(bc, ("t_boundary", "T_boundary"), 20.0)}
    out.update(_curvilinear_common(p, 20))
    return _order(out, ("r_inner", "r_outer", "nr", "ntheta", "nphi", "diffusivity", "T_boundary", "T_initial", "dt",
                        "num_steps", "data_dir", "steady", "source_type", "source_value", "initial_type", "initial_amplitude"))


def _quantity(mp):
    q = mp.get("quantity") or "stress"
    return q if q in ("stress", "strain") else "stress"


def elasticity_1d_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:979-1014"""
    dom = p.get("domain_size") or {}
    mp = p.get("material_params") or {}
    if "length" in dom or "L" in dom or "l" in dom:
        L = _truthy(dom, ("length", "L", "l", "Length"), 1.0)
    elif "domain_size" in dom and isinstance(dom["domain_size"], (int, float)):
        L = float(dom["domain_size"])
    else:
        L = 1.0
    return {"L": L, "nx": p.get("nx") or 50, "E": _val(p, "young_modulus", 210e9),
            "area": _truthy(mp, ("area", "cross_sectional_area"), 1.0),
            "body_force": _truthy(mp, ("body_force", "body_force_x"), 0.0), "quantity": _quantity(mp), "data_dir": "data"}


def elasticity_2d_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:1016-1077; gravity acts in -y (:1040-1048)"""
    dom = p.get("domain_size") or {}
    mp = p.get("material_params") or {}
    Lx, Ly = _box_lengths(dom, (_LX, _LY))
    gravity = mp.get("gravity") or mp.get("apply_gravity") or False
    body_fx = _truthy(mp, ("body_fx", "body_force_x"), 0.0)
    if gravity and mp.get("body_fy") is None and mp.get("body_force_y") is None:
        body_fy = -9.81 * _val(p, "density", 7800.0)
    else:
        body_fy = _truthy(mp, ("body_fy", "body_force_y"), 0.0)
    plane_stress = mp.get("plane_stress")
    return {"Lx": Lx, "Ly": Ly, "nx": p.get("nx") or 30, "ny": p.get("ny") or 30, "E": _val(p, "young_modulus", 210e9),
            "nu": _val(p, "poisson_ratio", 0.3), "body_fx": body_fx, "body_fy": body_fy, "quantity": _quantity(mp),
            "plane_stress": True if plane_stress is None else plane_stress, "data_dir": "data"}


def elasticity_3d_kwargs(p: Mapping) -> Dict[str, Any]:
    """dispatcher_agent.py:1079-1144; gravity acts in -z (:1113-1119)"""
    dom = p.get("domain_size") or {}
    mp = p.get("material_params") or {}
    if "Lx" in dom or "lx" in dom or "width" in dom:
        Lx, Ly, Lz = (_truthy(dom, keys, 1.0) for keys in (_LX, _LY, _LZ))
    elif "domain_size" in dom and isinstance(dom["domain_size"], (int, float)):
        Lx = Ly = Lz = float(dom["domain_size"])
    else:
        Lx = Ly = Lz = 1.0
    fz_default = -9.81 * _val(p, "density", 7800.0) if (mp.get("gravity") or mp.get("apply_gravity")) else 0.0
    return {"Lx": Lx, "Ly": Ly, "Lz": Lz, "nx": p.get("nx") or 10, "ny": p.get("ny") or 10, "nz": p.get("nz") or 10,
            "E": _val(p, "young_modulus", 210e9), "nu": _val(p, "poisson_ratio", 0.3),
            "body_fx": _truthy(mp, ("body_fx", "body_force_x"), 0.0), "body_fy": _truthy(mp, ("body_fy", "body_force_y"), 0.0),
            "body_fz": _truthy(mp, ("body_fz", "body_force_z"), fz_default), "quantity": _quantity(mp), "data_dir": "data"}


def route(params) -> Tuple[str, Dict[str, Any]]:
    """(tool name, kwargs) for a PDEParameters dict / dataclass — dispatcher_agent.py:115-197."""
    p = _as_dict(params)
    pde, dim = p.get("pde_type", "heat"), p.get("dim", 1)
    if pde == "heat":
        geom = geometry_of(p)                                      # dispatcher_agent.py:148-175
        if dim == 1:
            if geom == "cylinder":
                return "solve_heat_1D_cylindrical", heat_1d_radial_kwargs(p, "cylinder_radius")
            if geom == "sphere":
                return "solve_heat_1D_spherical", heat_1d_radial_kwargs(p, "sphere_radius")
            return "solve_heat_1D", heat_1d_kwargs(p)
        if dim == 2:
            if geom == "cylinder":
                return "solve_heat_2D_cylindrical", heat_2d_cylindrical_kwargs(p)
            if geom == "sphere":
                return "solve_heat_2D_spherical", heat_2d_spherical_kwargs(p)
            return "solve_heat_2D", heat_2d_kwargs(p)
        if dim == 3:
            if geom == "sphere":
                return "solve_heat_3D_spherical", heat_3d_spherical_kwargs(p)
            return "solve_heat_3D", heat_3d_kwargs(p, geom if geom == "cylinder" else "box")
        raise ValueError(f"Unsupported dimension: {dim}")
    if pde == "elasticity":
        if dim == 1:
            return "solve_elasticity_1D_static", elasticity_1d_kwargs(p)
        if dim == 2:
            return "solve_elasticity_2D_static", elasticity_2d_kwargs(p)
        if dim == 3:
            return "solve_elasticity_3D_static", elasticity_3d_kwargs(p)
        raise ValueError(f"Unsupported dimension: {dim}")
    raise ValueError(f"Currently only 'heat' and 'elasticity' PDE types are supported, got: {pde}")
