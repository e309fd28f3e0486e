"""Generates tests/golden/oracle_fields.npz: small solution fields of the CPU oracle (oracle/fem_p1.py, the restatement of the
reference algorithm: P1 assembly on the simplex split + sparse LU) for fixed tool arguments.  They are NOT outputs of the
reference itself (FEniCS 2019.1.0 is not installable offline) — they freeze the oracle so that (a) the CPU suite
notices any drift of the restatement and (b) the GPU suite can compare the HIP path with committed vectors, not only with an
oracle evaluated in the same process.  Run from the repo root: python tests/golden/make_oracle_goldens.py"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import fem_p1 as o  # noqa: E402

CASES = {
    # name: (oracle function, kwargs)
    "heat1d_config1": ("solve_heat_1d", dict(length=2.0, nx=1000, diffusivity=1.0, T_left=20.0, T_right=0.0, T_initial=0.0, dt=0.01, num_steps=200)),
    "heat1d_source_sine": ("solve_heat_1d", dict(length=1.5, nx=60, diffusivity=0.3, T_left=1.0, T_right=-2.0, T_initial=0.0, dt=0.02, num_steps=4,
                                               source_type="constant", source_value=2.5, initial_type="sine", initial_amplitude=3.0, initial_wavenumber=2.0)),
    "heat2d_transient": ("solve_heat_2d", dict(Lx=1.0, Ly=0.7, nx=18, ny=11, diffusivity=0.8, T_boundary=3.0, T_initial=20.0, dt=0.004, num_steps=3)),
    "heat2d_steady_source": ("solve_heat_2d", dict(Lx=1.0, Ly=1.0, nx=16, ny=16, diffusivity=1.3, T_boundary=0.5, T_initial=0.0, dt=0.01, num_steps=1,
                                                   steady=True, source_type="constant", source_value=4.0)),
    "heat3d_uniform": ("solve_heat_3d", dict(Lx=1.0, Ly=0.8, Lz=0.6, nx=12, ny=9, nz=7, diffusivity=0.9, T_boundary=1.5, T_initial=20.0, dt=0.003, num_steps=2)),
    "heat3d_natural_faces": ("solve_heat_3d", dict(Lx=1.0, Ly=0.5, Lz=0.4, nx=9, ny=6, nz=5, diffusivity=1.0, T_boundary=0.0, T_initial=2.0, dt=0.01, num_steps=3,
                                                   source_type="constant", source_value=1.5, T_left=10.0, T_right=None, T_side=3.0)),
    "elasticity2d_plane_stress": ("solve_elasticity_2d", dict(Lx=1.0, Ly=0.3, nx=14, ny=5, E=210e9, nu=0.3, body_fx=0.0, body_fy=-5.0e4, quantity="stress", plane_stress=True)),
    "elasticity3d_gravity": ("solve_elasticity_3d", dict(Lx=1.0, Ly=0.25, Lz=0.25, nx=12, ny=3, nz=3, E=210e9, nu=0.3, body_fx=0.0, body_fy=0.0, body_fz=-76518.0, quantity="stress")),
    "heat1d_cylindrical": ("solve_heat_1d_cylindrical", dict(r_inner=0.1, r_outer=1.0, nr=40, diffusivity=1.0, T_inner=100.0, T_outer=20.0, T_initial=20.0, dt=0.01, num_steps=3)),
}


def main():
    out = {}
    for name, (fn, kw) in CASES.items():
        f = getattr(o, fn)(**kw)
        v = np.asarray(f.values, dtype=np.float64)
        if name == "heat1d_config1":                    # 201 x 1001 is 1.6 MB: keep the times / nodes of the known answers
            v = v[np.ix_([0, 1, 50, 200], np.arange(0, 1001, 50))]
        out[name] = v
    np.savez_compressed(ROOT / "tests" / "golden" / "oracle_fields.npz", **out)
    (ROOT / "tests" / "golden" / "oracle_fields_cases.json").write_text(json.dumps(CASES, indent=1))
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
