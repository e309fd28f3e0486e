"""Committed solution vectors (tests/golden/oracle_fields.npz, made by tests/golden/make_oracle_goldens.py from the CPU oracle —
not from the reference itself).  CPU: the oracle still reproduces them (drift guard).  GPU: the HIP path,
called through the raw solvers that mirror the reference's, reproduces them to the stated tolerance (1e-9 heat / 1e-8 elasticity
relative to the field maximum, solver rtol 1e-12)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import fem_p1 as o

G = Path(__file__).parent / "golden"
FIELDS = np.load(G / "oracle_fields.npz", allow_pickle=False)
CASES = json.loads((G / "oracle_fields_cases.json").read_text())


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _cut(name, v):
    v = np.asarray(v, dtype=float)
    return v[np.ix_([0, 1, 50, 200], np.arange(0, 1001, 50))] if name == "heat1d_config1" else v


def test_fixture_inventory():
    assert set(FIELDS.files) == set(CASES) and len(CASES) == 9
    assert abs(FIELDS["heat1d_config1"][3, 10] - 9.902778170) < 5e-9          # u(1.0, t = 2.0), SURVEY §8c(iii)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_committed_fields(name):
    fn, kw = CASES[name]
    assert _rel(_cut(name, getattr(o, fn)(**kw).values), FIELDS[name]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_reproduces_committed_fields(lib, gpu_ctx, name):
    from pdegpu import heat, elasticity, curvilinear
    fn, kw = CASES[name]
    raw = {"solve_heat_1d": heat._solve_heat_1d_raw, "solve_heat_2d": heat._solve_heat_2d_raw, "solve_heat_3d": heat._solve_heat_3d_raw,
           "solve_elasticity_2d": elasticity._solve_elasticity_2d_static, "solve_elasticity_3d": elasticity._solve_elasticity_3d_static,
           "solve_heat_1d_cylindrical": curvilinear._solve_heat_1d_cylindrical_raw}[fn]
    got = raw(**kw, rtol=1e-12)
    tol = 1e-8 if "elasticity" in name else 1e-9
    assert _rel(_cut(name, got.values), FIELDS[name]) < tol
