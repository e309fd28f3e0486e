"""Checks on the device assembly the build leaves in pde-solver_amd/build (Makefile: -save-temps=obj).

A DPP lane shift reads 0 (bound_ctrl) or keeps the old value from a source lane that EXEC has switched off.  The kernels rely on every lane
being active when they shift: written as `edge_lane ? 0.0 : shift(U)` the compiler once moved the shift under exec = !edge_lane, and the
lane next to the edge silently lost its neighbour's contribution (elasticity x-faces, round 2).  The parity tests catch the wrong numbers
on a GPU; this one catches the pattern at build time: no DPP instruction may directly follow an instruction that narrows EXEC."""
import re
from pathlib import Path

import pytest

BUILD = Path(__file__).resolve().parent.parent / "pde-solver_amd" / "build"
NARROW = re.compile(r"^\s*(s_and_saveexec_b64|s_andn2_saveexec_b64|s_and_b64\s+exec|s_andn2_b64\s+exec|s_mov_b64\s+exec)")


def masked_dpp(path):
    hits = []
    window = []                                   # the last two instructions
    func = "?"
    for ln, line in enumerate(open(path, errors="replace"), 1):
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.endswith(":"):
            func = t[:-1]
            window = []
            continue
        if "_dpp" in t.split()[0] and any(NARROW.match(w) for w in window):
            hits.append((func, ln, t))
        window = (window + [t])[-2:]
    return hits


def test_no_dpp_under_a_narrowed_exec_mask():
    files = sorted(BUILD.glob("*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not files:
        pytest.skip("no device assembly in pde-solver_amd/build (built without the Makefile?)")
    seen_dpp = 0
    for f in files:
        text = f.read_text(errors="replace")
        seen_dpp += text.count("_dpp ")
        hits = masked_dpp(f)
        assert not hits, f"{f.name}: DPP right after an EXEC-narrowing instruction: {hits[:3]}"
    assert seen_dpp > 100                          # the marching kernels are there and do use DPP shifts


def test_the_scanner_sees_the_pattern(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_Zk:\n\ts_and_saveexec_b64 s[14:15], s[4:5]\n\tv_mov_b32_dpp v190, v126 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "\ts_or_b64 exec, exec, s[14:15]\n\tv_add_f64 v[0:1], v[0:1], v[2:3]\n\tv_mov_b32_dpp v1, v2 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
    hits = masked_dpp(bad)
    assert len(hits) == 1 and hits[0][0] == "_Zk" and "v190" in hits[0][2]
