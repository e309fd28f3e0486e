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


def test_wave_uniform_weights_arrive_by_scalar_loads():
    """csrc/wmarch.hip, weights that do not depend on x: the 128-byte line of a grid row must reach the kernel through the scalar cache
    (`s_load_dwordx16` + x8 + x4 + x2 into SGPRs, constant address space) — as vector loads of a uniform address it would cost the 15 load
    instructions per node pair the class-table kernel paid; and the kernels with per-lane weights must not spill"""
    f = BUILD / "wmarch-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not f.exists():
        pytest.skip("no device assembly of wmarch.hip in pde-solver_amd/build")
    text = f.read_text(errors="replace")
    bodies = dict(re.findall(r"^(_ZN\S*k_wmarchILi(?:1|2)\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M))
    cm2 = {k: v for k, v in bodies.items() if "k_wmarchILi2E" in k}
    assert len(cm2) >= 6
    for name, body in cm2.items():
        assert "s_load_dwordx16" in body, name
    # launch bounds (256, 2) leave 256 VGPRs: none of the instantiations the library launches by default may use scratch
    for m in re.finditer(r"\.amdhsa_kernel (\S*k_wmarch\S*)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, meta = m.group(1), m.group(2)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1))
        edge_handoff = name.endswith("Lb1EEEv11StencilArgsNS_5WGeomE")
        per_lane_two_rows = "k_wmarchILi1ELi2E" in name
        if not per_lane_two_rows or edge_handoff:      # (per-lane weights, two rows, WITHOUT the LDS edge hand-off: 5 dwords of spill, not a default)
            assert scratch == 0, (name, scratch)


def test_the_persistent_2d_kernels_keep_their_state_in_registers():
    """csrc/persist2d.hip: the CG state of a band lives in statically indexed register arrays.  A helper that is handed such an array by pointer and is not
    inlined at every call site moves it to scratch memory (the Chebyshev form once: 704 bytes per lane, four times the time per stencil application) —
    no instantiation may have a private segment."""
    f = BUILD / "persist2d-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not f.exists():
        pytest.skip("no device assembly of persist2d.hip in pde-solver_amd/build")
    text = f.read_text(errors="replace")
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*k_cg2d_persistent\S*)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, meta = m.group(1), m.group(2)
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1)) == 0, name
        seen += 1
    assert seen >= 11      # classic / single-reduction forms of 1, 2, 4, 8 rows + the Chebyshev form of 1, 2, 4 rows
