"""examples/*.c — compiled hosts of the drop-in boundary: include/pdegpu.h must be plain C99 (no C++, no torch types), and a program that links
only libpdegpu.so must be able to run the tool-level solve and the slab loop with the library-issued communication."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIBDIR = ROOT / "pde-solver_amd" / "lib"


def _build(name, out_dir):
    exe = out_dir / name
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(ROOT / "examples" / f"{name}.c"),
           "-L", str(LIBDIR), "-lpdegpu", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return exe


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    import __graft_entry__ as ge
    from pdegpu import _lib
    if not _lib.LIB_PATH.exists():
        ge.build()
    d = tmp_path_factory.mktemp("c_examples")
    return {name: _build(name, d) for name in ("box_heat", "slab_loop")}


def test_header_is_c99_and_the_examples_link(built):
    """-std=c99 -pedantic -Wall -Wextra -Werror against the header alone and against libpdegpu.so"""
    assert all(p.exists() for p in built.values())


@pytest.mark.gpu
def test_box_heat_from_c_matches_the_python_tool(lib, gpu_ctx, built):
    from pdegpu import heat
    out = subprocess.run([str(built["box_heat"]), "32"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"centre (\S+)\s+iterations (\d+)\s+converged (\d)", out.stdout)
    assert m and int(m.group(3)) == 1 and int(m.group(2)) > 0, out.stdout
    ref = heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 32, 32, 32, 1.0, 0.0, 0.0, 0.01, 1, steady=True, source_type="constant", source_value=1.0, rtol=1e-12)
    centre = ref.values[-1].reshape(33, 33, 33)[16, 16, 16]
    assert abs(float(m.group(1)) - centre) < 1e-8 * abs(centre)


@pytest.mark.gpu
def test_slab_loop_from_c_converges(lib, gpu_ctx, built):
    """one rank: the unique id made locally, a real (one-rank) RCCL communicator, chunks of 16 iterations per C call until ||r|| <= 1e-8 ||r_0||"""
    out = subprocess.run([str(built["slab_loop"]), "48"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"iterations (\d+)\s+rr/rr0 (\S+)", out.stdout)
    assert m and 16 <= int(m.group(1)) < 2000 and float(m.group(2)) <= 1e-16, out.stdout
