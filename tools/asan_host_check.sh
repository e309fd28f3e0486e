#!/bin/bash
# Host-side AddressSanitizer pass over libpdegpu.so (CPU only; GPU sanitizer runs are not available on the pool).
# Builds an instrumented copy of the library and drives every entry point that needs no GPU: macro-element table
# twins, weighted rows, elasticity block tables, slab partition, snapshot count, and the no-device error path.
set -e
cd "$(dirname "$0")/../pde-solver_amd"
rm -rf build_asan && mkdir -p build_asan
for f in tridiag tables stencil vecops heat elasticity weighted persist2d comm; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-omit-frame-pointer -c csrc/$f.hip -o build_asan/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address build_asan/*.o -o build_asan/libpdegpu_asan.so
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
PDEGPU_NO_TORCH_PRELOAD=1 LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python3 - <<'PY'
import ctypes as C, sys, numpy as np
from pathlib import Path
root = Path.cwd().parent
sys.path[:0] = [str(root), str(root / "pde-solver_amd")]
from pdegpu import _lib as L
L.LIB_PATH = root / "pde-solver_amd" / "build_asan" / "libpdegpu_asan.so"
lib = L.load_library()
for dim, n, Ls in ((1, (5,), (1.0,)), (2, (4, 3), (1.0, 2.0)), (3, (3, 4, 2), (1.0, 0.5, 2.0))):
    g = L.make_grid(dim, n, Ls)
    a, b, c = np.zeros(729), np.zeros(729), np.zeros(27)
    assert lib.pdegpu_heat_tables_host(g, L.dptr(a), L.dptr(b), L.dptr(c)) == 0
    N = int(lib.pdegpu_grid_num_nodes(g))
    for kind in (1, 2) + ((3,) if dim >= 2 else ()) + ((4, 0) if dim == 3 else ()):
        w = L.Weight(); w.kind = kind; w.core_radius, w.core_ratio = (0.3, 5.0) if kind in (0, 4) else (0.0, 1.0)
        rk, rm, ld = np.zeros((N, 15)), np.zeros((N, 15)), np.zeros(N)
        assert lib.pdegpu_weighted_rows_host(g, w, L.dptr(rk), L.dptr(rm), L.dptr(ld)) == 0
    if dim >= 2:
        t = np.zeros(27 * 15 * 9)
        assert lib.pdegpu_elasticity_tables_host(g, 1.0, 1.0, L.dptr(t)) == 0
k0, nk = C.c_int64(), C.c_int64()
for r in range(8):
    assert lib.pdegpu_slab_partition(513, 8, r, C.byref(k0), C.byref(nk)) == 0
for dim, n in ((1, (4,)), (2, (2, 3)), (3, (2, 3, 2)), (3, (4, 2, 3))):
    g = L.make_grid(dim, n, (1.0,) * dim)
    N = int(lib.pdegpu_grid_num_nodes(g))
    hh = L.Heat(); hh.bc_mode = L.BC_DIRECTIONAL
    for r in range(3): hh.bc_present[r], hh.bc_value[r] = 1, float(r + 1)
    m, v = np.zeros(N, np.uint8), np.zeros(N)
    assert lib.pdegpu_dirichlet_set_host(g, hh, m.ctypes.data_as(C.POINTER(C.c_uint8)), L.dptr(v)) == 0
g4 = (C.c_int32 * 4)()
assert lib.pdegpu_marching_geometry(256, 513, 513, 513, 2, 3, g4) == 0
h = L.Heat(); h.num_steps = 200; h.snapshot_stride = 7
assert lib.pdegpu_heat_num_snapshots(h) == 30
# the communicator's argument checks (no GPU: nothing is created)
assert lib.pdegpu_comm_unique_id(None) == L.EINVAL and lib.pdegpu_comm_rank(None) == -1 and lib.pdegpu_comm_world(None) == -1
assert lib.pdegpu_comm_set_self_loop(None, 1) == L.EINVAL and lib.pdegpu_session_iterate_slab(None, None, 1) == L.EINVAL
lib.pdegpu_comm_destroy(None)
ctx = C.c_void_p()
assert lib.pdegpu_create(C.byref(ctx), 0, None) in (0, -4)
print("ASan host pass OK")
PY
rm -rf build_asan
