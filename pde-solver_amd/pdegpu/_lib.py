"""ctypes binding of libpdegpu.so (include/pdegpu.h).  No CPU fallback: if the HIP library is missing
or no gfx950 device is usable, every solver entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

import numpy as np

PKG_ROOT = Path(__file__).resolve().parent.parent          # .../pde-solver_amd
LIB_PATH = PKG_ROOT / "lib" / "libpdegpu.so"

OK, EINVAL, EHIP, ENOCONV, ENODEV, EUNSUPPORTED = 0, -1, -2, -3, -4, -5
ELEM_P1SPLIT, ELEM_Q1 = 0, 1
IC_CONSTANT, IC_ZERO, IC_NODAL = 0, 1, 2
BC_UNIFORM, BC_DIRECTIONAL = 0, 1
QUANTITY_STRESS, QUANTITY_STRAIN = 0, 1


class PdeGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pdegpu error {code}: {msg}")
        self.code = code


class Grid(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n", C.c_int32 * 3), ("L", C.c_double * 3),
                ("element_mode", C.c_int32), ("reserved", C.c_int32)]


class Heat(C.Structure):
    _fields_ = [("kappa", C.c_double), ("dt", C.c_double), ("num_steps", C.c_int32), ("steady", C.c_int32),
                ("source", C.c_double), ("ic_type", C.c_int32), ("bc_mode", C.c_int32), ("ic_value", C.c_double),
                ("bc_value", C.c_double * 3), ("bc_present", C.c_int32 * 3), ("snapshot_stride", C.c_int32),
                ("rtol", C.c_double), ("max_iter", C.c_int32), ("fast_path", C.c_int32)]


class Weight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("origin", C.c_double * 3), ("core_radius", C.c_double),
                ("core_ratio", C.c_double)]


WEIGHT_ONE, WEIGHT_R, WEIGHT_R2, WEIGHT_R2SIN, WEIGHT_RCYL = 0, 1, 2, 3, 4


class Elasticity(C.Structure):
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("area", C.c_double), ("body", C.c_double * 3),
                ("plane_stress", C.c_int32), ("quantity", C.c_int32), ("rtol", C.c_double),
                ("max_iter", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_unknowns", C.c_int64), ("cg_iterations", C.c_int64),
                ("linear_solves", C.c_int32), ("converged", C.c_int32), ("final_rel_residual", C.c_double),
                ("solve_ms", C.c_double), ("cg_ms", C.c_double), ("kernel_ms", C.c_double * 4),
                ("kernel_launches", C.c_int64 * 4), ("alg_bytes", C.c_double * 4), ("path", C.c_int32),
                ("reserved", C.c_int32)]

    def as_dict(self):
        return {"n_nodes": self.n_nodes, "n_unknowns": self.n_unknowns, "cg_iterations": self.cg_iterations,
                "linear_solves": self.linear_solves, "converged": bool(self.converged),
                "final_rel_residual": self.final_rel_residual, "solve_ms": self.solve_ms, "cg_ms": self.cg_ms,
                "kernel_ms": list(self.kernel_ms), "kernel_launches": list(self.kernel_launches),
                "alg_bytes": list(self.alg_bytes), "path": {1: "fast", 2: "tridiagonal-direct", 3: "fast+boundary", 4: "weighted-marching"}.get(self.path, "general")}


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# every symbol include/pdegpu.h declares: (restype, argtypes)
SYMBOLS = {
    "pdegpu_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "pdegpu_destroy": (None, [_vp]),
    "pdegpu_last_error": (C.c_char_p, [_vp]),
    "pdegpu_device_name": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "pdegpu_version": (C.c_char_p, []),
    "pdegpu_set_tunable": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "pdegpu_marching_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "pdegpu_heat_num_snapshots": (C.c_int, [C.POINTER(Heat)]),
    "pdegpu_heat_solve": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Heat), _dp, _dp, _dp, C.POINTER(Stats)]),
    "pdegpu_heat_solve_weighted": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Weight), C.POINTER(Heat), _dp, _dp, C.POINTER(Stats)]),
    "pdegpu_heat_solve_weighted_ic": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Weight), C.POINTER(Heat), _dp, _dp, _dp, C.POINTER(Stats)]),
    "pdegpu_heat_apply_weighted": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Weight), C.c_double, C.c_double, _dp, _dp, C.c_int]),
    "pdegpu_weighted_rows_host": (C.c_int, [C.POINTER(Grid), C.POINTER(Weight), _dp, _dp, _dp]),
    "pdegpu_mass_solve": (C.c_int, [_vp, C.POINTER(Grid), _dp, _dp, C.c_double, C.c_int, C.POINTER(Stats)]),
    "pdegpu_heat_apply": (C.c_int, [_vp, C.POINTER(Grid), C.c_double, C.c_double, _dp, _dp, _dp, C.c_int]),
    "pdegpu_heat_tables": (C.c_int, [_vp, C.POINTER(Grid), _dp, _dp, _dp]),
    "pdegpu_heat_tables_host": (C.c_int, [C.POINTER(Grid), _dp, _dp, _dp]),
    "pdegpu_elasticity_solve": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Elasticity), _dp, _dp, C.POINTER(Stats)]),
    "pdegpu_elasticity_apply": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Elasticity), _dp, _dp, C.c_int]),
    "pdegpu_elasticity_time_apply": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Elasticity), C.c_int, C.c_int, _dp]),
    "pdegpu_elasticity_tables_host": (C.c_int, [C.POINTER(Grid), C.c_double, C.c_double, _dp]),
    "pdegpu_el_session_create": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Elasticity), C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "pdegpu_el_session_destroy": (None, [_vp]),
    "pdegpu_el_session_begin": (C.c_int, [_vp]),
    "pdegpu_el_session_k1": (C.c_int, [_vp]),
    "pdegpu_el_session_k2": (C.c_int, [_vp]),
    "pdegpu_el_session_flush": (C.c_int, [_vp]),
    "pdegpu_el_session_scalars": (_vp, [_vp]),
    "pdegpu_el_session_scalars_host": (C.c_int, [_vp, _dp]),
    "pdegpu_el_session_plane_doubles": (C.c_int64, [_vp]),
    "pdegpu_el_session_plane_ptr": (_vp, [_vp, C.c_int, C.c_int, C.c_int64]),
    "pdegpu_el_session_get_u": (C.c_int, [_vp, _dp]),
    "pdegpu_el_session_can_split_k1": (C.c_int, [_vp]),
    "pdegpu_el_session_k1_part": (C.c_int, [_vp, C.c_int]),
    "pdegpu_el_session_set_cg_variant": (C.c_int, [_vp, C.c_int]),
    "pdegpu_el_session_cg_variant": (C.c_int, [_vp]),
    "pdegpu_el_session_k1_pre": (C.c_int, [_vp]),
    "pdegpu_elasticity_postprocess": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Elasticity), _dp, _dp]),
    "pdegpu_session_create": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Heat), C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "pdegpu_session_destroy": (None, [_vp]),
    "pdegpu_session_begin": (C.c_int, [_vp, C.c_double]),
    "pdegpu_session_plane_ptr": (_vp, [_vp, C.c_int, C.c_int64]),
    "pdegpu_session_plane_doubles": (C.c_int64, [_vp]),
    "pdegpu_session_needs_p_halo": (C.c_int, [_vp]),
    "pdegpu_session_scalars": (_vp, [_vp]),
    "pdegpu_session_residual": (C.c_int, [_vp]),
    "pdegpu_session_k1": (C.c_int, [_vp]),
    "pdegpu_session_k2": (C.c_int, [_vp]),
    "pdegpu_session_can_split_k1": (C.c_int, [_vp]),
    "pdegpu_session_k1_part": (C.c_int, [_vp, C.c_int]),
    "pdegpu_session_can_split_k2": (C.c_int, [_vp]),
    "pdegpu_session_k2_part": (C.c_int, [_vp, C.c_int]),
    "pdegpu_session_set_cg_variant": (C.c_int, [_vp, C.c_int]),
    "pdegpu_session_cg_variant": (C.c_int, [_vp]),
    "pdegpu_session_sr_poll": (C.c_int, [_vp]),
    "pdegpu_comm_unique_id": (C.c_int, [_vp]),
    "pdegpu_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "pdegpu_comm_destroy": (None, [_vp]),
    "pdegpu_comm_rank": (C.c_int, [_vp]),
    "pdegpu_comm_world": (C.c_int, [_vp]),
    "pdegpu_comm_set_self_loop": (C.c_int, [_vp, C.c_int]),
    "pdegpu_comm_group_begin": (C.c_int, [_vp]),
    "pdegpu_comm_group_end": (C.c_int, [_vp]),
    "pdegpu_session_exchange_post": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pdegpu_session_exchange_begin": (C.c_int, [_vp, _vp, C.c_int]),
    "pdegpu_session_exchange_end": (C.c_int, [_vp, _vp]),
    "pdegpu_session_allreduce": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "pdegpu_session_iterate_slab": (C.c_int, [_vp, _vp, C.c_int]),
    "pdegpu_el_session_exchange_post": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "pdegpu_el_session_exchange_begin": (C.c_int, [_vp, _vp]),
    "pdegpu_el_session_exchange_end": (C.c_int, [_vp, _vp]),
    "pdegpu_el_session_allreduce": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "pdegpu_el_session_iterate_slab": (C.c_int, [_vp, _vp, C.c_int]),
    "pdegpu_session_set_timing": (C.c_int, [_vp, C.c_int]),
    "pdegpu_session_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "pdegpu_session_step_rhs": (C.c_int, [_vp]),
    "pdegpu_session_lift_norm": (C.c_int, [_vp]),
    "pdegpu_session_reset_cg": (C.c_int, [_vp]),
    "pdegpu_session_reduce_local": (C.c_int, [_vp, C.c_int, C.c_int]),
    "pdegpu_session_flush": (C.c_int, [_vp]),
    "pdegpu_session_iterate": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(Stats)]),
    "pdegpu_session_rr": (C.c_int, [_vp, _dp]),
    "pdegpu_session_time_apply": (C.c_int, [_vp, C.c_int, _dp]),
    "pdegpu_session_membench": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "pdegpu_session_debug_time_k1": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int, _dp]),
    "pdegpu_session_get_x": (C.c_int, [_vp, _dp]),
    "pdegpu_session_sync": (C.c_int, [_vp]),
    "pdegpu_grid_num_nodes": (C.c_int64, [C.POINTER(Grid)]),
    "pdegpu_dirichlet_set_host": (C.c_int, [C.POINTER(Grid), C.POINTER(Heat), C.POINTER(C.c_uint8), _dp]),
    "pdegpu_slab_partition": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None
_lock = threading.Lock()
_ctx_cache = {}


def _bind_single_hip_runtime() -> None:
    """One process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64.so.7 next to the system
    one in /opt/rocm (same SONAME): if libpdegpu.so pulls in the system copy first and torch is imported later,
    torch finds "no ROCm-capable device" (measured on the MI355X pool, tools/order_probe.py).  Importing torch
    first makes libpdegpu.so resolve its NEEDED libamdhip64.so.7 to the copy torch already loaded, which is the
    combination bench.py, the multi-GPU driver and the test-suite run.  Without torch installed the system
    runtime is used."""
    if os.environ.get("PDEGPU_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; fails loudly when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise PdeGpuError(ENODEV, f"{LIB_PATH} is missing — build it with `python -c 'import "
                                          f"__graft_entry__ as g; g.build()'` or `make -C {PKG_ROOT}`; "
                                          "there is no CPU fallback")
            _bind_single_hip_runtime()
            lib = C.CDLL(str(LIB_PATH))
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(lib, name)          # AttributeError if the header and the library disagree
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def context(device: int | None = None):
    """One pdegpu context per (process, device).  Raises if no gfx950 device is usable."""
    lib = load_library()
    if device is None:
        device = int(os.environ.get("PDEGPU_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        ctx = _ctx_cache.get(device)
        if ctx is None:
            h = _vp()
            rc = lib.pdegpu_create(C.byref(h), device, None)
            if rc != OK:
                raise PdeGpuError(rc, f"pdegpu_create(device={device}) failed — no usable MI355X/gfx950 device; "
                                      "this path has no CPU fallback")
            ctx = h
            _ctx_cache[device] = ctx
    return ctx


def close_contexts() -> None:
    """Destroy every cached context (its stream included).  Long-lived servers never need this; scripts that run under a
    profiler call it before the interpreter exits so that no HIP handle of this library outlives the tool's finalisation."""
    lib = load_library()
    with _lock:
        for dev in list(_ctx_cache):
            lib.pdegpu_destroy(_ctx_cache.pop(dev))


def check(ctx, rc: int, allow=()):
    if rc != OK and rc not in allow:
        msg = load_library().pdegpu_last_error(ctx)
        raise PdeGpuError(rc, msg.decode() if msg else "")
    return rc


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def make_grid(dim, n, L, element_mode=ELEM_P1SPLIT) -> Grid:
    g = Grid()
    g.dim = dim
    for r in range(3):
        g.n[r] = int(n[r]) if r < dim else 0
        g.L[r] = float(L[r]) if r < dim else 0.0
    g.element_mode = element_mode
    return g


def element_mode_from_env() -> int:
    """PDEGPU_ELEMENT = p1split (default, the reference's discretisation) | q1"""
    v = os.environ.get("PDEGPU_ELEMENT", "p1split").lower()
    if v in ("p1split", "p1"):
        return ELEM_P1SPLIT
    if v == "q1":
        return ELEM_Q1
    raise ValueError(f"PDEGPU_ELEMENT must be p1split or q1, got {v!r}")


# ---- output-volume policy ------------------------------------------------------------------------------------------------
# The reference stores EVERY step (fenics_mcp_server.py:300-315, 694-696, 711-713); at the grid sizes this build solves that is
# hundreds of GB (config 3 at the dispatcher's defaults: 201 states x 1.08 GB = 217 GB).  Policy: the host array may take at most
# PDEGPU_MAX_OUTPUT_GB, default 32 GB — a FIXED number, so that the same call returns arrays of the same shape on every machine that can hold
# them (a host or container with less than twice that still available gets half of what it has: meta says which bound decided).  A
# request beyond it is strided to the smallest stride that fits (first and last state always kept) and the stride actually used is
# recorded in meta; with PDEGPU_OUTPUT_POLICY=error it raises instead, naming snapshot_stride.
DEFAULT_OUTPUT_BUDGET_GB = 32.0


def _memory_available_bytes():
    """what this process may still allocate on the host: MemAvailable, cut by the cgroup's limit minus its usage where one is set; None if unknown"""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
                break
    except OSError:
        pass
    for lim_f, use_f in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                         ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            lim = open(lim_f).read().strip()
            if lim != "max" and int(lim) < (1 << 60):
                room = int(lim) - int(open(use_f).read().strip())
                avail = room if avail is None else min(avail, room)
            break
        except (OSError, ValueError):
            continue
    return avail


def output_budget_bytes(explain: bool = False):
    """the fixed default (or PDEGPU_MAX_OUTPUT_GB), never more than half of what the host can still give — on a small host or container the fixed
    32 GB would be allocated (and pickled) where the request should be strided; explain=True also returns what decided"""
    v = os.environ.get("PDEGPU_MAX_OUTPUT_GB")
    budget, why = int(float(v if v else DEFAULT_OUTPUT_BUDGET_GB) * 1e9), ("PDEGPU_MAX_OUTPUT_GB" if v else "default")
    avail = _memory_available_bytes()
    if avail is not None and avail // 2 < budget:
        budget, why = max(avail // 2, 0), "half of the host memory still available"
    return (budget, why) if explain else budget


def count_snapshots(num_steps: int, stride: int) -> int:
    """states stored for steps 0 .. num_steps with `stride` (the last step is always stored) — pdegpu_heat_num_snapshots in closed form"""
    return num_steps // stride + 1 + (1 if num_steps % stride else 0)


def snapshot_policy(num_steps: int, n_nodes: int, requested_stride: int, steady: bool):
    """-> (stride to use, meta additions).  Raises PdeGpuError when not even the first and last state fit, or when the request
    does not fit and PDEGPU_OUTPUT_POLICY=error."""
    req = max(1, int(requested_stride))
    if steady:
        return req, {"snapshot_stride": req}
    budget, why = output_budget_bytes(explain=True)
    need = count_snapshots(num_steps, req) * n_nodes * 8
    if need <= budget:
        return req, {"snapshot_stride": req}
    msg = (f"{count_snapshots(num_steps, req)} stored states x {n_nodes} nodes x 8 B = {need / 1e9:.1f} GB exceed the output budget "
           f"of {budget / 1e9:.1f} GB ({why}; PDEGPU_MAX_OUTPUT_GB)")
    if os.environ.get("PDEGPU_OUTPUT_POLICY", "auto").lower() == "error":
        raise PdeGpuError(EINVAL, msg + "; pass snapshot_stride=k (or PDEGPU_SNAPSHOT_STRIDE) to keep every k-th state")
    if 2 * n_nodes * 8 > budget:
        raise PdeGpuError(EINVAL, msg + "; not even the first and the last state fit")
    # the count never grows with the stride: bisect for the smallest stride in (req, num_steps] that fits (num_steps itself stores 2 states)
    lo, hi = req, max(req + 1, num_steps)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if count_snapshots(num_steps, mid) * n_nodes * 8 > budget:
            lo = mid
        else:
            hi = mid
    stride = hi
    return stride, {"snapshot_stride": stride, "snapshot_stride_requested": req,
                    "snapshot_policy": f"auto-strided: {msg}; every {stride}-th state (and the last) is kept"}
