"""ctypes loader of oracle/_build/libcgport.so (see cg_port.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SO = HERE / "_build" / "libcgport.so"
_lib = None


def load():
    global _lib
    if _lib is None:
        # pin the OpenMP threads before libgomp reads its environment: the first-touch placement of cg_port.c only holds while
        # a thread stays on its core
        import os
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
        if not SO.exists():
            subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
        _lib = C.CDLL(str(SO))
        _lib.cgport_run.restype = C.c_double
        _lib.cgport_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double,
                                    C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.cgport_max_threads.restype = C.c_int
    return _lib


def p1_weights(h, cM, cK):
    """interior node stencil of cM*M + cK*K for the Kuhn-split P1 space (checked against
    element assembly in tests/test_oracle_known_answers.py)"""
    hx, hy, hz = h
    v = hx * hy * hz
    return np.array([cM * v * 2 / 5 + cK * 2 * (hy * hz / hx + hx * hz / hy + hx * hy / hz),
                     cM * v / 20 - cK * hy * hz / hx, cM * v / 20 - cK * hx * hz / hy, cM * v / 20 - cK * hx * hy / hz,
                     cM * v / 30, cM * v / 30, cM * v / 30, cM * v / 20])


def run(nodes, w, rhs_const, bc_value, iters, nthreads=0, want_x=False):
    lib = load()
    NX, NY, NZ = nodes
    w = np.ascontiguousarray(w, dtype=np.float64)
    full15 = int(np.any(w[4:] != 0.0))
    rr = np.zeros(iters + 1)
    x = np.empty(NX * NY * NZ) if want_x else None
    dp = C.POINTER(C.c_double)
    secs = lib.cgport_run(NX, NY, NZ, w.ctypes.data_as(dp), full15, rhs_const, bc_value, iters, nthreads,
                          x.ctypes.data_as(dp) if want_x else None, rr.ctypes.data_as(dp))
    if secs < 0:
        raise MemoryError("cgport_run: allocation failed")
    return secs, rr, x


def max_threads():
    return load().cgport_max_threads()
