"""Result containers of the tool boundary — same attribute names as the reference's dataclasses
(fenics_mcp_server.py:168-198), numpy-backed instead of lists of lists (the
reference's own plot tool converts with np.array(...) and accepts arrays, :3453-3455)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict

import numpy as np


@dataclass
class TimeSeriesField:
    """coords [N,3], values [Nt,N], times [Nt], dim, meta — fenics_mcp_server.py:168-183."""
    coords: Any
    values: Any
    times: Any
    dim: int
    meta: Dict[str, Any] = field(default_factory=dict)


@dataclass
class SolveResult:
    """fenics_mcp_server.py:186-191."""
    data_file: str
    dim: int
    meta: Dict[str, Any] = field(default_factory=dict)


@dataclass
class PlotResult:
    """fenics_mcp_server.py:194-197."""
    html_path: str


class GridCoords:
    """Node coordinates [N, 3] of a structured box grid, x fastest then y then z (the order of every pdegpu array), embedded in 3-D as
    (x,0,0) / (x,y,0) / (x,y,z) — fenics_mcp_server.py:321, 450, 718 — WITHOUT holding the array: at 513^3 nodes the coordinates are
    3.24 GB (config 4: 6.5 GB) next to a 1.08 GB field, most of a tool call's wall time when materialised and pickled.  The object is what
    `field.coords` returns; it pickles as its three axes (a few KB), converts on demand — `np.asarray(coords)` / `np.array(coords,
    dtype=float)`, which is all the reference's plotter does with it (:3453-3455) — and answers shape / len / row indexing directly."""
    ndim = 2
    dtype = np.dtype(np.float64)

    def __init__(self, dim: int, n, L, origin=(0.0, 0.0, 0.0)):
        self.dim = int(dim)
        self.n = tuple(int(v) for v in n)
        self.L = tuple(float(v) for v in L)
        self.origin = tuple(float(v) for v in origin)
        self.axes = [self.origin[r] + np.linspace(0.0, self.L[r], self.n[r] + 1) if r < self.dim else np.zeros(1) for r in range(3)]

    def __reduce__(self):
        return (GridCoords, (self.dim, self.n, self.L, self.origin))

    @property
    def grid_shape(self):
        """(NX, NY, NZ) nodes per axis (1 on unused axes)"""
        return tuple(len(a) for a in self.axes)

    @property
    def shape(self):
        nx, ny, nz = self.grid_shape
        return (nx * ny * nz, 3)

    def __len__(self):
        return self.shape[0]

    def rows(self, idx) -> np.ndarray:
        """coordinates of the nodes `idx` (integer array-like), [len(idx), 3]"""
        idx = np.asarray(idx, dtype=np.int64)
        nx, ny, _ = self.grid_shape
        return np.stack([self.axes[0][idx % nx], self.axes[1][(idx // nx) % ny], self.axes[2][idx // (nx * ny)]], axis=-1)

    def __array__(self, dtype=None, copy=None):
        Z, Y, X = np.meshgrid(self.axes[2], self.axes[1], self.axes[0], indexing="ij")
        out = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        return out if dtype is None else out.astype(dtype, copy=False)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            rows, col = key[0], key[1:]
        else:
            rows, col = key, ()
        if isinstance(rows, (int, np.integer)):
            r = self.rows([int(rows) % len(self)])[0]
        else:
            r = self.rows(np.arange(len(self))[rows])
        return r[(Ellipsis,) + col] if col else r

    def __repr__(self):
        return f"GridCoords(dim={self.dim}, nodes={self.grid_shape}, L={self.L}, origin={self.origin})"


def grid_coords(dim: int, n, L) -> GridCoords:
    """Node coordinates of the box grid as the lazy GridCoords (np.asarray gives the [N, 3] array of the reference's layout)."""
    return GridCoords(dim, n, L)
