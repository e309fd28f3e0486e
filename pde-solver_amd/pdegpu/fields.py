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


def grid_coords(dim: int, n, L) -> np.ndarray:
    """Node coordinates embedded in 3-D, x fastest then y then z (the order of every pdegpu array);
    (x,0,0) / (x,y,0) / (x,y,z) as at fenics_mcp_server.py:321, 450, 718."""
    axes = [np.linspace(0.0, float(L[r]), int(n[r]) + 1) if r < dim else np.zeros(1) for r in range(3)]
    Z, Y, X = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
