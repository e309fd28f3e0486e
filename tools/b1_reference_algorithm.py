"""B1 of BASELINE.md §3: the REFERENCE ALGORITHM timed on the host cores — P1 assembly on the simplex split + sparse
LU factor + solve per backward-Euler step, i.e. what `solve(a == L, u, bcs)` does every step at fenics_mcp_server.py:309-318,
438-447, 707-716 — through the restatement oracle/fem_p1.py (FEniCS 2019.1.0 itself is not installable offline).
One step = assemble K, M (vectorised numpy) + splu(A_ff) + one triangular solve; single process (scipy's SuperLU is serial).
Prints one JSON document; sizes are bounded so the default run stays near 2 minutes (--full adds 48^3)."""
import json, os, sys, time
from pathlib import Path
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT)]
from oracle import fem_p1 as o


def time_case(name, coords, cells, dir_mask, dt):
    n = coords.shape[0]
    t0 = time.perf_counter()
    out = o.assemble_scalar(coords, cells)
    K, M = out[0], out[1]
    t_asm = time.perf_counter() - t0
    free = ~dir_mask
    A = (M + dt * K).tocsr()
    t0 = time.perf_counter()
    Aff = A[free][:, free].tocsc()
    lu = spla.splu(Aff)
    t_fac = time.perf_counter() - t0
    b = (M @ np.full(n, 20.0))[free]
    t0 = time.perf_counter()
    lu.solve(b)
    t_sol = time.perf_counter() - t0
    step = t_asm + t_fac + t_sol
    return {"case": name, "nodes": int(n), "unknowns": int(free.sum()), "assemble_s": t_asm, "factor_s": t_fac, "solve_s": t_sol,
            "s_per_step": step, "dof_per_s": float(free.sum()) / step, "lu_nnz": int(lu.L.nnz + lu.U.nnz)}


def main():
    res = {"what": "reference algorithm (assemble + sparse LU + solve per step), restatement oracle/fem_p1.py, single process",
           "host_cores": os.cpu_count(), "cases": []}
    # config 1: 1-D, 1000 elements (the reference's own CPU-runnable case), all 200 steps
    t0 = time.perf_counter()
    f = o.solve_heat_1d(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 200)
    res["config1_full_200_steps_s"] = time.perf_counter() - t0
    res["config1_u_mid_end"] = float(f.values[200, 500])
    for n in (128, 256, 512, 1024):
        c, cells = o.rectangle_mesh(1.0, 1.0, n, n)
        x, y = c[:, 0], c[:, 1]
        mask = (np.isclose(x, 0) | np.isclose(x, 1) | np.isclose(y, 0) | np.isclose(y, 1))
        res["cases"].append(time_case(f"2d_{n}x{n}", c, cells, mask, 0.0025))
        print(json.dumps(res["cases"][-1]), file=sys.stderr, flush=True)
    for n in ((16, 24, 32, 40, 48) if "--full" in sys.argv else (16, 24, 32, 40)):      # 48^3 alone takes ~2.5 min
        c, cells = o.box_mesh(1.0, 1.0, 1.0, n, n, n)
        mask = np.zeros(c.shape[0], bool)
        for d in range(3):
            mask |= np.isclose(c[:, d], 0) | np.isclose(c[:, d], 1)
        res["cases"].append(time_case(f"3d_{n}^3", c, cells, mask, 0.0025))
        print(json.dumps(res["cases"][-1]), file=sys.stderr, flush=True)
    # fitted exponent of the 3-D step time in the unknown count, and the (flagged) extrapolation to 511^3
    c3 = [r for r in res["cases"] if r["case"].startswith("3d")]
    p = np.polyfit(np.log([r["unknowns"] for r in c3[1:]]), np.log([r["s_per_step"] for r in c3[1:]]), 1)
    res["fit_3d"] = {"exponent": float(p[0]), "extrapolated_s_per_step_at_511^3": float(np.exp(np.polyval(p, np.log(511.0 ** 3)))),
                     "note": "extrapolation only; the LU fill (nnz ~ N^1.6 here) would not fit in memory long before 511^3"}
    pm = np.polyfit(np.log([r["unknowns"] for r in c3[1:]]), np.log([r["lu_nnz"] for r in c3[1:]]), 1)
    res["fit_3d"]["lu_nnz_exponent"] = float(pm[0])
    res["fit_3d"]["extrapolated_lu_bytes_at_511^3"] = float(np.exp(np.polyval(pm, np.log(511.0 ** 3))) * 12)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
