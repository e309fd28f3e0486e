"""Per-workgroup timeline of the marching elasticity apply (k_el_fast) from the 100 MHz clock: when workgroups start, how long one lives, what the
tail looks like.  usage: python tools/el_trace.py [nx ny nz] [key=value ...]   (writes gpurun_out/el_trace_<n>.txt and prints a summary)"""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
n = [int(v) for v in sys.argv[1:] if "=" not in v] or [256, 64, 64]
out = ROOT / "gpurun_out" / f"el_trace_{n[0]}x{n[1]}x{n[2]}.txt"
out.parent.mkdir(exist_ok=True)
os.environ["PDEGPU_EL_TRACE_FILE"] = str(out)
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
for kv in (v for v in sys.argv[1:] if "=" in v):
    k, v = kv.split("=")
    L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), int(v)))
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
g = L.make_grid(3, tuple(n), (1.0, 0.25, 0.25))
ms = C.c_double()
L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 50, C.byref(ms)))
print(f"{n}: {ms.value * 1e3:.1f} us per apply (50 launches back to back)")
print(open(out).readline().strip())
t = np.loadtxt(out, comments="#")
b, st, en, xcc, hw = t[:, 0], t[:, 1] * 0.01, t[:, 2] * 0.01, t[:, 3].astype(int), t[:, 4].astype(int)
dur = en - st
print(f"workgroups {len(b)}; kernel span {en.max():.2f} us; starts: median {np.median(st):.2f}, 90 % {np.percentile(st, 90):.2f}, last {st.max():.2f} us")
print(f"lifetime of a workgroup: min {dur.min():.2f}, median {np.median(dur):.2f}, 90 % {np.percentile(dur, 90):.2f}, max {dur.max():.2f} us")
edges = np.arange(0, en.max() + 1, 1.0)
alive = [(np.sum((st <= x) & (en > x))) for x in edges]
print("resident workgroups at t = 0, 1, 2 ... us:", " ".join(str(a) for a in alive))
first = st < np.percentile(st, 50)
print(f"lifetime of the first half to start {np.median(dur[first]):.2f} us, of the second half {np.median(dur[~first]):.2f} us")
cu = (xcc * 4096) + (hw >> 8 & 0xff)
print(f"distinct (XCD, SE, SH, CU) seen: {len(set(cu.tolist()))}; workgroups per CU: max {np.bincount(np.unique(cu, return_inverse=True)[1]).max()}")
# lifetimes by position of the workgroup (marching part): which y-group / z-chunk the stragglers are
hdr = open(out).readline().split()
hdr = dict(zip(hdr[1::2], (int(v) for v in hdr[2::2])))
nbf, nbb, nyg, nzc = hdr["nb_fast"], hdr["nb_bnd"], hdr["nyg"], hdr["nzc"]
m = (b >= nbb) if not hdr["bnd_last"] else (b < nbf)
bm = (b[m] - (0 if hdr["bnd_last"] else nbb)).astype(int)
if nbf % 8 == 0:
    bm = (bm & 7) * (nbf >> 3) + (bm >> 3)
d = dur[m]
per_chunk = hdr["ntx"] * nyg
yg, zc = (bm // hdr["ntx"]) % nyg, nzc - 1 - bm // per_chunk
print("boundary workgroups:", " ".join(f"{v:.1f}" for v in dur[~m]))
print("by y-group (median/max us):", " ".join(f"{np.median(d[yg == y]):.1f}/{d[yg == y].max():.1f}" for y in range(nyg) if np.any(yg == y)))
print("by z-chunk (median/max us):", " ".join(f"{np.median(d[zc == z]):.1f}/{d[zc == z].max():.1f}" for z in range(nzc) if np.any(zc == z)))
# phases of a marching workgroup (wave 0's stamps): entry -> prologue done -> first three planes landed -> z-loop done -> sums begin -> end
ph = t[m][:, 5:9] * 0.01
ok = ph[:, 0] > 0
if ok.any():
    s0, e0 = st[m][ok], en[m][ok]
    seg = np.stack([ph[ok, 0] - s0, ph[ok, 1] - ph[ok, 0], ph[ok, 2] - ph[ok, 1], ph[ok, 3] - ph[ok, 2], e0 - ph[ok, 3]], axis=1)
    names = ["prologue (args, weights, table fill + barrier)", "first three planes land", "z-loop", "after the loop (plane NZ-1 of the last chunk)", "block sums + partials"]
    for k, nm in enumerate(names):
        print(f"  {nm:48s} median {np.median(seg[:, k]):5.2f}  90 % {np.percentile(seg[:, k], 90):5.2f}  max {seg[:, k].max():5.2f} us")
