"""Test-only compute engine for pdegpu.dist.SlabCG: the slab operations of the HIP session (pdegpu_session_*),
restated with numpy on CPU tensors so that the N>1 driver logic (partition, ghost exchange, plane ownership in
the dot products, scalar flow) runs under the `gloo` backend without a GPU.  3-D heat, every boundary node
Dirichlet, P1 7-point stiffness stencil (weights from the oracle)."""
import numpy as np
import torch

from oracle import cg_port

SC_PQ_LOCAL, SC_RR_LOCAL, SC_PQ, SC_RR, SC_RR_PREV = 0, 1, 2, 3, 4


class NumpySlabEngine:
    def __init__(self, n, Ls, kappa, source, T_boundary, k0, nk):
        self.n, self.k0, self.nk = n, k0, nk
        self.NX, self.NY, self.NZ = n[0] + 1, n[1] + 1, n[2] + 1
        h = [Ls[r] / n[r] for r in range(3)]
        self.w = cg_port.p1_weights(h, 0.0, kappa)
        self.rhs_const = source * h[0] * h[1] * h[2]
        self.Tb = T_boundary
        shape = (nk + 2, self.NY, self.NX)                       # planes -1 .. nk
        self.x, self.r, self.q, self.b = (torch.zeros(shape, dtype=torch.float64) for _ in range(4))
        self.p = [torch.zeros(shape, dtype=torch.float64), torch.zeros(shape, dtype=torch.float64)]
        self.pcur = 0
        self.scal = torch.zeros(8, dtype=torch.float64)
        kg = np.arange(k0 - 1, k0 + nk + 1)
        free = np.zeros(shape, bool)
        free[:, 1:-1, 1:-1] = True
        free[(kg <= 0) | (kg >= self.NZ - 1)] = False
        self.inside = (kg >= 0) & (kg <= self.NZ - 1)            # planes that exist in the global grid
        self.free = free
        self.own = slice(1, nk + 1)

    # --- helpers ---
    def _apply(self, v):
        """A v on the owned planes (needs the ghost planes of v), zero on Dirichlet rows"""
        w = self.w
        c = v[1:-1, 1:-1, 1:-1]
        out = np.zeros((self.nk, self.NY, self.NX))
        out[:, 1:-1, 1:-1] = (w[0] * c + w[1] * (v[1:-1, 1:-1, :-2] + v[1:-1, 1:-1, 2:]) +
                              w[2] * (v[1:-1, :-2, 1:-1] + v[1:-1, 2:, 1:-1]) + w[3] * (v[:-2, 1:-1, 1:-1] + v[2:, 1:-1, 1:-1]))
        out[~self.free[self.own]] = 0.0
        return out

    def _reduce_rr(self, val):
        self.scal[SC_RR_PREV] = self.scal[SC_RR]
        self.scal[SC_RR_LOCAL] = val

    # --- engine interface used by SlabCG ---
    def begin(self, x_fill):
        self.scal.zero_()
        for t in self.p:
            t.zero_()
        self.pcur = 0
        x = np.full(self.x.shape, float(x_fill))
        x[~self.free] = self.Tb
        x[~self.inside] = 0.0
        self.x.copy_(torch.from_numpy(x))
        b = np.where(self.free, self.rhs_const, 0.0)
        self.b.copy_(torch.from_numpy(b))

    def residual(self):
        r = self.b.numpy()[self.own] - self._apply(self.x.numpy())
        r[~self.free[self.own]] = 0.0
        self.r.numpy()[self.own] = r
        self._reduce_rr(float((r * r).sum()))

    def k1(self):
        pq, rr, rr_prev = (float(self.scal[i]) for i in (SC_PQ, SC_RR, SC_RR_PREV))
        alpha_prev = rr_prev / pq if (pq > 0 and rr_prev > 0) else 0.0
        beta = rr / rr_prev if rr_prev > 0 else 0.0
        p_old = self.p[self.pcur].numpy()
        if alpha_prev != 0.0:
            self.x.numpy()[self.own] += alpha_prev * p_old[self.own]
        pn = self.r.numpy() + beta * p_old                       # ghost planes included (exchanged r, p)
        q = self._apply(pn)
        self.p[1 - self.pcur].numpy()[self.own] = pn[self.own]
        self.q.numpy()[self.own] = q
        self.pcur ^= 1
        self.scal[SC_PQ_LOCAL] = float((pn[self.own] * q).sum())

    def k2(self):
        pq, rr = float(self.scal[SC_PQ]), float(self.scal[SC_RR])
        alpha = rr / pq if pq > 0 else 0.0
        r = self.r.numpy()
        r[self.own] -= alpha * self.q.numpy()[self.own]
        self._reduce_rr(float((r[self.own] ** 2).sum()))

    def flush(self):
        pq, rr_prev = float(self.scal[SC_PQ]), float(self.scal[SC_RR_PREV])
        alpha = rr_prev / pq if (pq > 0 and rr_prev > 0) else 0.0
        self.x.numpy()[self.own] += alpha * self.p[self.pcur].numpy()[self.own]
        self.scal[SC_PQ] = 0.0

    def local_to_global(self, which):
        if which == 0:
            self.scal[SC_PQ] = self.scal[SC_PQ_LOCAL]
        else:
            self.scal[SC_RR] = self.scal[SC_RR_LOCAL]

    def scalars(self):
        return self.scal

    def plane_tensor(self, vec, which):
        t = {0: self.x, 1: self.r, 2: self.p[self.pcur], 3: self.q}[vec]
        return t[which + 1].view(-1)                             # contiguous plane, shares storage

    # the generalised driver protocol (same as GpuSlabEngine / GpuElasticitySlabEngine)
    def halo_tensors(self, which):
        return [self.plane_tensor(1, which), self.plane_tensor(2, which)]

    def global_slots(self, which):
        return (SC_PQ, SC_PQ + 1) if which == 0 else (SC_RR, SC_RR + 1)

    def owned_x(self):
        return self.x.numpy()[self.own].copy()
