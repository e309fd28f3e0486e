"""Test-only compute engine for pdegpu.dist.SlabCG: the slab operations of the HIP session (pdegpu_session_*),
restated with numpy on CPU tensors so that the N>1 driver logic (partition, ghost exchange, plane ownership in
the dot products, scalar flow) runs under the `gloo` backend without a GPU.  3-D heat, every boundary node
Dirichlet, P1 7-point stiffness stencil (weights from the oracle)."""
import numpy as np
import torch

from oracle import cg_port

SC_PQ_LOCAL, SC_RR_LOCAL, SC_PQ, SC_RR, SC_RR_PREV = 0, 1, 2, 3, 4
SC_QQ, SC_RR_POLL = 4, 6          # single-reduction form (pdegpu_session_set_cg_variant 1): message = scal[2:5], polls read scal[6]


class NumpySlabEngine:
    supports_sr = True
    sr = False

    def set_variant(self, sr):
        self.sr = bool(sr)
        self._k2_rr, self._skip_x, self._rr_slot = None, False, SC_RR

    def _k1_scalars(self):
        """(alpha_{n-1}, beta_n) as the HIP kernels form them (csrc/dot_finish.h: cg_scalars_in)"""
        if self.sr:
            pq, rr, qq = (float(self.scal[i]) for i in (SC_PQ, SC_RR, SC_QQ))
            a = rr / pq if (pq > 0 and rr > 0) else 0.0
            rr_next = a * a * qq - rr
            return (0.0 if self._skip_x else a), (rr_next / rr if (a > 0 and rr_next > 0) else 0.0)
        pq, rr, rr_prev = (float(self.scal[i]) for i in (SC_PQ, SC_RR, SC_RR_PREV))
        return (rr_prev / pq if (pq > 0 and rr_prev > 0) else 0.0), (rr / rr_prev if rr_prev > 0 else 0.0)

    def _k1_done(self, pq, qq):
        """K1's dot products: classic -> the local p.Ap slot; single-reduction -> the block {p.Ap, r.r of the last K2, Ap.Ap} (this slab's sums)"""
        if not self.sr:
            self.scal[SC_PQ_LOCAL] = pq
            return
        self.scal[SC_PQ], self.scal[SC_QQ] = pq, qq
        # r.r of the residual this K1 started from, THIS slab's share (the message is summed over ranks): K2's, or on the first
        # iteration the residual pass's
        self.scal[SC_RR] = self._k2_rr if self._k2_rr is not None else self.scal[SC_RR_LOCAL]
        self._k2_rr, self._skip_x, self._rr_slot = None, False, SC_RR

    def sr_poll(self):
        if self._k2_rr is None:
            self._rr_slot = SC_RR
        else:
            self.scal[SC_RR_POLL] = self._k2_rr
            self._rr_slot = SC_RR_POLL

    def __init__(self, n, Ls, kappa, source, T_boundary, k0, nk, dt=None):
        """dt = None: steady system kappa K u = f m;  dt > 0: one backward-Euler step matrix M + dt kappa K (15-point)"""
        self.n, self.k0, self.nk = n, k0, nk
        self.NX, self.NY, self.NZ = n[0] + 1, n[1] + 1, n[2] + 1
        h = [Ls[r] / n[r] for r in range(3)]
        self.w = cg_port.p1_weights(h, 0.0, kappa) if dt is None else cg_port.p1_weights(h, 1.0, dt * kappa)
        self.wM = cg_port.p1_weights(h, 1.0, 0.0)
        self.dt, self.source = dt, source
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
        self._k2_rr, self._skip_x, self._rr_slot = None, False, SC_RR

    # --- helpers ---
    def _apply(self, v, w=None):
        """A v on the owned planes (needs the ghost planes of v), zero on Dirichlet rows; 15-point when the diagonal weights
        +-(1,1,0), +-(1,0,1), +-(0,1,1), +-(1,1,1) are non-zero"""
        w = self.w if w is None else w
        c = v[1:-1, 1:-1, 1:-1]
        out = np.zeros((self.nk, self.NY, self.NX))
        acc = (w[0] * c + w[1] * (v[1:-1, 1:-1, :-2] + v[1:-1, 1:-1, 2:]) +
               w[2] * (v[1:-1, :-2, 1:-1] + v[1:-1, 2:, 1:-1]) + w[3] * (v[:-2, 1:-1, 1:-1] + v[2:, 1:-1, 1:-1]))
        if np.any(np.asarray(w[4:]) != 0.0):
            acc = acc + (w[4] * (v[1:-1, 2:, 2:] + v[1:-1, :-2, :-2]) + w[5] * (v[2:, 1:-1, 2:] + v[:-2, 1:-1, :-2]) +
                         w[6] * (v[2:, 2:, 1:-1] + v[:-2, :-2, 1:-1]) + w[7] * (v[2:, 2:, 2:] + v[:-2, :-2, :-2]))
        out[:, 1:-1, 1:-1] = acc
        out[~self.free[self.own]] = 0.0
        return out

    def _reduce_rr(self, val):
        self.scal[SC_RR_PREV] = self.scal[SC_RR]
        self.scal[SC_RR_LOCAL] = val

    # --- engine interface used by SlabCG ---
    def begin(self, x_fill):
        self.scal.zero_()
        self._k2_rr, self._skip_x, self._rr_slot = None, False, SC_RR
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
        self._k2_rr, self._rr_slot = None, SC_RR

    def k1(self):
        alpha_prev, beta = self._k1_scalars()
        p_old = self.p[self.pcur].numpy()
        if alpha_prev != 0.0:
            self.x.numpy()[self.own] += alpha_prev * p_old[self.own]
        pn = self.r.numpy() + beta * p_old                       # ghost planes included (exchanged r, p)
        q = self._apply(pn)
        self.p[1 - self.pcur].numpy()[self.own] = pn[self.own]
        self.q.numpy()[self.own] = q
        self.pcur ^= 1
        self._k1_done(float((pn[self.own] * q).sum()), float((q * q).sum()))

    def k1_part(self, part):
        """the split K1 of the HIP session: part 0 may only touch planes whose stencil reads no ghost plane — checked here by
        running it on copies of r and p whose ghost planes are NaN (the receives may land in the real ones meanwhile)"""
        if part == 0:
            self._pq_parts = self._qq_parts = 0.0
            self._alpha_prev, self._beta = self._k1_scalars()
            lo = 1 if self.k0 > 0 else 0
            hi = 1 if self.k0 + self.nk < self.NZ else 0
            r, p_old = self.r.numpy().copy(), self.p[self.pcur].numpy().copy()
            if lo:
                r[0] = p_old[0] = np.nan
            if hi:
                r[-1] = p_old[-1] = np.nan
            self._k1_planes(lo, self.nk - hi, r, p_old)
            self._edges = ([0] if lo else []) + ([self.nk - 1] if hi and not (lo and self.nk == 1) else [])
            return
        for lk in self._edges:
            self._k1_planes(lk, lk + 1, self.r.numpy(), self.p[self.pcur].numpy())
        self.pcur ^= 1
        self._k1_done(self._pq_parts, self._qq_parts)

    def _k1_planes(self, a, b, r, p_old):
        """K1 on the local planes [a, b)"""
        if b <= a:
            return
        sl = slice(a, b + 2)                                     # planes a-1 .. b of the ghosted arrays
        pn = r[sl] + self._beta * p_old[sl]
        sub = NumpySlabEngine.__new__(NumpySlabEngine)
        sub.w, sub.nk, sub.NY, sub.NX = self.w, b - a, self.NY, self.NX
        sub.free, sub.own = self.free[sl], slice(1, b - a + 1)
        q = sub._apply(pn)
        own = slice(a + 1, b + 1)
        assert np.isfinite(q).all(), "part 0 of the split K1 read a ghost plane"
        if self._alpha_prev != 0.0:
            self.x.numpy()[own] += self._alpha_prev * p_old[own]
        self.p[1 - self.pcur].numpy()[own] = pn[1:-1]
        self.q.numpy()[own] = q
        self._pq_parts += float((pn[1:-1] * q).sum())
        self._qq_parts += float((q * q).sum())

    # transient steps (pdegpu_session_step_rhs / lift_norm / reset_cg)
    def step_rhs(self):
        u = self.x.numpy() + self.dt * self.source
        u[~self.inside] = 0.0
        w = self.wM
        v = u
        acc = (w[0] * v[1:-1, 1:-1, 1:-1] + w[1] * (v[1:-1, 1:-1, :-2] + v[1:-1, 1:-1, 2:]) +
               w[2] * (v[1:-1, :-2, 1:-1] + v[1:-1, 2:, 1:-1]) + w[3] * (v[:-2, 1:-1, 1:-1] + v[2:, 1:-1, 1:-1]) +
               w[4] * (v[1:-1, 2:, 2:] + v[1:-1, :-2, :-2]) + w[5] * (v[2:, 1:-1, 2:] + v[:-2, 1:-1, :-2]) +
               w[6] * (v[2:, 2:, 1:-1] + v[:-2, :-2, 1:-1]) + w[7] * (v[2:, 2:, 2:] + v[:-2, :-2, :-2]))
        b = np.zeros(self.b.shape)
        b[1:-1, 1:-1, 1:-1] = acc
        b[~self.free] = 0.0
        self.b.copy_(torch.from_numpy(b))

    def lift_norm(self):
        g = np.where(self.free, 0.0, self.Tb)
        g[~self.inside] = 0.0
        r = self.b.numpy()[self.own] - self._apply(g)
        r[~self.free[self.own]] = 0.0
        self._reduce_rr(float((r * r).sum()))
        self._k2_rr, self._rr_slot = None, SC_RR

    def reset_cg(self):
        self.scal.zero_()
        self.pcur = 0
        self._k2_rr, self._skip_x, self._rr_slot = None, False, SC_RR

    def x_halo_tensors(self, which):
        return [self.plane_tensor(0, which)]

    def rr(self):
        if self.sr:
            assert self._k2_rr is None or self._rr_slot == SC_RR_POLL, "rr after k2 in the single-reduction form: sr_poll + all-reduce first"
            return float(self.scal[self._rr_slot])
        return float(self.scal[SC_RR])

    def k2(self):
        pq, rr = float(self.scal[SC_PQ]), float(self.scal[SC_RR])
        alpha = rr / pq if pq > 0 else 0.0
        r = self.r.numpy()
        r[self.own] -= alpha * self.q.numpy()[self.own]
        if self.sr:
            self._k2_rr = float((r[self.own] ** 2).sum())       # stays local until the next K1's message (or a poll)
        else:
            self._reduce_rr(float((r[self.own] ** 2).sum()))

    def k2_part(self, part):
        """the split K2 of the HIP session: part 0 = the planes next to the ghosts (after it the driver posts the next exchange: what travels must be
        final here), part 1 = the planes in between + the r.r sum"""
        lo = 1 if self.k0 > 0 else 0
        hi = 1 if self.k0 + self.nk < self.NZ else 0
        edges = sorted(set(([0] if lo else []) + ([self.nk - 1] if hi else [])))
        r, q = self.r.numpy(), self.q.numpy()
        if part == 0:
            pq, rr = float(self.scal[SC_PQ]), float(self.scal[SC_RR])
            self._k2_alpha = rr / pq if pq > 0 else 0.0
            for lk in edges:
                r[1 + lk] -= self._k2_alpha * q[1 + lk]
            return
        for lk in range(self.nk):
            if lk not in edges:
                r[1 + lk] -= self._k2_alpha * q[1 + lk]
        if self.sr:
            self._k2_rr = float((r[self.own] ** 2).sum())
        else:
            self._reduce_rr(float((r[self.own] ** 2).sum()))

    def flush(self):
        pq, rr_prev = float(self.scal[SC_PQ]), float(self.scal[SC_RR if self.sr else SC_RR_PREV])
        alpha = rr_prev / pq if (pq > 0 and rr_prev > 0) else 0.0
        self.x.numpy()[self.own] += alpha * self.p[self.pcur].numpy()[self.own]
        if self.sr:
            self._skip_x = True
        else:
            self.scal[SC_PQ] = 0.0

    def local_to_global(self, which):
        if which == 0:
            if not self.sr:                                       # (single-reduction form: K1 wrote its sums into the message block itself)
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
        if which == 0 and self.sr:
            return (SC_PQ, SC_QQ + 1)
        return (SC_PQ, SC_PQ + 1) if which == 0 else (SC_RR, SC_RR + 1)

    def owned_x(self):
        return self.x.numpy()[self.own].copy()
