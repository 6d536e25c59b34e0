"""CPU compute backend for cudafluidsimulator_amd.slab built on the ORACLE --
test infrastructure only (lives under tests/, injected into the slab driver by
tests).  Same three operations as HipSlabBackend on CPU torch tensors."""
import numpy as np
import torch

from oracle import oracle as O


class OracleSlabBackend:
    def __init__(self, settings, capacity):
        self.settings = O.make_settings(settings.numParticles, settings.randomInit)
        self.cap = int(capacity)
        self.device = torch.device("cpu")
        self.pos = [torch.zeros((self.cap, 4), dtype=torch.float32) for _ in range(2)]
        self.vel = [torch.zeros((self.cap, 4), dtype=torch.float32) for _ in range(2)]
        self.cs = self.ce = None
        self.num_cells = int(self.settings.numCellsPerDim) ** 3

    def sort(self, src_buf, offset, count, thresholds):
        P, V = self.pos[src_buf].numpy(), self.vel[src_buf].numpy()
        p = P[offset:offset + count].copy()
        v = V[offset:offset + count].copy()
        keys = O.cell_keys(self.settings, np.ascontiguousarray(p[:, :3]))
        perm = O.stable_sort(keys, self.num_cells)
        self.pos[src_buf ^ 1].numpy()[:count] = p[perm]
        self.vel[src_buf ^ 1].numpy()[:count] = v[perm]
        sk = keys[perm]
        self.cs, self.ce = O.cell_table(sk, self.num_cells)
        return [int(np.searchsorted(sk, t, side="left")) for t in thresholds]

    def partition(self, src_buf, offset, count, thresholds, hdr=None):
        P, V = self.pos[src_buf].numpy(), self.vel[src_buf].numpy()
        p = P[offset:offset + count].copy()
        v = V[offset:offset + count].copy()
        keys = O.cell_keys(self.settings, np.ascontiguousarray(p[:, :3]))
        cls = np.searchsorted(np.asarray(thresholds, dtype=np.uint32), keys, side="right")
        perm = np.argsort(cls, kind="stable")
        self.pos[src_buf ^ 1].numpy()[:count] = p[perm]
        self.vel[src_buf ^ 1].numpy()[:count] = v[perm]
        self.cs = self.ce = None
        bounds = [int((cls <= k).sum()) for k in range(len(thresholds))]
        if hdr is not None:
            hdr[:len(bounds) + 1] = torch.tensor(bounds + [count], dtype=torch.int32)
        return bounds

    def copy_segments(self, dst_buf, pieces):
        for pp, vv, at in pieces:
            if len(pp):
                self.pos[dst_buf][at:at + len(pp)].copy_(pp)
                self.vel[dst_buf][at:at + len(vv)].copy_(vv)

    def density(self, buf, i0, i1, n_all):
        pos = np.ascontiguousarray(self.pos[buf].numpy()[:n_all, :3])
        rho, _ = O.density(self.settings, pos, self.cs, self.ce, i0, i1)
        self.vel[buf].numpy()[i0:i1, 3] = rho[i0:i1]

    def force(self, buf, i0, i1, n_all):
        P, V = self.pos[buf].numpy(), self.vel[buf].numpy()
        pos = np.ascontiguousarray(P[:n_all, :3])
        vel = np.ascontiguousarray(V[:n_all, :3])
        rho = np.ascontiguousarray(V[:n_all, 3])
        prs = np.maximum(np.float32(0), np.float32(1.0) * (rho - np.float32(1000.0))).astype(np.float32)
        f = O.force(self.settings, pos, vel, rho, prs, self.cs, self.ce, i0, i1)
        O.integrate(self.settings, pos, vel, f, rho, i0, i1)
        Pn, Vn = self.pos[buf ^ 1].numpy(), self.vel[buf ^ 1].numpy()
        Pn[i0:i1, :3] = pos[i0:i1]
        Pn[i0:i1, 3] = P[i0:i1, 3]
        Vn[i0:i1, :3] = vel[i0:i1]
        Vn[i0:i1, 3] = rho[i0:i1]

    def sync(self):
        pass

    def close(self):
        pass
