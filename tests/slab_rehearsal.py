"""REHEARSAL ONLY (test infrastructure, lives under tests/): the z-slab protocol of
cudafluidsimulator_amd/csrc/mgpu.cpp restated over torch.distributed so that it can run on
CPU ranks (gloo, world 2/3) against the oracle backend.  The product's multi-GPU driver is
the C++ library libsph_mgpu.so (include/sph_mgpu.h); nothing in the package or bench.py
imports this file.

z-slab decomposition of the SPH step across GPUs (one process per GPU).

The reference is single-GPU; this is new design.  The box is cut
into slabs of whole cell layers along z -- z is the slowest digit of the
flattened cell key (simulator.cu:78-82), so a slab is a contiguous range of the
key-sorted particle streams, gravity (y) does not drain slabs, and halo layers
are contiguous too.  Support radius = cell size, so the halo is ONE layer.

Per step and rank (buffers: pos4[2], vel4[2]; `s` sorted, `t` the other pair):

  1. stable partition of own particles by the
     segment their NEW cell key falls in       -> [mig_dn|bnd_lo|interior|bnd_hi|mig_up]
  2. exchange A (RCCL send/recv, ONE round)    : to each neighbour a header (the partition
                                                 bounds, written on the device) and a FIXED
                                                 number of rows holding migrants + boundary
                                                 layer (pos, vel) -- no count exchange first;
                                                 a face that outgrows the fixed size sends
                                                 the rest in a second, exact-size round
  3. assemble t = [H_lo | my mig_dn | from below | mine | from above | my mig_up | H_hi]
     and sort it (stable, by cell key)         -> [halo_lo | owned | halo_hi] + cell table
  4. density for owned; exchange vel4 (carries rho) of the boundary layers
  5. force + integrate for owned               -> new owned state in t

Why that concatenation order: the canonical summation order (oracle) is the
stable sort of the previous GLOBAL order, which is rank order (z-major keys).
Particles arriving from below must therefore precede, and particles from above
follow, the local ones inside a cell; a neighbour's halo cells must list that
neighbour's stayers before (upper halo: after) the particles I just sent it.
With this order N slabs reproduce the single-domain result BIT FOR BIT.

The driver is transport- and backend-agnostic: `DistTransport` uses
torch.distributed P2P (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU);
`run_loopback` steps several slabs inside one process.  The compute backend is
`HipSlabBackend` (libsph_hip.so through the C-ABI); tests inject a CPU backend.
"""
import ctypes as C
import time

import numpy as np
import torch

from cudafluidsimulator_amd import _lib
from cudafluidsimulator_amd._lib import SphError, SphKernelTimes, SphOptions, load_library


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def partition_layers(layer_hist, world, min_layers=2):
    """Cut D cell layers into `world` contiguous slabs with ~equal particle
    counts (cuts on layer boundaries, every slab >= min_layers thick)."""
    hist = np.asarray(layer_hist, dtype=np.int64)
    D = len(hist)
    if world * min_layers > D:
        raise ValueError("too many slabs for the grid")
    cum = np.concatenate([[0], np.cumsum(hist)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        z = int(np.searchsorted(cum, target, side="left"))
        if z > 0 and abs(cum[z - 1] - target) <= abs(cum[min(z, D)] - target):
            z -= 1
        z = max(z, cuts[-1] + min_layers)
        z = min(z, D - (world - r) * min_layers)
        cuts.append(z)
    cuts.append(D)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def layer_of(pos_z, h, D):
    """Cell layer exactly as getGridCell computes it (fp32 divide, truncate)."""
    c = (np.asarray(pos_z, np.float32) / np.float32(h)).astype(np.float32).astype(np.int64)
    return np.clip(c, 0, D - 1)


# --------------------------------------------------------------------------
# compute backend over the C-ABI
# --------------------------------------------------------------------------
class HipSlabBackend:
    """Kernels of libsph_hip.so on ranges of four torch-owned float4 buffers."""

    def __init__(self, settings, capacity, device=0, sweep="list", flags=0):
        self.settings = settings
        self.cap = int(capacity)
        self.device = torch.device("cuda", device)
        self._L = load_library()
        torch.cuda.set_device(self.device)
        self.pos = [torch.zeros((self.cap, 4), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.vel = [torch.zeros((self.cap, 4), dtype=torch.float32, device=self.device) for _ in range(2)]
        opt = SphOptions()
        opt.struct_size = C.sizeof(SphOptions)
        opt.device = device
        opt.math_mode = _lib.SPH_MATH_STRICT
        opt.sweep = _lib.SWEEPS[sweep]
        opt.flags = flags | _lib.SPH_FLAG_EXTERNAL_STATE | _lib.SPH_FLAG_NO_READBACK
        opt.capacity = self.cap
        self._h = C.c_void_p()
        rc = self._L.sph_create(C.byref(settings), C.byref(opt), C.byref(self._h))
        if rc:
            raise SphError(f"sph_create failed ({rc}): {self._L.sph_last_error(None).decode()}")
        self._check(self._L.sph_bind_buffers(self._h, self.pos[0].data_ptr(), self.vel[0].data_ptr(),
                                             self.pos[1].data_ptr(), self.vel[1].data_ptr(),
                                             self.cap), "sph_bind_buffers")
        # run on torch's current stream: stream-ordered with the RCCL send/recv
        self._check(self._L.sph_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "sph_set_stream")

    def _check(self, rc, what):
        if rc:
            raise SphError(f"{what} failed ({rc}): {self._L.sph_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            torch.cuda.synchronize(self.device)
            self._L.sph_destroy(self._h)
            self._h = C.c_void_p()

    def sort(self, src_buf, offset, count, thresholds):
        thr = (C.c_uint32 * len(thresholds))(*[int(t) for t in thresholds])
        out = (C.c_int32 * len(thresholds))()
        self._check(self._L.sph_slab_sort(self._h, src_buf, offset, count, thr, len(thresholds), out),
                    "sph_slab_sort")
        return list(out)

    def partition(self, src_buf, offset, count, thresholds, hdr=None):
        """hdr: int32 device tensor (>= len(thresholds)+1) that receives the bounds and
        `count` on the device, stream-ordered (the exchange's message header)."""
        thr = (C.c_uint32 * len(thresholds))(*[int(t) for t in thresholds])
        out = (C.c_int32 * len(thresholds))()
        self._check(self._L.sph_slab_partition(self._h, src_buf, offset, count, thr, len(thresholds),
                                               out, C.c_void_p(hdr.data_ptr()) if hdr is not None else None),
                    "sph_slab_partition")
        return list(out)

    def copy_segments(self, dst_buf, pieces):
        """pieces: [(pos rows, vel rows, first destination row)] -> rows of pair dst_buf,
        eight pieces per kernel launch."""
        pieces = [p for p in pieces if len(p[0])]
        for a in range(0, len(pieces), 8):
            chunk = pieces[a:a + 8]
            k = len(chunk)
            sp = (C.c_void_p * k)(*[p[0].data_ptr() for p in chunk])
            sv = (C.c_void_p * k)(*[p[1].data_ptr() for p in chunk])
            cnt = (C.c_int32 * k)(*[len(p[0]) for p in chunk])
            dst = (C.c_int32 * k)(*[int(p[2]) for p in chunk])
            self._check(self._L.sph_slab_copy_segments(self._h, dst_buf, k, sp, sv, cnt, dst),
                        "sph_slab_copy_segments")

    def density(self, buf, i0, i1, n_all):
        self._check(self._L.sph_slab_density(self._h, buf, i0, i1, n_all), "sph_slab_density")

    def force(self, buf, i0, i1, n_all):
        self._check(self._L.sph_slab_force(self._h, buf, i0, i1, n_all), "sph_slab_force")

    def kernel_times(self, reset=False):
        kt = SphKernelTimes()
        self._check(self._L.sph_get_kernel_times(self._h, C.byref(kt), 1 if reset else 0),
                    "sph_get_kernel_times")
        return kt

    def sync(self):
        torch.cuda.synchronize(self.device)


# --------------------------------------------------------------------------
# one slab's state machine (no communication inside)
# --------------------------------------------------------------------------
DOWN, UP = -1, +1


class Slab:
    """One rank's slab.  `face_cap` (rows) is the fixed size of an exchange-A message
    per face; it MUST be the same on every rank (default: capacity // 4 of equally
    sized backends).  A face that needs more sends the excess in a second, exact-size
    message, so the value only affects speed."""

    HDR = 8  # int32 per message header: b0, b1, b2, b3, n (+ padding)

    def __init__(self, backend, rank, world, zlo, zhi, D, face_cap=None):
        self.b = backend
        self.rank, self.world = rank, world
        self.zlo, self.zhi, self.D = zlo, zhi, D
        self.has_dn, self.has_up = rank > 0, rank < world - 1
        if (self.has_dn or self.has_up) and zhi - zlo < 2:
            raise ValueError("a slab needs at least two cell layers")
        DD = D * D
        self.thr = [zlo * DD, (zlo + 1) * DD, (zhi - 1) * DD, zhi * DD]
        self.cur, self.off, self.n_own = 0, 0, 0
        self.seg = None
        self.steps = 0
        self.F = int(face_cap) if face_cap is not None else max(1, backend.cap // 4)
        if self.F > backend.cap:
            raise ValueError("face_cap exceeds the slab capacity")
        dev = backend.device
        self.hdr_tx = torch.zeros(self.HDR, dtype=torch.int32, device=dev)
        self.hdr_rx = torch.zeros((2, self.HDR), dtype=torch.int32, device=dev)  # [from dn, from up]
        # landing areas of the fixed-size messages: [0] from below, [1] from above
        self.rx_pos = [torch.zeros((self.F, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        self.rx_vel = [torch.zeros((self.F, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        self.overflows = 0

    # ---- initial state: owned particles in particle-id order ----
    def load(self, pos4, vel4):
        n = len(pos4)
        if n > self.b.cap:
            raise ValueError("slab capacity too small")
        self.cur, self.off, self.n_own = 0, 0, n
        self.b.pos[0][:n].copy_(pos4)
        self.b.vel[0][:n].copy_(vel4)

    def owned(self):
        """(pos4, vel4) views of the owned particles (sorted order of the last step)."""
        a, b = self.off, self.off + self.n_own
        return self.b.pos[self.cur][a:b], self.b.vel[self.cur][a:b]

    # ---- phase 1: split the owned particles into [migrants down | lower boundary
    # layer | interior | upper boundary layer | migrants up] by their NEW cell,
    # previous order kept inside each part (a stable partition, not a sort: the
    # combined array is sorted by key afterwards, and a stable sort of it only
    # needs the right order among EQUAL keys).  The bounds land in hdr_tx on the device.
    def local_sort(self):
        n = self.n_own
        b0, b1, b2, b3 = self.b.partition(self.cur, self.off, n, self.thr, self.hdr_tx)
        if not self.has_dn:
            assert b0 == 0
        if not self.has_up:
            assert b3 == n
        self.s = self.cur ^ 1
        self.seg = (b0, b1, b2, b3, n)

    # Where a message's payload sits.  DOWN message = [migrants down | lower boundary]
    # = rows [0, b1) of the sender's partitioned array; the window sent is rows [0, F).
    # UP message = [upper boundary | migrants up] = rows [b2, n); the window is the LAST
    # F rows, [n-F, n) (rows [0, F) if n < F).  Anything that does not fit the window
    # goes in a second message ("extra").  Both ends compute this from (b0..b3, n).
    def _down_layout(self, b0, b1, b2, b3, n):
        payload = b1
        extra = max(0, payload - self.F)           # rows [F, b1) of the sender
        return payload, 0, extra                   # payload, offset in window, rows in the extra message

    def _up_layout(self, b0, b1, b2, b3, n):
        payload = n - b2
        extra = max(0, payload - self.F)           # rows [b2, n-F) of the sender: the FIRST rows
        if extra:
            return payload, 0, extra
        return payload, (self.F - payload) if n >= self.F else b2, 0

    # ---- phase 2: post exchange A (fixed sizes: no count exchange beforehand) ----
    def plan_exchange_a(self):
        b0, b1, b2, b3, n = self.seg
        s = self.s
        P, V = self.b.pos, self.b.vel
        F = self.F
        sends, recvs = [], []
        if self.has_dn:
            sends += [(DOWN, self.hdr_tx), (DOWN, P[s][0:F]), (DOWN, V[s][0:F])]
            recvs += [(DOWN, self.hdr_rx[0]), (DOWN, self.rx_pos[0]), (DOWN, self.rx_vel[0])]
        if self.has_up:
            w0 = max(n - F, 0)
            sends += [(UP, self.hdr_tx), (UP, P[s][w0:w0 + F]), (UP, V[s][w0:w0 + F])]
            recvs += [(UP, self.hdr_rx[1]), (UP, self.rx_pos[1]), (UP, self.rx_vel[1])]
        return sends, recvs

    # ---- phase 2b: read the neighbours' headers; plan the (rare) overflow messages ----
    def plan_overflow(self):
        b0, b1, b2, b3, n = self.seg
        s = self.s
        P, V = self.b.pos, self.b.vel
        F = self.F
        hdr = self.hdr_rx.tolist() if (self.has_dn or self.has_up) else [[0] * self.HDR] * 2
        self.nb_dn = tuple(hdr[0][:5]) if self.has_dn else None   # neighbour below: its UP message
        self.nb_up = tuple(hdr[1][:5]) if self.has_up else None   # neighbour above: its DOWN message
        sends, recvs = [], []
        self.extra_rx = [None, None]
        # my own excess rows
        if self.has_dn:
            _, _, ex = self._down_layout(b0, b1, b2, b3, n)
            if ex:
                sends += [(DOWN, P[s][F:b1]), (DOWN, V[s][F:b1])]
        if self.has_up:
            _, _, ex = self._up_layout(b0, b1, b2, b3, n)
            if ex:
                sends += [(UP, P[s][b2:b2 + ex]), (UP, V[s][b2:b2 + ex])]
        # the neighbours' excess rows
        dev = self.b.device
        if self.has_dn:
            _, _, ex = self._up_layout(*self.nb_dn)
            if ex:
                self.extra_rx[0] = (torch.empty((ex, 4), dtype=torch.float32, device=dev),
                                    torch.empty((ex, 4), dtype=torch.float32, device=dev))
                recvs += [(DOWN, self.extra_rx[0][0]), (DOWN, self.extra_rx[0][1])]
        if self.has_up:
            _, _, ex = self._down_layout(*self.nb_up)
            if ex:
                self.extra_rx[1] = (torch.empty((ex, 4), dtype=torch.float32, device=dev),
                                    torch.empty((ex, 4), dtype=torch.float32, device=dev))
                recvs += [(UP, self.extra_rx[1][0]), (UP, self.extra_rx[1][1])]
        if sends or recvs:
            self.overflows += 1
        return sends, recvs

    def _payload_rows(self, which, kind, a, b):
        """Pieces (tensor views) holding payload rows [a, b) of the message received
        from below (which=0: an UP message) or above (which=1: a DOWN message)."""
        if b <= a:
            return []
        rx = (self.rx_pos if kind == 0 else self.rx_vel)[which]
        if which == 0:
            payload, off, ex = self._up_layout(*self.nb_dn)
        else:
            payload, off, ex = self._down_layout(*self.nb_up)
        pieces = []
        if which == 0:      # UP message: the extra rows come FIRST, the window holds the rest
            if a < ex:
                pieces.append(self.extra_rx[0][kind][a:min(b, ex)])
            if b > ex:
                lo = max(a, ex) - ex
                pieces.append(rx[off + lo: off + (b - ex)])
        else:               # DOWN message: the window holds rows [0, F), the extra rows follow
            inwin = payload - ex
            if a < inwin:
                pieces.append(rx[off + a: off + min(b, inwin)])
            if b > inwin:
                pieces.append(self.extra_rx[1][kind][max(a, inwin) - inwin: b - inwin])
        return pieces

    # ---- phase 2c: assemble t = [H_lo | my mig_dn | from below | mine | from above | my mig_up | H_hi]
    def assemble(self):
        b0, b1, b2, b3, n = self.seg
        if self.has_dn:
            _, _, nb2, nb3, nn = self.nb_dn
            bnd_from_dn, mig_from_dn = nb3 - nb2, nn - nb3
        else:
            bnd_from_dn = mig_from_dn = 0
        if self.has_up:
            ub0, ub1 = self.nb_up[0], self.nb_up[1]
            mig_from_up, bnd_from_up = ub0, ub1 - ub0
        else:
            mig_from_up = bnd_from_up = 0
        s, t = self.s, self.s ^ 1
        o = [0]
        for c in (bnd_from_dn, b0, mig_from_dn, b3 - b0, mig_from_up, n - b3, bnd_from_up):
            o.append(o[-1] + c)
        self.n_comb = o[-1]
        if self.n_comb > self.b.cap:
            raise SphError("slab capacity exceeded by halo + migrants")
        self.halo_lo_expected = bnd_from_dn + b0
        self.halo_hi_expected = bnd_from_up + (n - b3)
        P, V = self.b.pos, self.b.vel
        pieces = [(P[s][0:b0], V[s][0:b0], o[1]),          # my migrants down
                  (P[s][b0:b3], V[s][b0:b3], o[3]),        # what stays mine
                  (P[s][b3:n], V[s][b3:n], o[5])]          # my migrants up

        def put(dst0, which, a, b):
            at = dst0
            for pp, vv in zip(self._payload_rows(which, 0, a, b), self._payload_rows(which, 1, a, b)):
                pieces.append((pp, vv, at))
                at += len(pp)
            assert at == dst0 + max(b - a, 0)

        # from below: [its upper boundary | its migrants up]
        put(o[0], 0, 0, bnd_from_dn)
        put(o[2], 0, bnd_from_dn, bnd_from_dn + mig_from_dn)
        # from above: [its migrants down | its lower boundary]
        put(o[4], 1, 0, mig_from_up)
        put(o[6], 1, mig_from_up, mig_from_up + bnd_from_up)
        self.b.copy_segments(t, pieces)

    # ---- phase 3: sort the combined array, density, plan exchange B ----
    def combined_sort_and_density(self):
        t = self.s ^ 1
        i0, e_lo, s_hi, i1 = self.b.sort(t, 0, self.n_comb, self.thr)
        self.s = t ^ 1
        if i0 != self.halo_lo_expected or self.n_comb - i1 != self.halo_hi_expected:
            raise SphError(f"rank {self.rank}: halo layers hold particles of other layers "
                           f"({i0} vs {self.halo_lo_expected}, {self.n_comb - i1} vs "
                           f"{self.halo_hi_expected}): a particle crossed more than one cell in z")
        self.rng = (i0, e_lo, s_hi, i1)
        self.b.density(self.s, i0, i1, self.n_comb)
        V = self.b.vel[self.s]
        sends, recvs = [], []
        if self.has_dn:
            sends.append((DOWN, V[i0:e_lo]))
            recvs.append((DOWN, V[0:i0]))
        if self.has_up:
            sends.append((UP, V[s_hi:i1]))
            recvs.append((UP, V[i1:self.n_comb]))
        return sends, recvs

    # ---- phase 4: force + integrate for the owned range ----
    def force(self):
        i0, _, _, i1 = self.rng
        self.b.force(self.s, i0, i1, self.n_comb)
        self.cur, self.off, self.n_own = self.s ^ 1, i0, i1 - i0
        self.steps += 1


# --------------------------------------------------------------------------
# transports
# --------------------------------------------------------------------------
def _nonempty(msgs):
    return [(d, x) for d, x in msgs if x.numel() > 0]


class DistTransport:
    """torch.distributed point-to-point: backend 'nccl' (= RCCL over xGMI) for
    CUDA tensors, 'gloo' for CPU tensors.  Chain topology: rank r talks to r-1
    and r+1 only; no collective is needed on the data path."""

    def __init__(self, dist, rank, world, device, group=None, via_cpu=False):
        self.dist, self.rank, self.world, self.device, self.group = dist, rank, world, device, group
        # via_cpu: bounce every message through host memory (gloo has no CUDA
        # point-to-point).  Only for rehearsing the multi-rank path on ONE GPU.
        self.via_cpu = via_cpu

    def exchange(self, sends, recvs):
        dist = self.dist
        ops, landing = [], []
        for d, x in _nonempty(sends):
            ops.append(dist.P2POp(dist.isend, x.cpu() if self.via_cpu else x, self.rank + d,
                                  group=self.group))
        for d, x in _nonempty(recvs):
            buf = torch.empty(x.shape, dtype=x.dtype) if self.via_cpu else x
            landing.append((x, buf))
            ops.append(dist.P2POp(dist.irecv, buf, self.rank + d, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.via_cpu:
            for x, buf in landing:
                x.copy_(buf)

    def abort(self):
        """This rank cannot go on (capacity, a particle beyond the halo layer): tear the
        process group down so that neighbours blocked in a receive fail with a transport
        error instead of waiting for a message that will never come.  (The C++ driver
        carries a status word in its exchange headers instead, sph_mgpu.h.)"""
        try:
            self.dist.destroy_process_group(self.group)
        except Exception:
            pass


def step_distributed(slab, tr):
    """One step of this rank's slab over a DistTransport: two message rounds (A:
    particles, B: densities); a third only when a face outgrew its fixed-size message.
    A rank whose checks fail raises SphError AFTER closing its end of the transport."""
    try:
        slab.local_sort()
        tr.exchange(*slab.plan_exchange_a())
        sends, recvs = slab.plan_overflow()
        if sends or recvs:
            tr.exchange(sends, recvs)
        slab.assemble()
        tr.exchange(*slab.combined_sort_and_density())
        slab.force()
    except SphError:
        tr.abort()
        raise


def nccl_self_copy(dist, group=None):
    """A `copy` for run_loopback that moves every message of a round through RCCL:
    one rank sends to ITSELF (RCCL implements self send/recv as a device copy), in
    one batch_isend_irecv per round like DistTransport.  With a world of one rank it
    exercises the real tensors (views, offsets, dtypes) against torch's RCCL
    point-to-point API on a one-GPU box."""
    def copy(pairs):
        me = dist.get_rank(group)
        ops = []
        for src, _ in pairs:
            ops.append(dist.P2POp(dist.isend, src, me, group=group))
        for _, dst in pairs:
            ops.append(dist.P2POp(dist.irecv, dst, me, group=group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    return copy


def run_loopback(slabs, steps, copy=None):
    """Step several slabs of ONE process in lock-step (1-GPU / CPU testing):
    messages are delivered by tensor copies in the same order DistTransport
    would deliver them (`copy`: a callable taking the round's (src, dst) pairs,
    default plain tensor copies)."""
    def deliver(all_sends, all_recvs):
        pairs = []
        for r, slab_sends in enumerate(all_sends):
            for d in (DOWN, UP):
                out = [x for dd, x in _nonempty(slab_sends) if dd == d]
                if not out:
                    continue
                peer = r + d
                inn = [x for dd, x in _nonempty(all_recvs[peer]) if dd == -d]
                assert len(out) == len(inn), "send/recv plans disagree"
                for src, dst in zip(out, inn):
                    assert src.shape == dst.shape, (src.shape, dst.shape)
                    pairs.append((src, dst))
        if copy is not None:
            copy(pairs)
        else:
            for src, dst in pairs:
                dst.copy_(src)

    for _ in range(steps):
        for s in slabs:
            s.local_sort()
        plans = [s.plan_exchange_a() for s in slabs]
        deliver([p[0] for p in plans], [p[1] for p in plans])
        plans = [s.plan_overflow() for s in slabs]
        if any(p[0] or p[1] for p in plans):
            deliver([p[0] for p in plans], [p[1] for p in plans])
        for s in slabs:
            s.assemble()
        plans = [s.combined_sort_and_density() for s in slabs]
        deliver([p[0] for p in plans], [p[1] for p in plans])
        for s in slabs:
            s.force()


# --------------------------------------------------------------------------
# initial condition split + bench entry
# --------------------------------------------------------------------------
def make_initial(settings):
    """Host arrays pos4 (id bits in .w) / vel4 of the reference initialiser."""
    import ctypes
    n = settings.numParticles
    pos = np.zeros((n, 3), np.float32)
    rc = load_library().sph_initial_positions(ctypes.byref(settings),
                                              pos.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc:
        raise SphError("sph_initial_positions failed")
    return pack_state(pos, None)


def pack_state(pos, vel, ids=None):
    n = len(pos)
    pos4 = np.zeros((n, 4), np.float32)
    pos4[:, :3] = pos
    pos4[:, 3] = (np.arange(n, dtype=np.uint32) if ids is None else np.asarray(ids, np.uint32)).view(np.float32)
    vel4 = np.zeros((n, 4), np.float32)
    if vel is not None:
        vel4[:, :3] = vel
    return pos4, vel4


def split_initial(pos4, vel4, h, D, world):
    """Static slab bounds from the z-layer histogram and each rank's particles."""
    layers = layer_of(pos4[:, 2], h, D)
    bounds = partition_layers(np.bincount(layers, minlength=D), world)
    parts = []
    for zlo, zhi in bounds:
        m = (layers >= zlo) & (layers < zhi)
        parts.append((np.ascontiguousarray(pos4[m]), np.ascontiguousarray(vel4[m])))
    return bounds, parts


def default_face_cap(pos4, h, D):
    """Rows per fixed-size exchange-A message: 1.25 x the fullest z-layer (one boundary
    layer plus the step's migrants; a fuller face falls back to a second message)."""
    hist = np.bincount(layer_of(pos4[:, 2], h, D), minlength=D)
    return int(1.25 * int(hist.max())) + 4096
