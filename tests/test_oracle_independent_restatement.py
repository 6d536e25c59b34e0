"""A SECOND, separately written strict-fp32 restatement of the reference's step
 that the C oracle must match bit for bit.

Why: oracle/sph_oracle.c and the HIP kernels were written from the same reading of
simulator.cu, and the reference ships nothing that could arbitrate (no tests, no
fixtures, CUDA-only source).  This file breaks that common mode as far as the
environment allows: it is written from the reference's expressions again, in a
different shape -- numpy arrays, one pair LIST per particle instead of nested cell
loops, sequential sums as float32 cumulative sums -- and shares no code with the
oracle.  CPU only, small n, a dense block so that the pressure term is on.

Reference lines restated here:
  cell hash             simulator.cu:57-82
  densityKernel         simulator.cu:84-97     density/pressure  :149-190
  pressureKernel        simulator.cu:99-117    viscosityKernel   :119-130
  force accumulation    simulator.cu:192-256   integration       :258-318
Canonical neighbour order: cells z-outer, y, x-inner as the loops at
:163-176; inside a cell ascending index of the stable sort by flattened cell key.
"""
import numpy as np
import pytest

from helpers import assert_bit_equal, dense_block
from oracle import oracle as O

F = np.float32

# simulator.h:6-12, simulator.cu:13-14, main.cpp:57-63 -- typed exactly as there
PI = F(3.14159265)
MASS = F(0.02)
GAS_CONSTANT = F(1.0)
REST_DENSITY = F(1000.0)
VISCOSITY = F(1.0)
GRAVITY = F(-9.8)
ELASTICITY = F(0.5)
EPS_F = F(1e-4)
H = F(0.1)
H_POW_6 = F(float(H) ** 6)          # (float)pow((double)h, 6)
H_POW_9 = F(float(H) ** 9)
V_COEFF = F(45.0) / (PI * H_POW_6)
D_COEFF = F(315.0) / (F(64.0) * PI * H_POW_9)
BOX = F(10.0)
CELLS = 100
DT = F(0.01)


def seqsum(terms):
    """((0 + t0) + t1) + ... with every partial sum rounded to fp32."""
    if len(terms) == 0:
        return F(0.0)
    return np.cumsum(np.asarray(terms, dtype=F), dtype=F)[-1]


def restated_step(pos, vel):
    """One simulate() on particle-id ordered input; returns new pos, vel, rho, force
    (all particle-id ordered)."""
    pos = np.array(pos, dtype=F)
    vel = np.array(vel, dtype=F)
    n = len(pos)
    # --- getGridCell: (int)(p / h), flattenGridCoord: x + y*100 + z*100*100
    cell = (pos / H).astype(F).astype(np.int64)
    key = cell[:, 0] + cell[:, 1] * CELLS + cell[:, 2] * CELLS * CELLS
    order = np.argsort(key, kind="stable")
    members = {}
    for slot, pid in enumerate(order):
        members.setdefault(int(key[pid]), []).append(int(pid))

    def neighbours(pid):
        cx, cy, cz = (int(v) for v in cell[pid])
        out = []
        for dz in (-1, 0, 1):
            z = cz + dz
            if z < 0 or z >= CELLS:
                continue
            for dy in (-1, 0, 1):
                y = cy + dy
                if y < 0 or y >= CELLS:
                    continue
                for dx in (-1, 0, 1):
                    x = cx + dx
                    if x < 0 or x >= CELLS:
                        continue
                    out.extend(members.get(x + y * CELLS + z * CELLS * CELLS, ()))
        return np.asarray(out, dtype=np.int64)

    nb = [neighbours(i) for i in range(n)]
    h2 = H * H

    # --- density (A.3)
    rho = np.zeros(n, F)
    for i in range(n):
        d = pos[i] - pos[nb[i]]
        dist2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        diff = h2 - dist2
        w = ((D_COEFF * diff) * diff) * diff
        w = np.where(dist2 > h2, F(0.0), w).astype(F)
        rho[i] = seqsum(MASS * w)
    rho = np.maximum(rho, EPS_F)
    prs = np.maximum(F(0.0), GAS_CONSTANT * (rho - REST_DENSITY)).astype(F)

    # --- forces (A.4): per neighbour first the pressure term, then the viscosity term
    force = np.zeros((n, 3), F)
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(n):
            j = nb[i]
            d = pos[i] - pos[j]
            dist2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            dist = np.sqrt(dist2)
            f_pressure = ((-MASS) * (prs[i] + prs[j])) / (F(2.0) * rho[j])
            scale = (((-V_COEFF) * (H - dist)) * (H - dist)) / dist
            gate_p = (dist2 > h2) | (dist < EPS_F)
            k = np.where(gate_p[:, None], F(0.0), d * scale[:, None]).astype(F)
            term_p = (k * f_pressure[:, None]).astype(F)
            lap = np.where((dist > H) | (dist < EPS_F), F(0.0), V_COEFF * (H - dist)).astype(F)
            f_visc = ((VISCOSITY * MASS) * lap) / rho[j]
            term_v = ((vel[j] - vel[i]) * f_visc[:, None]).astype(F)
            both = np.empty((2 * len(j), 3), F)
            both[0::2] = term_p
            both[1::2] = term_v
            for a in range(3):
                force[i, a] = seqsum(both[:, a])

    # --- integration (A.5)
    v = vel.copy()
    v[:, 0] = v[:, 0] + (DT * force[:, 0]) / rho
    v[:, 1] = v[:, 1] + DT * (force[:, 1] / rho + GRAVITY)
    v[:, 2] = v[:, 2] + (DT * force[:, 2]) / rho
    p = (pos + DT * v).astype(F)
    hi = BOX - H
    for a in range(3):
        low = p[:, a] < H
        high = (~low) & (p[:, a] > hi)
        p[:, a] = np.where(low, H, np.where(high, hi, p[:, a]))
        v[:, a] = np.where(low | high, v[:, a] * (-ELASTICITY), v[:, a])
    v = np.where(np.abs(v) < EPS_F, F(0.0), v).astype(F)
    return p.astype(F), v, rho, force


def oracle_step(pos, vel):
    sim = O.OracleSim(len(pos), False)
    sim.upload(pos, vel)
    sim.step()
    d = sim.download(want_force=True)
    sim.close()
    return d


def check(pos, vel, what):
    p, v, rho, force = restated_step(pos, vel)
    d = oracle_step(pos, vel)
    assert_bit_equal(d["rho"], rho, what + " rho")
    assert_bit_equal(d["force"], force, what + " force")
    assert_bit_equal(d["vel"], v, what + " vel")
    assert_bit_equal(d["pos"], p, what + " pos")
    return d


def test_constants_agree_with_the_oracles_settings():
    s = O.make_settings(10, False)
    assert F(s.h) == H and F(s.v_kernel_coeff) == V_COEFF and F(s.d_kernel_coeff) == D_COEFF
    assert F(s.timestep) == DT and F(s.boxDim) == BOX and int(s.numCellsPerDim) == CELLS


def test_dense_block_with_pressure_on():
    """12^3 particles 0.025 apart, resting on the floor with random velocities:
    rho > REST_DENSITY, so pressureKernel's 1/r branch and both force terms are live,
    and the bottom layer hits the wall clamp."""
    pos = dense_block(12, origin=(4.0, 0.1, 4.0), jitter=0.004, seed=5)
    pos[:, 1] = np.maximum(pos[:, 1], F(0.1))
    rng = np.random.default_rng(6)
    vel = rng.uniform(-1.5, 1.5, pos.shape).astype(F)
    d = check(pos, vel, "dense block")
    assert (d["rho"] > 1000).sum() > len(pos) // 2, "pressure must be active in this case"
    assert (d["pos"][:, 1] == F(0.1)).any(), "some particles must hit the floor clamp"


def test_random_cloud_and_coincident_particles():
    """Sparse random cloud (most cells hold 0-2 particles, many empty neighbour
    cells), plus exact duplicates and near-duplicates: dist < EPS_F gates
    (simulator.cu:110,125)."""
    rng = np.random.default_rng(11)
    pos = rng.uniform(2.0, 3.5, (1500, 3)).astype(F)
    pos[100:110] = pos[90:100]                       # coincident pairs: dist == 0
    pos[200:210] = pos[190:200] + F(3e-5)            # closer than EPS_F
    pos[300:310] = pos[290:300] + F(0.02)
    vel = rng.uniform(-3, 3, pos.shape).astype(F)
    check(pos, vel, "random cloud")


@pytest.mark.parametrize("seed", [1, 2])
def test_particles_near_the_walls(seed):
    """Box corners: neighbour cells outside the grid are skipped (simulator.cu:165-173),
    clamps on both sides of every axis, velocity dead zone."""
    rng = np.random.default_rng(seed)
    lo = rng.uniform(0.1, 0.35, (400, 3)).astype(F)
    hi = rng.uniform(9.65, 9.9, (400, 3)).astype(F)
    pos = np.concatenate([lo, hi]).astype(F)
    vel = rng.uniform(-30, 30, pos.shape).astype(F)
    vel[::7] = rng.uniform(-2e-4, 2e-4, (len(vel[::7]), 3)).astype(F)
    check(pos, vel, "walls")
