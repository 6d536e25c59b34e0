"""Shared test helpers: seeded inputs and bit-exact comparison."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = bits(a) != bits(b)
    if bad.any():
        idx = np.argwhere(bad)[:5]
        msg = "; ".join(f"{tuple(i)}: {a[tuple(i)]!r} vs {b[tuple(i)]!r}" for i in idx)
        raise AssertionError(f"{what}: {int(bad.sum())} of {a.size} values differ bitwise: {msg}")


def dense_block(n_side, spacing=0.025, origin=(4.0, 1.0, 4.0), jitter=0.0, seed=0):
    """A block dense enough (rho > REST_DENSITY) to switch the pressure term on
    (SURVEY.md Appendix B item 2): the reference's own initialisers never do
    inside 100 steps."""
    rng = np.random.default_rng(seed)
    g = np.arange(n_side, dtype=np.float32) * np.float32(spacing)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    pos = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.float32)
    pos += np.asarray(origin, dtype=np.float32)
    if jitter:
        pos += rng.uniform(-jitter, jitter, pos.shape).astype(np.float32)
    return np.ascontiguousarray(pos, dtype=np.float32)


def random_state(n, seed, lo=0.5, hi=9.5, vmax=0.5):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    vel = rng.uniform(-vmax, vmax, (n, 3)).astype(np.float32)
    return pos, vel


def clustered_state(n, seed):
    """Skewed occupancy: half the particles in a thin floor layer (what gravity
    produces late in a run)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(1.0, 9.0, (n, 3)).astype(np.float32)
    k = n // 2
    pos[:k, 1] = rng.uniform(0.1, 0.16, k).astype(np.float32)
    pos[:k, 0] = rng.uniform(3.0, 4.0, k).astype(np.float32)
    pos[:k, 2] = rng.uniform(3.0, 4.0, k).astype(np.float32)
    vel = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    return pos, vel
