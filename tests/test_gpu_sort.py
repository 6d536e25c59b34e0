"""The grid build's hand-written radix sort against numpy's stable argsort
(bit-exact: integer/index work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check(keys, bits=20):
    from cudafluidsimulator_amd.simulator import sort_check
    keys = np.asarray(keys, dtype=np.uint32)
    perm, sk = sort_check(keys, key_bits=bits)
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(sk, keys[want])
    assert np.array_equal(perm, want)  # stability: ties keep input order


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 12289, 100000])
def test_random_keys_all_tile_edges(n):
    rng = np.random.default_rng(n)
    check(rng.integers(0, 1000000, n))


def test_empty():
    from cudafluidsimulator_amd.simulator import sort_check
    perm, sk = sort_check(np.zeros(0, np.uint32))
    assert len(perm) == 0


def test_all_equal_sorted_reversed():
    n = 50000
    check(np.full(n, 123456))
    check(np.arange(n) % 1000000)
    check((n - np.arange(n)) * 7 % 1000000)


def test_few_distinct_heavy_collisions():
    rng = np.random.default_rng(1)
    check(rng.choice([5, 70000, 999999, 65536, 255, 256], 200003))


def test_21bit_keys_and_large():
    rng = np.random.default_rng(2)
    check(rng.integers(0, 1 << 21, 300007), bits=21)
    check(rng.integers(0, 1000000, 4194304))
