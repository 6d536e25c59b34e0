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


@pytest.mark.parametrize("n", [1500000, 1572863, 1572864, 1572865])
def test_sizes_around_the_tile_size_switch(n):
    """sort.hip sorts 1024-key tiles below RS_SMALL_N keys and 4096-key tiles from there on: the sizes either side
    of the switch, cell-sorted-like keys (long runs of equal digits: the run-head histogram's case) with a tenth moved."""
    rng = np.random.default_rng(n)
    keys = (np.arange(n, dtype=np.int64) * 800000 // n + 100000).astype(np.uint32)
    move = rng.random(n) < 0.1
    keys[move] = (keys[move].astype(np.int64) + rng.choice([-10000, -100, -1, 1, 100, 10000], int(move.sum()))).astype(np.uint32)
    check(keys)


def test_adversarial_digit_patterns():
    """Every key in ONE first-pass digit (a single bucket takes the whole tile), every key in one second-pass digit,
    strictly alternating digits (no run longer than one key), descending runs, and the largest 20-bit key."""
    n = 70001
    i = np.arange(n, dtype=np.uint32)
    check((i % 977) << 10)                 # low digit constant 0
    check(i % 1024)                        # high digit constant 0
    check(((i & 1) * 1023) | ((i % 7) << 10))
    check(np.where(i % 2 == 0, 0, (1 << 20) - 1))
    check((n - i) % 1024 + ((i // 4096) % 1024 << 10))
    check(np.full(n, (1 << 20) - 1))
