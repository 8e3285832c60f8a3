"""The oracle's `intersect` against the reference's own vectors (src/pseudoaligner.rs:542-571) and property
(src/pseudoaligner.rs:573-586)."""
import numpy as np

import helpers

# src/pseudoaligner.rs:544-559
VECS = [
    [1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2, 3], [1, 4, 5], [7, 8, 9], [9], [], [1, 2, 3, 6, 7, 8, 9], [1, 7, 8, 9, 10],
    [10, 15, 20], [21, 22, 23], [0], [0, 1000, 5000], [0, 1000, 1000001], [5], [100000000], [1, 23, 45, 1000001, 100000000],
]


def test_intersect_known_vectors():
    for a in VECS:
        for b in VECS:
            assert helpers.oracle_intersect(a, b) == sorted(set(a) & set(b))
            assert helpers.oracle_intersect(b, a) == sorted(set(a) & set(b))


def test_intersect_property():
    rng = np.random.RandomState(20180101)
    for _ in range(1000):   # proptest: 1000 cases of vec(0..100, 0..5000) sorted+dedup'd
        a = np.unique(rng.randint(0, 100, rng.randint(0, 5000))).tolist()
        b = np.unique(rng.randint(0, 100, rng.randint(0, 5000))).tolist()
        want = sorted(set(a) & set(b))
        assert helpers.oracle_intersect(a, b) == want
        assert helpers.oracle_intersect(b, a) == want


def test_intersect_wide_values():
    rng = np.random.RandomState(7)
    for _ in range(200):
        a = np.unique(rng.randint(0, 2**31, rng.randint(0, 300))).tolist()
        b = np.unique(np.concatenate([rng.choice(a, min(len(a), 20)) if a else [], rng.randint(0, 2**31, 50)]).astype(np.int64)).tolist()
        assert helpers.oracle_intersect(a, b) == sorted(set(a) & set(b))
