"""Synthetic data generator (bench plumbing): deterministic, device-independent, shaped as SURVEY.md 8d asks."""
import numpy as np

from relate_amd import synth


def test_deterministic_and_chunk_invariant():
    a = synth.make_dataset(32, 6000, n_related=4, chunk=1 << 16)
    b = synth.make_dataset(32, 6000, n_related=4, chunk=1000)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_shape_and_statistics():
    g, pos, chr_, rel = synth.make_dataset(64, 20000, n_related=6)
    assert g.shape == (64, 20000) and g.dtype == np.uint8 and g.max() <= 3
    miss = (g == 0).mean()
    assert 0.0003 < miss < 0.003
    d = np.diff(pos)
    assert (d < 0).sum() == 21                      # 22 chromosomes: positions restart
    assert (np.diff(chr_) >= 0).all() and chr_[0] == 1 and chr_[-1] == 22
    assert len(rel) == 6
    # planted IBD2 segments make related pairs share more genotypes than unrelated ones
    same_rel = np.mean([(g[i] == g[j]).mean() for i, j, _, _ in rel])
    same_unrel = np.mean([(g[40 + k] == g[50 + k]).mean() for k in range(6)])
    assert same_rel > same_unrel + 0.05


def test_all_pairs_order_is_the_references():
    p = synth.all_pairs(5)
    assert p.tolist()[:5] == [[0, 1], [0, 2], [0, 3], [0, 4], [1, 2]] and len(p) == 10
