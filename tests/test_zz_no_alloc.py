"""relate_hip_like_noalloc (include/relate_hip.h): the likelihood entry point for callers that cannot allocate.
Called through the C ABI with device pointers and checked against relate_hip_like on the same points -- like()
at fixed parameters, 1e-10 relative (the bar of tests/test_gpu_parity.py) -- on calls in a row that shrink, over the
three launch shapes (cut along the SNP axis = chunk products in the scratch array, two-sided = half vectors in the
scratch array, single sweep), with the device's free memory unchanged across the calls."""
import numpy as np
import pytest

import pathcheck as pc

pytestmark = pytest.mark.gpu


def _points(P, E, seed):
    rng = np.random.default_rng(seed)
    par = np.empty((P, E, 3))
    par[:, :, 0] = rng.uniform(0.002, 0.12, size=(P, E))
    par[:, :, 1] = rng.uniform(0.0, 0.4, size=(P, E))
    par[:, :, 2] = rng.uniform(0.0, 0.6, size=(P, E))
    par[0, 0] = (0.001, 0.0, 0.0)       # uLike / poLike lanes and a rejected point (k0 < 0)
    par[0, E - 1] = (0.001, 0.0, 1.0)
    par[P - 1, 0] = (0.05, 0.7, 0.6)
    return par


def test_like_noalloc_matches_like_on_shrinking_calls():
    import torch
    import relate_amd
    from relate_amd import synth
    N, M = 96, 6000
    g, pos, chr_, _ = synth.make_dataset(N, M, n_related=8, device="cuda:0")
    opt = dict(pc.options("options.all.pair.txt"))
    H = relate_amd.Relate(0)
    H.init_globals(g, pos, chr_, opt)
    allp = synth.all_pairs(N)
    dev = torch.device("cuda:0")

    def run(pairs, E, seed, expect_chunked):
        P = len(pairs)
        par = _points(P, E, seed)
        want = H.like(pairs, par)                      # stages the block (allocating entry point)
        d_par = torch.from_numpy(par).to(dev)
        d_out = torch.full((P, E), -7.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        H.like_noalloc(P, d_par.data_ptr(), E, d_out.data_ptr())
        assert torch.cuda.mem_get_info(0)[0] == free0   # nothing was allocated or freed on the device
        assert (H.last_chunks()[0] > 1) == expect_chunked
        got = d_out.cpu().numpy()
        rej = want == 1e7
        assert rej.any() and np.array_equal(got == 1e7, rej)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=0)
        return d_par, d_out, want

    # cut along the SNP axis, then a smaller call on the same block right after it (fewer points per pair)
    big = allp[:1200]
    d_par, d_out, want = run(big, 5, 1, True)
    d_out.fill_(-7.0)
    H.like_noalloc(len(big), d_par.data_ptr(), 2, d_out.data_ptr())   # reads the first 2 x P points of the same buffer
    want2 = H.like(big, d_par.cpu().numpy().reshape(-1, 3)[:2 * len(big)].reshape(len(big), 2, 3))
    got2 = d_out.cpu().numpy().reshape(-1)
    np.testing.assert_allclose(got2[:2 * len(big)], want2.reshape(-1), rtol=1e-10, atol=0)
    assert (got2[2 * len(big):] == -7.0).all()                       # and writes nothing beyond its P x E values
    # a smaller block: the scratch still holds the previous call's products
    run(allp[5:45], 3, 2, True)
    # two-sided sweep (45 056 .. 65 536 lanes) and the plain single sweep above that
    run(allp[:4560], 10, 3, False)
    run(allp[:4560], 15, 4, False)
    # a block the handle has not staged is refused, and so is a call beyond the stated maximum
    with pytest.raises(relate_amd.RelateError, match="previous"):
        H.like_noalloc(17, d_par.data_ptr(), 1, d_out.data_ptr())
    with pytest.raises(relate_amd.RelateError, match="at most"):
        H.like_noalloc(4560, d_par.data_ptr(), 300, d_out.data_ptr())
    # After a fit: relate_hip_fit may stage the last live pairs of a block again, side by side (the tail compaction),
    # and then NO block of P pairs is staged any more -- the call must say so (never compute on other pairs' records);
    # if the block survived the fit, the values must be those of relate_hip_like.  Either way a relate_hip_like on
    # the block brings it back.
    blk = allp[:4560]
    H.fit(blk)
    par = _points(len(blk), 3, 9)
    d_par = torch.from_numpy(par).to(dev)
    d_out = torch.full((len(blk), 3), -7.0, dtype=torch.float64, device=dev)
    try:
        H.like_noalloc(len(blk), d_par.data_ptr(), 3, d_out.data_ptr())
        intact = True
    except relate_amd.RelateError as e:
        intact = False
        assert "staged" in str(e)
        assert (d_out.cpu().numpy() == -7.0).all()
    want = H.like(blk, par)
    if intact:
        np.testing.assert_allclose(d_out.cpu().numpy(), want, rtol=1e-10, atol=0)
    d_out.fill_(-7.0)
    H.like_noalloc(len(blk), d_par.data_ptr(), 3, d_out.data_ptr())
    np.testing.assert_allclose(d_out.cpu().numpy(), want, rtol=1e-10, atol=0)
    print("block intact after the fit: %s" % intact)
