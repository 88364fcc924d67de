"""The last-workgroup election behind every fused "partials -> final value in the same launch" kernel (TD / actor /
augmented-Lagrangian scalars, the dopri5 mode-0 / mode-1 controller): csrc/common.h publish_and_elect orders "partials
published" before "ticket taken" with relaxed agent-scope atomics and gfx950 behaviour instead of fences (the contract is
written out there).  This drives it alone — thousands of workgroups over all eight XCDs, tickets drawn in a scrambled
order, values salted per launch so that a stale or late partial changes the sum, hundreds of launches back to back on
the same buffers — and holds the elected workgroup's sums against host arithmetic, exactly (integers in fp32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def expected(n_blocks, n_vals, salt):
    b = np.arange(n_blocks, dtype=np.uint64)[:, None]
    k = np.arange(n_vals, dtype=np.uint64)[None, :]
    return ((b * 31 + k * 7 + np.uint64(salt) * 13) % 251).sum(axis=0).astype(np.float64)


@pytest.mark.parametrize("n_blocks,n_vals", [(4096, 4), (4096, 15), (1024, 64), (37, 4), (1, 3), (8191, 20),
                                             (768, 2), (4099, 2), (17, 2), (1, 2)])      # n_vals 2: the two-level election
def test_elected_sums_match_the_host_over_many_launches(n_blocks, n_vals):
    from nlbac_amd import _lib
    from nlbac_amd.arena import stream_ptr
    launches = 300
    partials = torch.full((n_blocks * n_vals,), -1.0, device="cuda")
    ticket = torch.zeros(2 + n_blocks // 16, dtype=torch.int32, device="cuda")      # (two-level: 1 + ceil(n_blocks / 16) words)
    out = torch.zeros(launches, n_vals + 1, device="cuda")
    s = stream_ptr()
    for i in range(launches):        # back to back, no host sync: launch i + 1 overwrites what launch i's elected block reads
        _lib.call("nlbac_elect_selftest", partials.data_ptr(), ticket.data_ptr(), out[i].data_ptr(), n_blocks, n_vals,
                  1000 + i, s)
    torch.cuda.synchronize()
    res = out.cpu().numpy().astype(np.float64)
    assert int(ticket.abs().sum().item()) == 0
    for i in range(launches):
        assert res[i, n_vals] == 1.0, "launch %d: %g workgroups were elected" % (i, res[i, n_vals])
        np.testing.assert_array_equal(res[i, :n_vals], expected(n_blocks, n_vals, 1000 + i), "launch %d" % i)


def test_add_cols_plus_is_add_cols_followed_by_the_sum():
    """nlbac_add_cols_plus (SimulatedCars: the three gradients w.r.t. x_t+1 in one launch) == nlbac_add_cols followed by
    nlbac_axpby(1, dst, 1, add), bit for bit."""
    import torch
    from nlbac_amd import _lib
    from nlbac_amd.arena import stream_ptr
    g = torch.Generator().manual_seed(3)
    n, ld, B = 2 * 777, 10, 777
    dst = torch.randn(n, ld, generator=g).cuda()
    src = torch.randn(B, 4, generator=g).cuda()
    add = torch.randn(n, ld, generator=g).cuda()
    a, b = dst.clone(), dst.clone()
    s = stream_ptr()
    _lib.call("nlbac_add_cols", a.data_ptr(), ld, 4, src.data_ptr(), 4, 4, B, s)
    _lib.call("nlbac_axpby", 1.0, a.data_ptr(), 1.0, add.data_ptr(), n * ld, a.data_ptr(), s)
    _lib.call("nlbac_add_cols_plus", b.data_ptr(), ld, 4, src.data_ptr(), 4, 4, B, add.data_ptr(), n, s)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    ref = dst.clone()
    ref[:B, 4:8] += src
    ref += add
    assert torch.equal(b, ref)
