"""sr_op_conv3x3_wgrad_noalloc (the entry point that allocates nothing; include/sr_hip.h) against the existing
sr_op_conv3x3_bwd on the same inputs, through the C ABI.  No reference counterpart: the reference never
touches device memory itself.  Two calls in a row share the static scratch array; the second one is smaller, so
stale partials of the first would show.  Both entry points launch the same kernels with the same slice plan, so the
results must agree to the last bit (the summation order is fixed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sh():
    import sr_hip
    assert sr_hip.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return sr_hip


def _case(sh, torch, seed, B, H, W, Cin, Cout, dtype, want_db=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = rng.standard_normal((3, 3, Cin, Cout), dtype=np.float32) * 0.1
    dy = rng.standard_normal((B, H, W, Cout), dtype=np.float32)
    _, dw_ref, db_ref = sh.conv3x3_bwd(x, w, dy, dtype=dtype, want_dx=False)
    d_x, d_dy = torch.from_numpy(x).cuda(), torch.from_numpy(dy).cuda()
    d_dw = torch.full((3, 3, Cin, Cout), float("nan"), device="cuda")
    d_db = torch.full((Cout,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    sh.conv3x3_wgrad_noalloc(B, H, W, Cin, Cout, d_x.data_ptr(), d_dy.data_ptr(), d_dw.data_ptr(),
                             d_db.data_ptr() if want_db else None, dtype=dtype,
                             stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_dw.cpu().numpy(), dw_ref)
    if want_db:
        np.testing.assert_array_equal(d_db.cpu().numpy(), db_ref)
    else:
        assert bool(torch.isnan(d_db).all())
    return dw_ref


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_noalloc_wgrad_equals_allocating_entry_point_twice(sh, dtype):
    import torch
    big = _case(sh, torch, 1, 3, 40, 56, 64, 64, dtype)              # many slices
    small = _case(sh, torch, 2, 1, 9, 13, 5, 3, dtype, want_db=False)  # fewer, ragged; taps-in-N kernel in fp32
    again = _case(sh, torch, 1, 3, 40, 56, 64, 64, dtype)
    np.testing.assert_array_equal(big, again)
    assert np.isfinite(small).all()


def test_noalloc_rejects_what_does_not_fit(sh):
    import torch
    t = torch.zeros(16, device="cuda")
    with pytest.raises(ValueError, match="static"):
        sh.conv3x3_wgrad_noalloc(1, 8, 8, 2048, 2048, t.data_ptr(), t.data_ptr(), t.data_ptr())
    with pytest.raises(ValueError):
        sh.conv3x3_wgrad_noalloc(1, 8, 8, 4, 4, 0, t.data_ptr(), t.data_ptr())
