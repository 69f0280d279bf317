"""GPU: fdgs_l1_loss_fwd_static, the entry point that allocates nothing (its partial sums live in a scratch array
built into the library), against fdgs_l1_loss_fwd through the C ABI: bit-identical loss and PSNR on two calls in a
row, the second with a smaller input (the scratch array then still holds the first call's partials), then the
size limit."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(a, b, scale, with_psnr):
    from fdgs import _lib
    from fdgs._lib import lib, ptr
    dev = a.device
    n = a.numel()
    sp = torch.cuda.current_stream().cuda_stream
    out = {}
    for name in ("fdgs_l1_loss_fwd", "fdgs_l1_loss_fwd_static"):
        loss = torch.full((1,), float("nan"), device=dev)
        psnr = torch.full((1,), float("nan"), device=dev) if with_psnr else None
        partial = None
        if name == "fdgs_l1_loss_fwd":
            partial = torch.empty(2 * lib.fdgs_l1_loss_partials(n), device=dev)
        args = _lib.L1FwdArgs(n, ptr(a), ptr(b), scale, ptr(loss), ptr(partial), ptr(psnr))
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        rc = getattr(lib, name)(C.byref(args), sp)
        assert rc == 0, lib.fdgs_strerror(rc)
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
        torch.cuda.synchronize()
        out[name] = (loss.clone(), None if psnr is None else psnr.clone())
    return out["fdgs_l1_loss_fwd"], out["fdgs_l1_loss_fwd_static"]


def test_static_scratch_entry_point_matches_the_allocating_one():
    from fdgs import _lib
    from fdgs._lib import lib, ptr
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(31)
    # first call: a full 1352 x 1014 RGB image (1004 partials); second: a smaller, odd-sized one (tail path)
    for shape, scale, with_psnr in (((3, 1014, 1352), 1.0, True), ((3, 37, 53), 0.125, True), ((1, 5), 1.0, False)):
        a = torch.rand(shape, generator=gen).to(dev)
        b = torch.rand(shape, generator=gen).to(dev)
        (l0, p0), (l1, p1) = _both(a, b, scale, with_psnr)
        assert torch.equal(l0, l1) and torch.isfinite(l1).all()
        ref = (a - b).abs().mean() * scale
        assert float((l1[0] - ref).abs()) <= 1e-6 * float(ref)
        if with_psnr:
            assert torch.equal(p0, p1) and torch.isfinite(p1).all()
    # beyond the built-in limit: refused with its own status, nothing launched
    loss = torch.zeros(1, device=dev)
    args = _lib.L1FwdArgs(_lib.L1_STATIC_MAX_N + 1, ptr(a), ptr(b), 1.0, ptr(loss), None, None)
    rc = lib.fdgs_l1_loss_fwd_static(C.byref(args), torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.ETOOLARGE and b"limit" in lib.fdgs_strerror(rc)
    args = _lib.L1FwdArgs(0, ptr(a), ptr(b), 1.0, ptr(loss), None, None)
    assert lib.fdgs_l1_loss_fwd_static(C.byref(args), torch.cuda.current_stream().cuda_stream) == -1
