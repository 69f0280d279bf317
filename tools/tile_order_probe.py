"""Times the two blend kernels at 300k Gaussians / 1352x1014 under different tile orders (which tile workgroup i
takes): the library's own (longest lists first), image order, and orders that keep spatial neighbourhoods on one XCD
(workgroup i runs on XCD i mod 8: positions r, r + 8, ... hold the tiles of the super-tiles with
(sx + 3 sy) mod 8 == r) with different orders inside an XCD's share.

Result on MI355X: with settled clocks the XCD-aware orders are no faster than the library's; a share in random or
image order is slower (long lists must start first).  Every variant runs 100 warm-up launches and the first ones are
measured again at the end: without that the variants measured first look 15-20 % slower than they are.

    python tools/tile_order_probe.py [super-tile edge, in tiles: 0 = contiguous bands] ...
"""
import math
import sys

import torch

sys.path.insert(0, "4d-gaussian-splatting_amd")
from fdgs import _lib, rasterizer as fr, scene_synth as ss      # noqa: E402

W, H = 1352, 1014
dev = torch.device("cuda:0")


def region_sequence(gx, gy, edge):
    """Tiles in the order that is cut into eight equal parts, one per XCD."""
    t = torch.arange(gx * gy)
    ty, tx = t // gx, t % gx
    if edge == 0:
        return t
    sx, sy = tx // edge, ty // edge
    nsx = (gx + edge - 1) // edge
    xcd = (sx + 3 * sy) % 8
    key = (xcd * (nsx * ((gy + edge - 1) // edge)) + sy * nsx + sx) * (edge * edge) + (ty % edge) * edge + tx % edge
    return torch.argsort(key, stable=True)


def length_class(lengths, per_octave):
    e = torch.floor(torch.log2(lengths.clamp(min=1).double()))
    return torch.floor((lengths.double() / 2 ** e - 1.0) * per_octave) + e * per_octave


def xcd_order(lengths, gx, gy, edge, within="exact"):
    T = gx * gy
    seq = region_sequence(gx, gy, edge)
    order = torch.empty(T, dtype=torch.int64)
    start = 0
    gen = torch.Generator().manual_seed(1)
    for r in range(8):
        c = T // 8 + (1 if r < T % 8 else 0)
        mine = seq[start:start + c]
        start += c
        if within == "exact":
            mine = mine[torch.argsort(lengths[mine], descending=True, stable=True)]
        elif within == "random":
            mine = mine[torch.randperm(c, generator=gen)]
        elif within == "class8 stable":
            mine = mine[torch.argsort(length_class(lengths[mine], 8), descending=True, stable=True)]
        elif within == "class8 random":
            mine = mine[torch.randperm(c, generator=gen)]
            mine = mine[torch.argsort(length_class(lengths[mine], 8), descending=True, stable=True)]
        elif within == "class2 stable":
            mine = mine[torch.argsort(length_class(lengths[mine], 2), descending=True, stable=True)]
        elif within == "sequence":
            pass
        order[torch.arange(c) * 8 + r] = mine
    assert torch.equal(torch.sort(order).values, torch.arange(T))
    return order.to(torch.int32)


def timed(fn, n=30):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    edges = [int(v) for v in sys.argv[1:]] or [0, 4, 8]
    cam = ss.make_cameras(21, W, H)[7]
    g = ss.GaussianParams(300000, scale_mult=1.0, requires_grad=False)
    S = fr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                         torch.ones(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                         cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
    ins = [g._xyz.to(dev), g.get_features.contiguous().to(dev), torch.sigmoid(g._opacity).reshape(-1).to(dev),
           torch.exp(g._scaling).to(dev), torch.nn.functional.normalize(g._rotation).to(dev)]
    st = fr.rasterize_forward(*ins, S, keep_masks=True)
    torch.cuda.synchronize()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lengths = (st.ranges[:, 1] - st.ranges[:, 0]).cpu().long()
    print("entries", st.num_rendered, "tiles", gx * gy)
    ptr = _lib.ptr
    vm, pm, cp, bg = st.cam
    gout = torch.rand(3, H, W, device=dev)
    dgeom = torch.zeros(300000, _lib.GEOM_FLOATS, device=dev)
    orders = {"library": st.tile_order.clone(), "image order": torch.arange(gx * gy, dtype=torch.int32, device=dev)}
    for e in edges:
        orders["xcd, super-tile %d" % e] = xcd_order(lengths, gx, gy, e).to(dev)
    perm = torch.randperm(gx * gy, generator=torch.Generator().manual_seed(2))
    orders["global longest first"] = perm[torch.argsort(length_class(lengths[perm], 16), descending=True,
                                                         stable=True)].to(torch.int32).to(dev)
    for within in ("random", "class8 stable", "class8 random", "class2 stable", "sequence"):
        orders["xcd 4, " + within] = xcd_order(lengths, gx, gy, 4, within).to(dev)
    for name, order in list(orders.items()) + list(orders.items())[:4]:     # the first two again: clocks have settled
        c = _lib.BlendFwdArgs(W, H, ptr(st.ranges), ptr(st.vals_sorted), ptr(st.geom), ptr(bg), ptr(st.color),
                              ptr(st.depth), ptr(st.final_T), ptr(st.n_contrib), ptr(order), ptr(st.reach_masks))
        b = _lib.BlendBwdArgs(W, H, ptr(st.ranges), ptr(st.vals_sorted), ptr(st.geom), ptr(bg), ptr(st.final_T),
                              ptr(st.n_contrib), ptr(gout), ptr(dgeom), ptr(order), ptr(st.reach_masks))
        o = order.cpu().long()
        home = ((o % gx) // 4 + 3 * ((o // gx) // 4)) % 8
        away = sum(int((home[r::8] != r).sum()) for r in range(8))
        unsorted = sum(int((lengths[o[r::8]][1:] > lengths[o[r::8]][:-1]).sum()) for r in range(8))
        worst = max(float((lengths[o[r::8]][1:].float() / lengths[o[r::8]][:-1].float().clamp(min=1)).max()) for r in range(8))
        head = [int(v) for v in lengths[o[0::8]][:6]] + [int(v) for v in lengths[o[0::8]][-3:]]
        print("   largest step up %.3f, share 0 lengths %s" % (worst, head))
        print("   tiles outside their XCD's super-tiles: %d, ascending steps within the shares: %d" % (away, unsorted))
        tf = timed(lambda: _lib.call("fdgs_raster_blend_fwd", c))
        tb = timed(lambda: _lib.call("fdgs_raster_blend_bwd", b))
        print("%-26s blend_fwd %.4f ms   blend_bwd %.4f ms" % (name, tf, tb), flush=True)


if __name__ == "__main__":
    main()
