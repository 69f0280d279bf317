"""Where does the deformation forward's rare deviation under GPU sharing come from?

The forward chain of a view (plane copies / weight pieces -> HexPlane features -> MLP) has no atomics: with frozen
parameters every buffer it writes is a bit-exact function of the view.  This probe runs the training step of
tests/dp_gpu_worker.py (TrainStep on 6000 Gaussians, optimizer update replaced by a no-op) over a fixed cycle of
views, keeps every forward buffer of the first, undisturbed cycles as the reference, and then repeats the cycle
while a second process uses the same GPU, comparing every view's buffers bit by bit on the device:

    out0..out4   the five outputs of fdgs_deform_fwd
    fws          its workspace: plane copies | weight pieces | biases | feature pieces (feat_split)
    yfeat, act, mask, yhid   what it saves for the backward
    geom, radii  the rasterizer's per-Gaussian records (consumers of out0..out4)

For a deviating view it reports which buffers differ, in which rows / 64-byte sectors, and whether the deviating
values are those of ANOTHER view of the cycle (a stale read or a late write) or of no view (a wrong computation).

    python tools/fwd_stale_probe.py main  <dir> [cycles]      # the observed process (tools/share_probe.sh runs both)
    python tools/fwd_stale_probe.py load  <dir>               # the disturbing process (loops until <dir>/done exists)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "4d-gaussian-splatting_amd")]

VIEWS_PER_STEP = int(os.environ.get("FDGS_TEST_VIEWS", "4"))
STEPS_PER_CYCLE = 3
N_POINTS = int(os.environ.get("FDGS_PROBE_POINTS", "6000"))


def build(dev):
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    from fdgs.renderer import render
    from fdgs.train_step import TrainStep, DeformHolder
    torch.manual_seed(ss.SEED)
    n, w, h = N_POINTS, 160, 120
    g = ss.GaussianParams(n, device=dev, scale_mult=3.0)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS)).to(dev)
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    deforms = DeformHolder(net)
    cams = ss.make_cameras(6, w, h, device=dev)
    bg = torch.ones(3, device=dev)
    g2 = ss.GaussianParams(n, seed=ss.SEED + 100, device=dev, requires_grad=False, scale_mult=3.0)
    with torch.no_grad():
        for cam in cams:
            cam.original_image = render(cam, g2, deforms, bg, stage="coarse")["render"].clamp(0, 1).clone()
    step = TrainStep(g, deforms, bg)
    step.optimizer.step = lambda: None
    return step, cams


def cycle_views(cams, i):
    return [cams[(2 * i + v) % len(cams)] for v in range(VIEWS_PER_STEP)]


def load(flag_dir):
    """The disturbing process: by default the same training step in a loop, restarting nothing.
    FDGS_PROBE_LOAD selects a part of it: render (no-grad fine render: deformation forward + rasterizer forward),
    coarse (rasterizer forward + backward + loss, no deformation), deform (deformation forward + backward through
    autograd), deform_fwd (deformation training forward only)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    step, cams = build(dev)
    mode = os.environ.get("FDGS_PROBE_LOAD", "step")
    from fdgs.renderer import render
    g, deforms = step.g, step.deforms
    net = deforms.deformation_fields
    i = 0
    while not os.path.exists(os.path.join(flag_dir, "done")):
        views = cycle_views(cams, i % STEPS_PER_CYCLE)
        if mode == "step":
            step(views)
        elif mode == "coarse":
            step(views, stage="coarse")
        elif mode == "render":
            with torch.no_grad():
                for cam in views:
                    render(cam, g, deforms, step.bg, stage="fine")
        else:
            for cam in views:
                outs = net(g._xyz, g._scaling, g._rotation, g._opacity, g.get_features, float(cam.time))
                if mode == "deform":
                    sum(o.sum() for o in outs).backward()
                    for prm in list(net.parameters()) + [g._xyz, g._scaling, g._rotation, g._opacity, g._features_dc,
                                                         g._features_rest]:
                        prm.grad = None
        i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("load:", mode, "steps", i, flush=True)


def main(flag_dir, cycles):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    step, cams = build(dev)
    p = step.pipeline
    names = ("out0", "out1", "out2", "out3", "out4", "fws", "yfeat", "act", "mask", "yhid", "geom", "radii", "touched",
             "offsets", "ranges", "color", "finalT", "ncontrib", "gradimage", "arena")
    # "arena" = every gradient of the view (deformation field, positions, view-space, scaling, rotation, opacity, SH).
    # The backward kernels accumulate with atomics, so it is compared to a tolerance: 2e-5 of the largest magnitude of
    # each region (run-to-run noise of the atomics' order is about 2e-7), not bit by bit.
    TOL = {"arena": 2e-5}

    def buffers():
        return (p.d_out[0], p.d_out[1], p.d_out[2], p.d_out[3], p.d_out[4], p._deform_ws[0], p.yfeat, p.act, p.mask,
                p.yhid, p.geom, p.radii, p.tiles_touched, p.point_offsets, p.ranges, p.color, p.final_T, p.n_contrib,
                p.grad_image, p.arenas[state["set"]])

    def as_int(t):
        return t.view(torch.int32) if t.dtype in (torch.float32,) else t

    def differs(name, b, r):
        if name not in TOL:
            return (as_int(b) != as_int(r)).any()
        m = p.arena_marks
        bad = torch.zeros((), dtype=torch.bool, device=b.device)
        marks = sorted(m.values())
        for lo, hi in zip(marks, marks[1:]):
            if hi > lo:
                scale = r[lo:hi].abs().max()
                bad = bad | ((b[lo:hi] - r[lo:hi]).abs().max() > TOL[name] * scale + 1e-30)
        return bad

    state = {"view": 0, "mode": "ref", "ref": [], "flags": [], "kept": [], "set": 0}
    inner = p.run

    def run(cam, *a, **k):
        out = inner(cam, *a, **k)
        v = state["view"]
        state["view"] += 1
        state["set"] = k.get("grad_set", 0)
        bufs = buffers()
        if state["mode"] == "ref":
            state["ref"].append([b.clone() for b in bufs])
        elif state["mode"] == "ref2":
            ref = state["ref"][v]
            state["flags"].append(torch.stack([differs(nm, b, r) for nm, b, r in zip(names, bufs, ref)]))
        else:
            ref = state["ref"][v]
            flag = torch.stack([differs(nm, b, r) for nm, b, r in zip(names, bufs, ref)])
            state["flags"].append(flag)
            # keep the buffers of this view until the cycle's flags have been read (cheap at this size)
            state["kept"].append([b.clone() for b in bufs])
        return out
    p.run = run

    # FDGS_PROBE_INPROC=k: no second process; before every step of the probed cycles, k deformation forward +
    # backward passes of an independent copy of the scene are enqueued on a second stream of THIS process, so that
    # their kernels share the compute units with the observed step's
    inproc = int(os.environ.get("FDGS_PROBE_INPROC", "0"))
    if inproc:
        step_b, cams_b = build(dev)
        side = torch.cuda.Stream()
        net_b, g_b = step_b.deforms.deformation_fields, step_b.g
        prm_b = list(net_b.parameters()) + [g_b._xyz, g_b._scaling, g_b._rotation, g_b._opacity, g_b._features_dc,
                                            g_b._features_rest]

    def one_cycle():
        state["view"] = 0
        state["flags"], state["kept"] = [], []
        for i in range(STEPS_PER_CYCLE):
            if inproc and state["mode"] == "probe":
                with torch.cuda.stream(side):
                    for k in range(inproc):
                        outs = net_b(g_b._xyz, g_b._scaling, g_b._rotation, g_b._opacity, g_b.get_features,
                                     float(cams_b[k % len(cams_b)].time))
                        sum(o.sum() for o in outs).backward()
                        for prm in prm_b:
                            prm.grad = None
            step(cycle_views(cams, i))
        torch.cuda.synchronize()

    one_cycle()                                   # reference
    state["mode"] = "ref2"
    for _ in range(3):                            # the reference reproduces itself while the GPU is ours alone
        one_cycle()
        bad = torch.stack(state["flags"]).cpu()
        if bool(bad.any()):
            print("UNDISTURBED cycles differ:", bad.nonzero().tolist(), flush=True)
    print("reference cycles agree; views per cycle:", len(state["ref"]), flush=True)
    times = [float(c.time) for i in range(STEPS_PER_CYCLE) for c in cycle_views(cams, i)]
    print("view times:", [round(t, 4) for t in times], flush=True)
    n = N_POINTS
    from fdgs._lib import lib
    import ctypes as C
    from fdgs import deformation as D
    P = D._fill_params(step.deforms.deformation_fields.deformation_net)
    fws_bytes = lib.fdgs_deform_fwd_workspace_bytes(C.byref(P), n)
    feat_bytes = ((n + 31) // 32) * 2 * 3 * 64 * 16
    feat_off = fws_bytes - feat_bytes
    print("fws bytes", fws_bytes, "feat_split at byte", feat_off, "(mod 256 =", feat_off % 256, ") base mod 256 =",
          p._deform_ws[0].data_ptr() % 256, flush=True)

    order = sorted((off, nm) for nm, off in p.arena_marks.items())
    state["regions"] = [(lo, hi, nm) for (lo, nm), (hi, _) in zip(order, order[1:])]
    open(os.path.join(flag_dir, "ready"), "w").close()
    state["mode"] = "probe"
    t0 = time.time()
    events = 0
    for cyc in range(cycles):
        one_cycle()
        flags = torch.stack(state["flags"]).cpu()
        if bool(flags.any()):
            for v in flags.any(1).nonzero().flatten().tolist():
                events += 1
                report(cyc, v, flags[v], state, names, n, feat_off, feat_bytes, times)
        if cyc % 20 == 0:
            print(f"cycle {cyc} t={time.time() - t0:.0f}s events={events}", flush=True)
    open(os.path.join(flag_dir, "done"), "w").close()
    print("cycles", cycles, "views", cycles * len(state["ref"]), "deviating views", events, flush=True)
    if FEAT_EVENTS and os.environ.get("FDGS_PROBE_DUMP"):
        torch.save(FEAT_EVENTS, os.environ["FDGS_PROBE_DUMP"])


FEAT_EVENTS = []


def decode_features(fws, feat_off, point, level):
    """The 16 features of (point, level) from the split pieces in the workspace: exact fp32 = h + m + l."""
    G, c = point // 32, point % 32
    lane = c + 32 * level
    words = fws.view(torch.uint8)[feat_off:].view(torch.int32)
    out = []
    for f in range(16):
        j, e = f // 8, f % 8
        val = 0.0
        for piece in range(3):
            w = int(words[(((G * 2 + j) * 3 + piece) * 64 + lane) * 4 + e // 2]) & 0xffffffff
            half = (w >> 16) if (e % 2) else (w & 0xffff)
            val += torch.tensor([half << 16], dtype=torch.int64).to(torch.int32).view(torch.float32).double().item()
        out.append(val)
    return out


def report(cyc, v, flag, state, names, n, feat_off, feat_bytes, times):
    got, refs = state["kept"][v], state["ref"]
    ref = refs[v]
    differing = [names[i] for i in flag.nonzero().flatten().tolist()]
    print(f"== cycle {cyc} view {v} (time {times[v]:.4f}, previous view's time {times[v - 1]:.4f}): differing {differing}",
          flush=True)
    for i, name in enumerate(names):
        if not bool(flag[i]):
            continue
        g, r = got[i], ref[i]
        if name == "arena":
            line = "   arena (largest deviation / largest magnitude per region):"
            for lo, hi, nm in state["regions"]:
                if hi > lo:
                    line += f" {nm} {float((g[lo:hi] - r[lo:hi]).abs().max() / (r[lo:hi].abs().max() + 1e-30)):.1e}"
            print(line, flush=True)
            continue
        if name == "fws":
            gb, rb = g.view(torch.uint8), r.view(torch.uint8)
            idx = (gb != rb).nonzero().flatten()
            sect = torch.unique(idx // 64).tolist()
            inside = [s for s in sect if s * 64 >= feat_off]
            print(f"   fws: {len(sect)} sectors of 64 bytes differ, {len(inside)} of them inside feat_split", flush=True)
            for s in sect[:40]:
                off = s * 64
                line = f"      byte {off} (mod 256 = {off % 256})"
                if off >= feat_off:
                    rel = off - feat_off
                    blk, lane0 = rel // 1024, (rel % 1024) // 16
                    G, j, piece = blk // 6, (blk // 3) % 2, blk % 3
                    c, h = lane0 % 32, lane0 // 32
                    line += f" group {G} chunk {j} piece {piece} lanes {lane0}..{lane0 + 3}: points {32 * G + c}..{32 * G + c + 3} level {h}"
                    same = [u for u in range(len(refs)) if u != v and
                            torch.equal(refs[u][i].view(torch.uint8)[off:off + 64], gb[off:off + 64])]
                    line += f"; equals the reference of views {same}"
                    if len(FEAT_EVENTS) < 64 and piece == 0 and j == 0:
                        for pt in range(32 * G + c, 32 * G + c + 4):
                            FEAT_EVENTS.append(dict(view=v, point=pt, level=h, time=times[v],
                                                    got=decode_features(g, feat_off, pt, h),
                                                    ref=decode_features(r, feat_off, pt, h)))
                print(line, flush=True)
            continue
        rows = g.shape[0]
        g2, r2 = g.reshape(rows, -1), r.reshape(rows, -1)
        gi = g2.view(torch.int32) if g2.dtype == torch.float32 else g2
        ri = r2.view(torch.int32) if r2.dtype == torch.float32 else r2
        bad_rows = (gi != ri).any(1).nonzero().flatten().tolist()
        line = f"   {name}: {len(bad_rows)} rows differ: {bad_rows[:48]}"
        if name.startswith("out") or name in ("geom", "radii", "mask"):
            br = torch.tensor(bad_rows, device=g.device)
            same = [u for u in range(len(refs)) if u != v and
                    torch.equal((refs[u][i].reshape(rows, -1))[br], g2[br])]
            line += f"; these rows equal the reference of views {same}"
            if name == "out0":
                rel = ((g2[br] - r2[br]).abs().max() / r2.abs().max()).item()
                line += f"; max deviation {rel:.2e} of scale"
            if name == "geom":
                cols = (gi[br] != ri[br]).sum(0).tolist()
                line += f"; rows differing per column {cols}; first row got {g2[br[0]].tolist()} ref {r2[br[0]].tolist()}"
        print(line, flush=True)


if __name__ == "__main__":
    role, d = sys.argv[1], sys.argv[2]
    if role == "load":
        while not os.path.exists(os.path.join(d, "ready")):
            time.sleep(0.2)
        load(d)
    else:
        main(d, int(sys.argv[3]) if len(sys.argv) > 3 else 200)
