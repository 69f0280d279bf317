"""Which arithmetic step went wrong in a deviating HexPlane feature?  (CPU; reads a dump of tools/fwd_stale_probe.py)

`FDGS_PROBE_DUMP=<file>` makes the probe keep, for the points of deviating feature sectors, the 16 features of the
affected level as the observed run and as the undisturbed run wrote them.  The probe's scene is seeded, so this script
rebuilds it on the CPU, evaluates the six plane factors of each such point in float64 and asks, per plane: which
weights on the four bilinear corners reproduce (deviating feature / correct feature) x (correct factor)?  A least-
squares fit over the deviating channels that is exact names the plane and the weights the GPU actually used.

    python tools/feat_forensics.py gpurun_out/r03/feat_events.pt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "4d-gaussian-splatting_amd")]
PAIRS = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def main(path):
    from fdgs import scene_synth as ss
    from fdgs.deformation import DeformationFields
    events = torch.load(path)
    torch.manual_seed(ss.SEED)
    g = ss.GaussianParams(6000, device="cpu", scale_mult=3.0)
    net = ss.perturb_deformation_(DeformationFields(ss.DYNERF_NET_KWARGS))
    net.deformation_net.set_aabb(list(ss.SCENE_MAX), list(ss.SCENE_MIN))
    grid = net.deformation_net.grid
    aabb = grid.aabb.detach().numpy().astype(np.float32)
    xyz = g._xyz.detach().numpy().astype(np.float32)

    def factors(p, level, t):
        co = np.zeros(4, np.float32)
        for k in range(3):
            co[k] = (xyz[p, k] - aabb[0, k]) * (np.float32(2.0) / np.float32(aabb[1, k] - aabb[0, k])) - np.float32(1.0)
        co[3] = np.float32(t)
        out = []
        for ci, (a, b) in enumerate(PAIRS):
            pl = grid.grids[level][ci].detach().numpy()[0].astype(np.float64)       # [16, H, W]

            def axis(c, size):
                x = min(max(np.float32((c + 1) * 0.5) * np.float32(size - 1), 0.0), size - 1)
                x0 = int(np.floor(x))
                return x0, min(x0 + 1, size - 1) - x0, float(x - x0)
            x0, dx, fx = axis(co[a], pl.shape[2])
            y0, dy, fy = axis(co[b], pl.shape[1])
            tex = np.stack([pl[:, y0, x0], pl[:, y0, x0 + dx], pl[:, y0 + dy, x0], pl[:, y0 + dy, x0 + dx]], 1)
            w = np.array([(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy])
            out.append((tex @ w, tex, w))
        return out

    np.set_printoptions(precision=5, suppress=True)
    print(len(events), "deviating (point, level) records")
    for e in events:
        got, ref = np.array(e["got"]), np.array(e["ref"])
        fac = factors(e["point"], e["level"], e["time"])
        cpu = np.prod([f[0] for f in fac], axis=0)
        ratio = got / ref
        wrong = np.nonzero(np.abs(ratio - 1) > 1e-6)[0]
        best = None
        for k in range(6):
            want = ratio[wrong] * fac[k][0][wrong]
            alpha = np.linalg.lstsq(fac[k][1][wrong], want, rcond=None)[0]
            fit = float(np.max(np.abs(fac[k][1][wrong] @ alpha - want) / np.abs(want)))
            if best is None or fit < best[1]:
                best = (k, fit, alpha)
        k, fit, alpha = best
        print(f"point {e['point']:5d} level {e['level']} view {e['view']:2d}: cpu vs undisturbed {np.max(np.abs(cpu - ref) / np.abs(ref)):.1e};"
              f" wrong channels {wrong.tolist()}; plane {k} {PAIRS[k]} fit {fit:.1e}: weights used {alpha} true {fac[k][2]}")


if __name__ == "__main__":
    main(sys.argv[1])
