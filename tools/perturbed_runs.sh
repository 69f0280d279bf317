#!/bin/bash
# perturbed one-rank runs with every view's intermediate buffers kept: where does a deviating run first differ
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MASTER_ADDR=127.0.0.1 FDGS_TEST_VIEWS=4
D=$(mktemp -d)
FDGS_TEST_TAG=_a timeout -k 10 200 python tools/perturbed_worker.py $D > $D/log 2>&1 || { echo "run failed"; tail -20 $D/log; exit 1; }
for i in $(seq 1 30); do
  E=$(mktemp -d)
  FDGS_TEST_TAG=_x timeout -k 10 200 python tests/dp_gpu_worker.py $E > /dev/null 2>&1 &
  BG=$!
  FDGS_TEST_TAG=_b timeout -k 10 200 python tools/perturbed_worker.py $D > /dev/null 2>&1 || { echo "run failed"; kill $BG 2>/dev/null; exit 1; }
  wait $BG
  python - $D $i <<'PY'
import sys, torch
d, it = sys.argv[1], sys.argv[2]
a = torch.load(f"{d}/w1_r0_a.pt"); b = torch.load(f"{d}/w1_r0_b.pt")
bad = False
for k in ("grad0", "grad1", "grad2"):
    ref = a[k].float(); sc = float(ref.abs().max()) + 1e-30
    if float((b[k].float() - ref).abs().max()) / sc > 2e-5: bad = True
print(it, "OUTLIER" if bad else "ok", flush=True)
if bad:
    for k in sorted(x for x in a if x.startswith("v")):
        ra, rb = a[k], b[k]
        if ra.dtype in (torch.int32, torch.int64):
            nd = int((ra != rb).sum()); print("   ", k, "int, differing:", nd)
            if nd and k.endswith("_list"):
                pos = (ra != rb).nonzero().flatten().tolist()
                v = k[:3]; rng = a[v + "_ranges"].reshape(-1, 2); geom = a[v + "_geom"]
                print("      positions", pos, "a:", ra[pos].tolist(), "b:", rb[pos].tolist(),
                      "same multiset:", sorted(ra[pos].tolist()) == sorted(rb[pos].tolist()))
                for p_ in pos:
                    t = int(((rng[:, 0] <= p_) & (rng[:, 1] > p_)).nonzero().flatten()[0])
                    ga, gb = int(ra[p_]), int(rb[p_])
                    print(f"      pos {p_} tile {t} range {rng[t].tolist()} a: g{ga} geom {geom[ga, :4].tolist()}  b: g{gb} geom {geom[gb, :4].tolist()}")
        else:
            ra = ra.float(); diff = (rb.float() - ra).abs(); sc = float(ra.abs().max()) + 1e-30
            j = int(diff.flatten().argmax())
            if k.endswith("_dout0") and float(diff.max()) > 0:
                rows = (diff.reshape(-1, 3).max(1).values > 0).nonzero().flatten().tolist()
                print("      dout0 rows", rows, "radii", a[k[:3] + "_radii"][rows].tolist(),
                      "a", ra.reshape(-1, 3)[rows[:3]].tolist(), "b", rb.float().reshape(-1, 3)[rows[:3]].tolist())
            print("   ", k, f"max {float(diff.max()) / sc:.1e} of scale at flat {j}, n>2e-5: {int((diff > 2e-5 * sc).sum())}")
PY
done
