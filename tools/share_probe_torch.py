"""Control experiment for tools/fwd_stale_probe.py: do kernels that are NOT this library's also return different
bits when a second process uses the GPU?

The observed process runs a fixed list of deterministic PyTorch operations (no atomics, fixed shapes: elementwise
chains, a bilinear grid_sample gather, a matrix product, a row softmax) in a loop and compares every result bit by bit,
on the device, with the result of the first, undisturbed evaluation.  The disturbing process is the one of
tools/fwd_stale_probe.py (`python tools/fwd_stale_probe.py load <dir>`), or a second copy of this loop.

    python tools/share_probe_torch.py main <dir> [iterations]
    python tools/share_probe_torch.py load <dir>
"""
import os
import sys
import time

import torch


def make_ops(dev):
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(6000, 48, generator=g).to(dev)
    big = torch.randn(1 << 20, generator=g).to(dev)
    plane = torch.rand(1, 16, 150, 128, generator=g).to(dev)
    coords = (torch.rand(1, 1, 6000, 2, generator=g) * 2 - 1).to(dev)
    a = torch.randn(1024, 512, generator=g).to(dev)
    b = torch.randn(512, 768, generator=g).to(dev)
    idx = torch.randint(0, 6000, (20000,), generator=g).to(dev)

    def chain_small():
        return torch.exp(-x * x) * torch.sin(x) + torch.sqrt(x.abs() + 1.0) / (1.0 + x * x)

    def chain_big():
        return torch.exp(-big * big) * torch.sin(big) + torch.sqrt(big.abs() + 1.0)

    def gather():
        return torch.nn.functional.grid_sample(plane, coords, mode="bilinear", padding_mode="border",
                                               align_corners=True)

    def matmul():
        return a @ b

    def softmax():
        return torch.softmax(x, dim=1)

    def index():
        return x[idx] * 2.0 + 1.0

    return [("chain_small", chain_small), ("chain_big", chain_big), ("grid_sample", gather), ("matmul", matmul),
            ("softmax", softmax), ("index", index)]


def main(flag_dir, iterations):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ops = make_ops(dev)
    refs = [f().clone() for _, f in ops]
    for _ in range(20):         # the reference reproduces itself while the GPU is ours alone
        for (name, f), r in zip(ops, refs):
            if not torch.equal(f().view(torch.int32), r.view(torch.int32)):
                print("UNDISTURBED runs differ:", name, flush=True)
    torch.cuda.synchronize()
    print("reference reproduces itself", flush=True)
    open(os.path.join(flag_dir, "ready"), "w").close()
    counts = torch.zeros(len(ops), dtype=torch.int64, device=dev)
    first_bad = {}
    t0 = time.time()
    for it in range(iterations):
        outs = [f() for _, f in ops]
        bad = torch.stack([(o.view(torch.int32) != r.view(torch.int32)).any() for o, r in zip(outs, refs)])
        counts += bad
        if it % 50 == 49:
            flags = bad.cpu()
            if bool(flags.any()):
                for k in flags.nonzero().flatten().tolist():
                    if ops[k][0] not in first_bad:
                        d = (outs[k].view(torch.int32) != refs[k].view(torch.int32)).flatten().nonzero().flatten()
                        first_bad[ops[k][0]] = (it, d[:64].tolist(), int(d.numel()))
        if it % 500 == 0:
            print(f"iteration {it} t={time.time() - t0:.0f}s deviating so far {counts.tolist()}", flush=True)
    torch.cuda.synchronize()
    open(os.path.join(flag_dir, "done"), "w").close()
    print("iterations", iterations, "deviating evaluations per op:",
          {n: int(c) for (n, _), c in zip(ops, counts.tolist())}, flush=True)
    for name, (it, where, total) in first_bad.items():
        print(f"   {name}: caught at iteration {it}: {total} elements differ, flat indices {where}", flush=True)


def load(flag_dir):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ops = make_ops(dev)
    # SHARE_PROBE_STREAMS=k: the operations go round k streams (k hardware queues kept busy by this process)
    streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("SHARE_PROBE_STREAMS", "0")))]
    i = 0
    while not os.path.exists(os.path.join(flag_dir, "done")):
        for k, (_, f) in enumerate(ops):
            if streams:
                with torch.cuda.stream(streams[(i + k) % len(streams)]):
                    f()
            else:
                f()
        i += 1
        if i % 20 == 0:
            torch.cuda.synchronize()
    print("load: iterations", i, flush=True)


if __name__ == "__main__":
    role, d = sys.argv[1], sys.argv[2]
    if role == "load":
        while not os.path.exists(os.path.join(d, "ready")):
            time.sleep(0.2)
        load(d)
    else:
        main(d, int(sys.argv[3]) if len(sys.argv) > 3 else 3000)
