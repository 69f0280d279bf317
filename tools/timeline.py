#!/usr/bin/env python
"""Per-queue timeline of one training step from a rocprofv3 --kernel-trace CSV: the kernels between two
consecutive adam_kernel launches (by default the last complete step of the trace; a third argument -k picks the
k-th step from the end), with start offset, duration and the
hardware queue each ran on, so that it shows which of two concurrently running streams sets the step's length."""
import csv
import sys


def main(path, out=None, which=-2):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if r["name"].startswith("adam_kernel")]
    if len(adam) < 3:
        raise SystemExit("need at least three steps in the trace")
    lo, hi = adam[which - 1] + 1, adam[which] + 1
    step = rows[lo:hi]
    t0 = step[0]["s"]
    qkey = "Queue_Id" if "Queue_Id" in step[0] else "Stream_Id"
    queues = sorted({r[qkey] for r in step})
    lines = [f"step of {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us from first start to last end; "
             f"queues: {', '.join(queues)}"]
    busy = {q: 0 for q in queues}
    for r in step:
        busy[r[qkey]] += r["e"] - r["s"]
        lines.append(f"{(r['s'] - t0) / 1e3:9.1f} us  +{(r['e'] - r['s']) / 1e3:8.1f} us  q{queues.index(r[qkey])}  {r['name'][:70]}")
    for q in queues:
        lines.append(f"queue {q}: {busy[q] / 1e3:.1f} us of kernel time")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else -2)
