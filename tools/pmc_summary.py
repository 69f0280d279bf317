#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter CSVs (one pass per counter) into per-kernel
averages and an HBM-traffic estimate per launch.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B
per 128-B request of a wide (16 B per lane) read, so it is doubled before it is
compared with a byte count; WRITE_SIZE is exact for 16-B-per-lane stores and for
float atomics.  traffic = 2*FETCH + WRITE.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main(fetch_csv, write_csv, out_json):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if k.startswith("void at::") or k.startswith("__amd"):
            continue
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        out[k] = {"fetch_size_kib_avg": fk, "write_size_kib_avg": wk, "launches": f.get(k, (0, 0))[1],
                  "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]):
        print(f"{k:40s} fetch {v['fetch_size_kib_avg']/1024:9.1f} MiB  write {v['write_size_kib_avg']/1024:9.1f} MiB  "
              f"traffic {v['traffic_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
