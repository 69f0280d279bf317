#!/usr/bin/env python
"""LDS traffic per kernel from one rocprofv3 --pmc pass (SQ_INSTS_LDS_LOAD_BANDWIDTH, _STORE_BANDWIDTH,
_ATOMIC_BANDWIDTH: units of 64 bytes per lane-instruction; SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT: LDS cycles summed
over the CUs; GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs), joined with the average launch durations of a
--kernel-trace --stats pass of the same command.  Peak: 128 bytes per cycle per CU x 256 CUs x 2.4 GHz = 78.6 TB/s."""
import collections
import csv
import json
import sys

CUS, XCDS, CLOCK_GHZ, BYTES_PER_CLK_CU = 256, 8, 2.4, 128.0
PEAK_TBS = CUS * CLOCK_GHZ * BYTES_PER_CLK_CU / 1e3


def main(pmc_csv, stats_csv, out_json):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc_csv)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(stats_csv))}
    out = {}
    for k, c in agg.items():
        if k.startswith("void at::") or k.startswith("__amd") or k not in dur:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        nbytes = 64.0 * (m.get("SQ_INSTS_LDS_LOAD_BANDWIDTH", 0) + m.get("SQ_INSTS_LDS_STORE_BANDWIDTH", 0)
                         + m.get("SQ_INSTS_LDS_ATOMIC_BANDWIDTH", 0))
        if nbytes == 0:
            continue
        cu_cycles = m.get("GRBM_GUI_ACTIVE", 0.0) / XCDS * CUS
        idx = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
        out[k] = {"avg_launch_us": dur[k] / 1e3, "lds_bytes_per_launch": nbytes,
                  "lds_tb_per_s": nbytes / dur[k] / 1e3, "lds_frac_of_peak": nbytes / dur[k] / 1e3 / PEAK_TBS,
                  "lds_peak_tb_per_s": PEAK_TBS,
                  "atomic_share_of_bytes": 64.0 * m.get("SQ_INSTS_LDS_ATOMIC_BANDWIDTH", 0) / nbytes,
                  "lds_busy_frac_of_cu_cycles": idx / cu_cycles if cu_cycles else None,
                  "bank_conflict_frac_of_lds_cycles": m.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else None,
                  "lds_insts_per_launch": m.get("SQ_INSTS_LDS", 0.0)}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["lds_bytes_per_launch"])[:12]:
        print(f"{k:36s} {v['avg_launch_us']:8.1f} us  LDS {v['lds_bytes_per_launch']/1e9:6.2f} GB  "
              f"{v['lds_tb_per_s']:5.1f} TB/s = {100*v['lds_frac_of_peak']:4.1f}% of {PEAK_TBS:.1f}  "
              f"atomics {100*v['atomic_share_of_bytes']:4.1f}% of bytes  LDS busy {100*(v['lds_busy_frac_of_cu_cycles'] or 0):4.1f}%  "
              f"bank conflicts {100*(v['bank_conflict_frac_of_lds_cycles'] or 0):4.1f}% of LDS cycles")


if __name__ == "__main__":
    main(*sys.argv[1:4])
