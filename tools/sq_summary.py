#!/usr/bin/env python
"""Per-kernel averages of one rocprofv3 --pmc pass over SQ counters, joined with the average launch durations of a
--kernel-trace --stats pass of the same command: vector instructions per launch, their issue rate against the
chip's peak, and how the waves' cycles divide into issuing / stalled at issue / parked (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES;
these four count quad-cycles).

Issue peak: 1024 SIMDs x 2.4 GHz / the cycles one wave64 vector instruction occupies its SIMD.  The nominal figure
is 4 cycles (614 G/s); a microbenchmark of independent v_fma_f32 with two or more waves per SIMD measured 2.92
(842 G/s; v_pk_fma_f32 5.0, v_exp_f32 8, a DPP v_mov 4.1-4.6), so fractions are quoted against 842 G/s and the
nominal figure is kept beside it."""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
CLOCK_GHZ = 2.4
VALU_CYCLES_MEASURED = 2.92
VALU_CYCLES_NOMINAL = 4.0


def main(pmc_csv, stats_csv, out_json):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc_csv)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        dur[r["Name"].split("(")[0]] = float(r["AverageNs"])
    out = {}
    for k, c in agg.items():
        if k.startswith("void at::") or k.startswith("__amd") or k not in dur:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        ns = dur[k]
        valu = m.get("SQ_INSTS_VALU", 0.0)
        wc = max(m.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        out[k] = {"launches": len(next(iter(c.values()))), "avg_launch_us": ns / 1e3,
                  "valu_insts_per_launch": valu, "salu_insts_per_launch": m.get("SQ_INSTS_SALU", 0.0),
                  "lds_insts_per_launch": m.get("SQ_INSTS_LDS", 0.0),
                  "valu_ginst_per_s": valu / ns,
                  "valu_issue_peak_ginst_per_s": SIMDS * CLOCK_GHZ / VALU_CYCLES_MEASURED,
                  "valu_issue_peak_nominal_ginst_per_s": SIMDS * CLOCK_GHZ / VALU_CYCLES_NOMINAL,
                  "valu_issue_frac": valu / ns / (SIMDS * CLOCK_GHZ / VALU_CYCLES_MEASURED),
                  "mfma_insts_per_launch": m.get("SQ_INSTS_MFMA", 0.0),
                  "mfma_busy_cycles_per_launch": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                  "wave_cycles_issuing": m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                  "wave_cycles_issue_stalled": m.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                  "wave_cycles_parked": m.get("SQ_WAIT_ANY", 0.0) / wc,
                  "valu_active_of_wave_cycles": m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_launch_us"])[:14]:
        print(f"{k:36s} {v['avg_launch_us']:8.1f} us  VALU {v['valu_insts_per_launch']/1e6:8.1f} M  "
              f"{v['valu_ginst_per_s']:6.1f} G/s = {100*v['valu_issue_frac']:5.1f}% of issue peak  "
              f"issuing {100*v['wave_cycles_issuing']:4.1f}%  stalled {100*v['wave_cycles_issue_stalled']:4.1f}%  "
              f"parked {100*v['wave_cycles_parked']:4.1f}%")


if __name__ == "__main__":
    main(*sys.argv[1:4])
