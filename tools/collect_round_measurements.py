#!/usr/bin/env python3
"""Condenses what tools/round_measurements.sh left under gpurun_out/<tag>_meas into the files kept under profiles/:
    python tools/collect_round_measurements.py r3d r3        # -> profiles/r3_harness.json, r3_config5.json,
                                                             #    r3_config5_b4_kernel_stats.csv, r3_sweep.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_profile import short  # noqa: E402  (kernel-name shortening shared with the per-round profile sets)


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    tag, out = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"{tag}_meas")
    prof = os.path.join(ROOT, "profiles")

    harness = {"note": "tools/harness_bench.py on one MI355X box (files on disk -> decode -> upload -> device "
                       "preprocessing -> batched forward -> per-image rows); bf16, B=32, 16 host threads",
               "jpeg_518": last_json_line(os.path.join(src, "harness_518.json")),
               "jpeg_12mp": last_json_line(os.path.join(src, "harness_12mp.json"))}
    json.dump(harness, open(os.path.join(prof, f"{out}_harness.json"), "w"), indent=1)

    keep = ("value", "unit", "ms_per_step", "mfma_frac_end_to_end", "algorithmic_gflop_per_image", "streams_per_gpu",
            "kernel_breakdown_ms_per_step", "roofline", "mfma_frac_by_kernel")
    runs = {}
    for b in (1, 4, 16):
        line = last_json_line(os.path.join(src, f"config5_b{b}.json"))
        runs[f"B={b}"] = {k: line[k] for k in keep if k in line}
    json.dump({"note": "bench.py --size 1036 --iters 8 --dtype fp16 --batch B (BASELINE config 5: 5476 patches, 8 focal "
                       "iterations, focal scores never stored); same box, one call", "runs": runs},
              open(os.path.join(prof, f"{out}_config5.json"), "w"), indent=1)

    stats = glob.glob(os.path.join(src, "config5_trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = {}
        with open(stats[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Name"])
                c, t = rows.get(k, (0, 0.0))
                rows[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
        total = sum(t for _, t in rows.values())
        with open(os.path.join(prof, f"{out}_config5_b4_kernel_stats.csv"), "w") as f:
            f.write("kernel,calls,total_ms,avg_us,percent\n")
            for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{k},{c},{t / 1e6:.3f},{t / c / 1e3:.1f},{100 * t / total:.4g}\n")

    json.dump(last_json_line(os.path.join(src, "sweep.json")), open(os.path.join(prof, f"{out}_sweep.json"), "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
