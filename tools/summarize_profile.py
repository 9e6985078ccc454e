#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/ into the small files kept under profiles/.

    python tools/summarize_profile.py <tag> <kernel_stats.csv> [<fetch counter csv> <write counter csv>]

Traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read at half its
bytes (128-B requests tallied as 64 B), so reads are doubled; the two counters come from separate passes.
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for key, tag in (("gemm256p_kernel", "gemm256p"), ("gemm256p_fixup", "gemm256p_fixup"), ("gemm_kernel", "gemm128"),
                     ("mhsa_qkv_kernel", "mhsa"), ("layernorm_kernel", "layernorm")):
        if key in name:
            epi = [e for e in ("EpiBiasGeluHalf", "EpiBiasHalf", "EpiResidualF32", "EpiStoreF32", "EpiPatchEmbed") if e in name]
            return tag + ("<" + epi[0] + ">" if epi else "")
    m = re.search(r"_GLOBAL__N_1(\d+)", name)                 # mangled: _ZN12_GLOBAL__N_1<len><identifier>...
    if m:
        n0 = m.end()
        return name[n0:n0 + int(m.group(1))]
    m = re.search(r"\(anonymous namespace\)::(\w+)", name)  # demangled
    if m:
        return m.group(1)
    m = re.search(r"at::native::(\w+)", name)
    if m:
        return "at::native::" + m.group(1)
    return name.split("(")[0][-60:]


def per_kernel_counter(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows[:24]:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.1f}", r["Percentage"]])
    if len(sys.argv) >= 5:
        fetch, nf = per_kernel_counter(sys.argv[3], "FETCH_SIZE")
        write, _ = per_kernel_counter(sys.argv[4], "WRITE_SIZE")
        traffic = {}
        for k in fetch:
            rd, wr = 2.0 * fetch[k] * 1024, write.get(k, 0.0) * 1024
            traffic[k] = {"launches_sampled": nf[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                          "hbm_bytes_per_launch": rd + wr}
        with open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w") as f:
            json.dump({"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of "
                                 "`bench.py --steps 2 --warmup 1`; KiB units; reads doubled (gfx950 FETCH_SIZE "
                                 "tallies 128-B requests as 64 B)", "kernels": traffic}, f, indent=1)
    print("wrote", tag)


if __name__ == "__main__":
    main()
