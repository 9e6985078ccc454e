#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/ into the small files kept under profiles/.

    python tools/summarize_profile.py <tag> <kernel_stats.csv> [<fetch counter csv> <write counter csv>]
    python tools/summarize_profile.py <tag> <directory written by tools/profile_round.sh>

The directory form also copies the bench line of the same run and condenses the matrix-pipe pass
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_*, GRBM_GUI_ACTIVE) into
<tag>_mfma_util.json: per kernel the mean counter values per launch and
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs), the share of SIMD cycles in
which the matrix pipe was busy (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs).

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
    for key, tag in (("gemm256f_kernel", "gemm256f"), ("gemm256q_kernel", "gemm256q"), ("gemm128s_kernel", "gemm128s"),
                     ("gemm256p_kernel", "gemm256p"), ("gemm256p_fixup", "gemm256p_fixup"), ("gemm_kernel", "gemm128"),
                     ("mhsa_qkv_kernel", "mhsa"), ("layernorm_kernel", "layernorm")):
        if key in name:
            epi = [e for e in ("EpiBiasGeluHalf", "EpiBiasHalf", "EpiResidualF32", "EpiStoreF32", "EpiPatchEmbed", "EpiRowStats",
                               "EpiColSum") if e in name]
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


def find_csv(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError(f"{suffix} under {d}")


def mfma_util(path, out_path):
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
             "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
    per = {n: per_kernel_counter(path, n)[0] for n in names}
    res = {}
    for k in per["GRBM_GUI_ACTIVE"]:
        row = {n: per[n].get(k, 0.0) for n in names}
        gui = row["GRBM_GUI_ACTIVE"]
        row["mfma_util"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 256 * 4), 4) if gui else None
        # SQ_* wave counters are in quad-cycles: x4 for cycles
        wc = row["SQ_WAVE_CYCLES"]
        row["valu_issue_share_of_wave_cycles"] = round(row["SQ_ACTIVE_INST_VALU"] / wc, 4) if wc else None
        row["wait_share_of_wave_cycles"] = round(row["SQ_WAIT_ANY"] / wc, 4) if wc else None
        res[k] = row
    with open(out_path, "w") as f:
        json.dump({"method": "rocprofv3 --pmc (one pass, own run) of `bench.py --steps 3 --warmup 1`, mean per launch; "
                             "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)",
                   "kernels": {k: v for k, v in res.items()
                               if any(t in k for t in ("gemm", "mhsa", "layernorm", "focal", "weighted"))}}, f, indent=1)


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    if os.path.isdir(stats):
        d = stats
        import shutil
        shutil.copy(os.path.join(d, "bench.json"), os.path.join(out, f"{tag}_bench.json"))
        mfma_util(find_csv(os.path.join(d, "mfma"), "counter_collection.csv"), os.path.join(out, f"{tag}_mfma_util.json"))
        sys.argv[2:] = [find_csv(os.path.join(d, "trace"), "kernel_stats.csv"),
                        find_csv(os.path.join(d, "fetch"), "counter_collection.csv"),
                        find_csv(os.path.join(d, "write"), "counter_collection.csv")]
        stats = sys.argv[2]
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
