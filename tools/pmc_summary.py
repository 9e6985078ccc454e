#!/usr/bin/env python3
"""Median per-dispatch PMC counter values per kernel from a rocprofv3 results .db (rocpd schema).
usage: pmc_summary.py results.db [kernel-name-substring]"""
import collections
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
sfx = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch_"))[len("rocpd_kernel_dispatch_"):]
q = f"""select k.kernel_name, d.id, d.end - d.start, p.name, e.value
        from rocpd_kernel_dispatch_{sfx} d
        join rocpd_info_kernel_symbol_{sfx} k on k.id = d.kernel_id
        join rocpd_pmc_event_{sfx} e on e.event_id = d.event_id
        join rocpd_info_pmc_{sfx} p on p.id = e.pmc_id"""
vals = collections.defaultdict(lambda: collections.defaultdict(float))
dur = {}
for kn, did, du, pn, v in db.execute(q):
    if pat in kn:
        vals[(kn, did)][pn] += v          # summed over XCDs / shader engines
        dur[(kn, did)] = du
by_kernel = collections.defaultdict(list)
for (kn, did) in vals:
    by_kernel[kn].append(did)
for kn, dids in by_kernel.items():
    print(f"{kn[:100]}  dispatches={len(dids)}  median_us={statistics.median(dur[(kn, d)] for d in dids) / 1e3:.1f}")
    for n in sorted(vals[(kn, dids[0])]):
        print(f"    {n:32s} {statistics.median(vals[(kn, d)][n] for d in dids):.4g}")
