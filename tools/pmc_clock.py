#!/usr/bin/env python
"""Effective shader clock and pipe activity of the bench-size gp_eval launch from one rocprofv3 --pmc pass:
    python tools/pmc_clock.py gpurun_out/pmc_dir [...]
clock = GRBM_GUI_ACTIVE / 8 XCDs / (End - Start)."""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not f:
        print(d, "no counter file"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "gp_eval" in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in rows)
    acc, dur = collections.defaultdict(list), []
    for r in rows:
        if int(r["Grid_Size"]) == big:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    ms = sum(dur) / len(dur)
    line = "%s: %.3f ms" % (d, ms)
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        line += "  clock %.0f MHz" % (cyc / ms / 1e3)
        for k, v in sorted(c.items()):
            if k != "GRBM_GUI_ACTIVE":
                line += "  %s/cycle/CU %.3f" % (k, v / 256.0 / cyc)
    print(line)
