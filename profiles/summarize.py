#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/) into the small summaries committed here.

    python profiles/summarize.py r01 gpurun_out/prof_r01 [gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq]

* <tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary, verbatim
* <tag>_kernel_by_grid.txt      the same trace grouped by (kernel, grid) so that the bench-size
                                launches are not averaged with the small setup / harness launches
  (extra key=value arguments are copied into the summary: n_inf=... d=... mode=reference|reference-geometry|none -- bench.py quotes the
   summary of the surrogate it runs; out=<name> replaces <tag>_gp_eval_pmc.json / <tag>_picard_pmc.json by <name>_...)
* <tag>_gp_eval_pmc.json        PMC counters of the bench-size gp_eval launch.  HBM bytes follow
                                MI355X_MICROARCH.md "HBM": bytes = 2*FETCH_SIZE*1024 (gfx950 reports
                                half of a wide coalesced read) + WRITE_SIZE*1024, separate passes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def one(pattern):
    """First match of dir/*/*suffix or, for rocprofv3 runs with -o, dir/*suffix."""
    f = glob.glob(pattern) or glob.glob(pattern.replace(os.sep + "*" + os.sep, os.sep, 1))
    return f[0] if f else None


def main():
    tag, trace_dir = sys.argv[1], sys.argv[2]
    pmc_dirs = [a for a in sys.argv[3:] if "=" not in a]
    extra = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a)      # e.g. n_inf=10911744 d=100 split=3
    stats = one(os.path.join(trace_dir, "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(HERE, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(one(os.path.join(trace_dir, "*", "*_kernel_trace.csv")))))
    groups = collections.defaultdict(list)
    meta = {}
    for r in rows:
        key = (r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        meta[key] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
    with open(os.path.join(HERE, tag + "_kernel_by_grid.txt"), "w") as f:
        f.write("# kernel | grid threads | calls | avg ms | min ms | max ms | median ms | vgpr accum_vgpr sgpr scratch lds\n"
                "# (the first launch of a kernel in a process is cold -- code load, clock ramp -- and is the max; bench.py's\n"
                "#  avg_launch_ms covers the timed steps only and should be compared with the median)\n")
        for key in sorted(groups, key=lambda k: -sum(groups[k])):
            v = groups[key]
            f.write("%s | %d | %d | %.4f | %.4f | %.4f | %.4f | %s\n" % (key[0], key[1], len(v), sum(v) / len(v), min(v), max(v),
                                                                     sorted(v)[len(v) // 2], " ".join(meta[key])))
    if pmc_dirs:
        big = max((k for k in groups if "gp_eval" in k[0]), key=lambda k: k[1])
        out = {"kernel": big[0], "grid_threads": big[1], "avg_ms_kernel_trace": sum(groups[big]) / len(groups[big])}
        out.update({k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in extra.items()})
        # the code the counters were taken on: bench.py quotes a summary only for the same kernel sources
        import bench
        out["source_sha1"] = bench.kernel_source_sha1(bench.GP_EVAL_SOURCES["reference" if "compat" in big[0] else "none"])
        for d in pmc_dirs:
            for r in csv.DictReader(open(one(os.path.join(d, "*", "*_counter_collection.csv")))):
                if r["Kernel_Name"] == big[0] and int(r["Grid_Size"]) == big[1]:
                    out.setdefault("counters", {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        c = {k: sum(v) / len(v) for k, v in out.pop("counters", {}).items()}
        out["counters_avg_per_launch"] = c
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out["hbm_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # sum over 8 XCDs
            out["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)   # 1024 SIMDs
            out["effective_clock_ghz"] = cyc / (out["avg_ms_kernel_trace"] * 1e-3) / 1e9
            if "SQ_ACTIVE_INST_VALU" in c:                        # quad-cycles (MI355X_MICROARCH.md, cycle constants)
                out["valu_active_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc)
                out["cycles_per_valu_instruction"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"] if "SQ_INSTS_VALU" in c else None
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
                out["coexec_frac_of_mfma_busy"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]
        out_tag = out.pop("out", tag)
        json.dump(out, open(os.path.join(HERE, out_tag + "_gp_eval_pmc.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))
        # the two Picard-tree passes of the same step (bench-size launches = the largest grid of each kernel)
        pic = {}
        for name in sorted({k[0] for k in groups if "picard_tree_kernel" in k[0]}):
            big_p = max((k for k in groups if k[0] == name), key=lambda k: k[1])
            ent = {"grid_threads": big_p[1], "avg_ms_kernel_trace": sum(groups[big_p]) / len(groups[big_p])}
            cnt = {}
            for d in pmc_dirs:
                for r in csv.DictReader(open(one(os.path.join(d, "*", "*_counter_collection.csv")))):
                    if r["Kernel_Name"] == name and int(r["Grid_Size"]) == big_p[1]:
                        cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            cnt = {k: sum(v) / len(v) for k, v in cnt.items()}
            if "FETCH_SIZE" in cnt and "WRITE_SIZE" in cnt:
                ent["hbm_bytes_per_launch"] = 2.0 * cnt["FETCH_SIZE"] * 1024 + cnt["WRITE_SIZE"] * 1024
                ent["hbm_gb_per_s"] = ent["hbm_bytes_per_launch"] / (ent["avg_ms_kernel_trace"] * 1e-3) / 1e9
                ent["hbm_frac_of_8tbs"] = ent["hbm_gb_per_s"] / 8000.0
            ent["counters_avg_per_launch"] = cnt
            pic[name] = ent
        pic["source_sha1"] = bench.kernel_source_sha1(bench.PICARD_SOURCES)
        json.dump(pic, open(os.path.join(HERE, out_tag + "_picard_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
