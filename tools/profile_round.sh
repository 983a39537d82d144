#!/bin/bash
# Run on the GPU box (through gpurun): the rocprofv3 passes whose summaries profiles/summarize.py condenses.
#   tools/profile_round.sh <tag> [bench args...]          writes gpurun_out/<tag>_{trace,pmc_sq,pmc_sq2,pmc_fetch,pmc_write}/
# Counters are collected in runs of their own with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; 8 SQ slots per pass).
set -e
tag=$1; shift
out=$PWD/gpurun_out
export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline --no-gp-train-large --no-reference-logs-check --no-other-runs $*"
python3 bench.py --steps 5 --warmup 2 $* > $out/${tag}_bench_line.json
rocprofv3 -L > $out/${tag}_counters_list.txt 2>&1 || true
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o run --output-format csv -- python3 bench.py $args > $out/${tag}_bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/${tag}_pmc_sq -o run --output-format csv -- python3 bench.py $args > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $out/${tag}_pmc_sq2 -o run --output-format csv -- python3 bench.py $args > /dev/null || echo "pmc_sq2 pass failed" >&2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o run --output-format csv -- python3 bench.py $args > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o run --output-format csv -- python3 bench.py $args > /dev/null
# gpurun merges at most 64 MiB back: keep the counter rows of the path's kernels only, drop the per-dispatch traces of the PMC passes
python3 - "$out" "$tag" <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(out, tag + "_pmc_*", "*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "gp_eval" in r["Kernel_Name"] or "picard_tree" in r["Kernel_Name"]]
    with open(f, "w", newline="") as h:
        w = csv.DictWriter(h, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
for f in glob.glob(os.path.join(out, tag + "_pmc_*", "*_kernel_trace.csv")):
    os.remove(f)
PY
du -sh $out/${tag}_* | tail -12
cat $out/${tag}_bench_line.json
