#!/bin/bash
# Run on the GPU box (through gpurun): matrix-pipe counters of the factorisation kernels at a given M.
# (SQ counters only: a FETCH_SIZE / WRITE_SIZE pass over the ~17 000 launches of two factorisations at M = 70 016 did not finish in 7 minutes.)
#   tools/chol_counters.sh <M> <tag>     writes gpurun_out/<tag>_trace, gpurun_out/<tag>_pmc and gpurun_out/<tag>_cholesky_counters.txt
set -e
M=$1; tag=$2
out=$PWD/gpurun_out
export TMPDIR=/tmp
python3 tools/chol_bench.py $M 3 > $out/${tag}_cholesky_counters.txt
rocprofv3 --kernel-trace -d $out/${tag}_trace -o run --output-format csv -- python3 tools/chol_bench.py $M 1 > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $out/${tag}_pmc -o run --output-format csv -- python3 tools/chol_bench.py $M 1 > /dev/null
python3 - "$out" "$tag" "$M" >> $out/${tag}_cholesky_counters.txt <<'PY'
import collections, csv, glob, os, sys
out, tag, M = sys.argv[1], sys.argv[2], int(sys.argv[3])
tr = glob.glob(os.path.join(out, tag + "_trace", "*kernel_trace.csv"))[0]
t = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(tr)):
    k = r["Kernel_Name"]
    t[k][0] += 1
    t[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
c = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(glob.glob(os.path.join(out, tag + "_pmc", "*counter_collection.csv"))[0])):
    c[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
# durations under the counters (the PMC pass serialises the two streams): from the PMC pass's own trace
tp = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob(os.path.join(out, tag + "_pmc", "*kernel_trace.csv"))[0])):
    tp[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("\n# rocprofv3 --kernel-trace (2 factorisations: warm-up + 1) | kernel | launches | total ms")
for k in sorted(t, key=lambda k: -t[k][1])[:8]:
    print("%-100s | %6d | %9.2f" % (k[:100], t[k][0], t[k][1]))
print("\n# rocprofv3 --pmc pass (separate run): kernel | MFMA instr | matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE) | clock GHz = GRBM_GUI_ACTIVE / 8 / time | FP64 TFLOP/s of its MFMAs (2048 flop each) over its time")
for k in sorted(c, key=lambda k: -c[k].get("SQ_INSTS_MFMA", 0))[:4]:
    v = c[k]
    if not v.get("GRBM_GUI_ACTIVE"):
        continue
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128 * v["GRBM_GUI_ACTIVE"])
    ghz = v["GRBM_GUI_ACTIVE"] / 8 / (tp[k] * 1e-3) / 1e9 if tp[k] else 0
    tf = v.get("SQ_INSTS_MFMA", 0) * 2048 / (tp[k] * 1e-3) / 1e12 if tp[k] else 0
    print("%-100s | %.3g | %.3f | %.2f | %.1f" % (k[:100], v.get("SQ_INSTS_MFMA", 0), busy, ghz, tf))
PY
rm -f $out/${tag}_pmc/*kernel_trace.csv $out/${tag}_trace/*kernel_trace.csv
cat $out/${tag}_cholesky_counters.txt
