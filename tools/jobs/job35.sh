set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out
{
echo "# tools/chol_bench.py 35008: scasml_cholesky alone, M = 35 008 (9.8 GB float64), current code"
python tools/chol_bench.py 35008 3 2>/dev/null
echo
echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES (2 factorisations; counters serialise the two streams)"
} > $out/r02_cholesky_counters.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $out/r02_chol_pmc -o run --output-format csv -- python3 tools/chol_bench.py 35008 1 > /dev/null 2>&1
python3 - "$out" >> $out/r02_cholesky_counters.txt <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
dur = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(out, "r02_chol_pmc", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[k] += 1
    os.remove(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "r02_chol_pmc", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    os.remove(f)
print("# kernel | launches | total ms under the counters | MFMA instr | matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE) | clock GHz = GRBM_GUI_ACTIVE / 8 / time")
for k in sorted(dur, key=dur.get, reverse=True):
    if "scasml" not in k: continue
    a = acc[k]
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    print("%-95s | %5d | %9.2f | %.3g | %.3f | %.2f" % (k[:95], n[k], dur[k], a.get("SQ_INSTS_MFMA", 0), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128 * gui) if gui else 0, gui / 8 / (dur[k] * 1e6) if dur[k] else 0))
PY
cat $out/r02_cholesky_counters.txt
