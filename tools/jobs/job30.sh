set -e
export TMPDIR=/tmp
python tools/chol_bench.py 35008 3
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $out/r02_chol_pmc -o run --output-format csv -- python3 tools/chol_bench.py 35008 1 > /dev/null
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "r02_chol_pmc", "*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
    for k in acc:
        a = acc[k]
        print(k, cnt[k], {c: "%.3g" % v for c, v in a.items()})
    os.remove(f)
for f in glob.glob(os.path.join(out, "r02_chol_pmc", "*_kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    t = collections.defaultdict(float); n = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"][:60]; t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[k] += 1
    for k in sorted(t, key=t.get, reverse=True)[:10]: print("%-62s calls %6d total %9.2f ms" % (k, n[k], t[k]))
    os.remove(f)
PY
