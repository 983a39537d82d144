#!/bin/bash
# Run on the GPU box (through gpurun), ONCE: HBM traffic of the trailing update of ONE factorisation at a given M.
# The FETCH_SIZE / WRITE_SIZE passes are restricted to the trailing-update kernel with rocprofv3's kernel filter (the unfiltered passes
# over the ~8 500 tiny diag / panel launches of a factorisation did not finish in 7 minutes: tools/chol_counters.sh), one factorisation, no warm-up.
#   tools/chol_hbm_counters.sh <M> <tag>     writes gpurun_out/<tag>_cholesky_hbm.txt
set -e
M=$1; tag=$2
out=$PWD/gpurun_out
export TMPDIR=/tmp
filter='chol_update_dma_kernel|chol_update_k_kernel<2, 4>'   # the 128 x 128 trailing update: LDS-DMA tile (round 6) or the register-staged one
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$filter" -d $out/${tag}_chol_fetch -o run --output-format csv -- python3 tools/chol_bench.py $M 0 > $out/${tag}_chol_fetch.log
echo "fetch pass done" >&2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$filter" -d $out/${tag}_chol_write -o run --output-format csv -- python3 tools/chol_bench.py $M 0 > $out/${tag}_chol_write.log
echo "write pass done" >&2
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-include-regex "$filter" -d $out/${tag}_chol_sq -o run --output-format csv -- python3 tools/chol_bench.py $M 0 > $out/${tag}_chol_sq.log
echo "sq pass done" >&2
python3 - "$out" "$tag" "$M" > $out/${tag}_cholesky_hbm.txt <<'PY'
import collections, csv, glob, os, sys
out, tag, M = sys.argv[1], sys.argv[2], int(sys.argv[3])
res = {}
for name in ("fetch", "write"):
    d = os.path.join(out, "%s_chol_%s" % (tag, name))
    tot = collections.defaultdict(float)
    n = collections.defaultdict(int)
    unit = None
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
        tot[(r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        n[(r["Kernel_Name"], r["Counter_Name"])] += 1
    t = collections.defaultdict(float)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*kernel_trace.csv"))[0])):
        t[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    res[name] = (tot, n, t)
    print("# %s pass: %s" % (name, open(os.path.join(out, "%s_chol_%s.log" % (tag, name))).read().strip()))
for (k, c), v in sorted(res["fetch"][0].items()):
    ms = res["fetch"][2][k]
    # FETCH_SIZE is in KiB of 64-byte requests; on gfx950 a wide streaming read is tallied at half its bytes (MI355X_MICROARCH.md, HBM): doubled
    gb = v * 1024 * 2 / 1e9
    print("%s | launches %d | FETCH_SIZE raw %.4g KiB -> %.2f GB read (doubled) | kernel time in this pass %.1f ms | %.2f TB/s" % (k[:80], res["fetch"][1][(k, c)], v, gb, ms, gb / ms))
for (k, c), v in sorted(res["write"][0].items()):
    ms = res["write"][2][k]
    gb = v * 1024 / 1e9
    print("%s | launches %d | WRITE_SIZE raw %.4g KiB -> %.2f GB written | kernel time in this pass %.1f ms | %.2f TB/s" % (k[:80], res["write"][1][(k, c)], v, gb, ms, gb / ms))
# matrix pipe and clock of the same kernel (third pass)
d = os.path.join(out, "%s_chol_sq" % tag)
c = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
    c[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
tp = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob(os.path.join(d, "*kernel_trace.csv"))[0])):
    tp[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("# sq pass: %s" % open(os.path.join(out, "%s_chol_sq.log" % tag)).read().strip())
for k, v in c.items():
    if v.get("GRBM_GUI_ACTIVE") and tp[k]:
        print("%s | MFMA instr %.4g | matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE) %.3f | clock %.2f GHz | %.1f TFLOP/s of MFMA work over %.1f ms" % (
            k[:80], v.get("SQ_INSTS_MFMA", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128 * v["GRBM_GUI_ACTIVE"]), v["GRBM_GUI_ACTIVE"] / 8 / (tp[k] * 1e-3) / 1e9,
            v.get("SQ_INSTS_MFMA", 0) * 2048 / (tp[k] * 1e-3) / 1e12, tp[k]))
# algorithmic traffic of a blocked right-looking factorisation with nb-column outer panels and 128 x 128 trailing tiles: every trailing element is read
# and written once per panel step: sum over panels of (remaining rows)^2 / 2 * 8 bytes, each way
nb = 256 if M < 8192 else (512 if M < 49152 else 1024)            # outer_rows(M), csrc/gp_train.hip
alg = sum(max(M - (k + 1) * nb, 0) ** 2 / 2 * 8 for k in range((M + nb - 1) // nb)) / 1e9
print("algorithmic: trailing matrix read once + written once per %d-column outer panel = %.1f GB each way (M = %d)" % (nb, alg, M))
PY
rm -rf $out/${tag}_chol_fetch $out/${tag}_chol_write $out/${tag}_chol_sq
cat $out/${tag}_cholesky_hbm.txt
