set -e
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_small_trace -o run --output-format csv -- python3 tools/chol_bench.py 4224 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
t=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open("gpurun_out/r02_small_trace/run_kernel_trace.csv")):
    k=r["Kernel_Name"][:70]; t[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; n[k]+=1
for k in sorted(t,key=t.get,reverse=True)[:8]: print("%-72s calls %5d avg %8.1f us total %9.1f us"%(k,n[k],t[k]/n[k],t[k]))
PY
rm -rf gpurun_out/r02_small_trace
