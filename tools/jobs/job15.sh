set -e
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_gp_large_trace -o run --output-format csv -- python3 tools/gp_large.py --roots 256 > gpurun_out/r02_gp_large.txt 2>&1
cat gpurun_out/r02_gp_large.txt | grep -v amdgpu
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02_gp_large_trace/run_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), ("%.1f ms" % (int(r['TotalDurationNs'])/1e6)).rjust(12), r['Percentage'])
PY
