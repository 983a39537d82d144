#!/bin/bash
# Run on the GPU box (through gpurun), ONCE: the solver half of configs[4] staged (tools/xl_solver_bench.py) and the rocprofv3 passes of its evaluation
# kernel, gp_eval_compat_mfma_kernel<16, ...> -- counters restricted to that kernel, each pass loads the fit the first run saved.
#   tools/xl_solver_counters.sh <tag>
set -e
tag=$1
out=$PWD/gpurun_out
export TMPDIR=/tmp
state=$out/${tag}_xl_state.npz
python3 tools/xl_solver_bench.py --state $state > $out/${tag}_xl_solver_line.json
echo "fit + timed run done" >&2
filter='gp_eval_compat_mfma_kernel'
run="python3 tools/xl_solver_bench.py --state $state --steps 3"
rocprofv3 --kernel-trace --stats -d $out/${tag}_xl_trace -o run --output-format csv -- $run > /dev/null
rocprofv3 --kernel-trace --kernel-include-regex $filter --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/${tag}_xl_pmc_sq -o run --output-format csv -- $run > /dev/null
echo "sq pass done" >&2
rocprofv3 --kernel-trace --kernel-include-regex $filter --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $out/${tag}_xl_pmc_sq2 -o run --output-format csv -- $run > /dev/null || echo "pmc_sq2 pass failed" >&2
rocprofv3 --kernel-trace --kernel-include-regex $filter --pmc FETCH_SIZE -d $out/${tag}_xl_pmc_fetch -o run --output-format csv -- $run > /dev/null
rocprofv3 --kernel-trace --kernel-include-regex $filter --pmc WRITE_SIZE -d $out/${tag}_xl_pmc_write -o run --output-format csv -- $run > /dev/null
rocprofv3 --kernel-trace --kernel-include-regex $filter --pmc TCC_HIT_sum TCC_MISS_sum -d $out/${tag}_xl_pmc_l2 -o run --output-format csv -- $run > /dev/null || echo "l2 pass failed" >&2
echo "counter passes done" >&2
rm -f $out/${tag}_xl_pmc_*/*kernel_trace.csv $state
du -sh $out/${tag}_xl_* | tail -8
cat $out/${tag}_xl_solver_line.json
