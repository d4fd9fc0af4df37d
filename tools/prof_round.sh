# rocprofv3 passes behind profiles/<tag>_* (run on the GPU box from the repository root: bash tools/prof_round.sh r04).
# Kernel trace + stats of a short bench run; PMC passes, one group of counters each (never combined with a trace domain other
# than --kernel-trace): HBM traffic of the projection, MFMA-busy, and for the lean GuSTO kernel the instruction mix, where its
# wave cycles go (parked / issue-stalled / active), LDS bank conflicts and instruction-cache misses.
TAG=${1:-r04}
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT    # the repository root (gpurun exports it; derived from $0 elsewhere)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B1="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o $TAG -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_mfma -o m -- $B1 > $R/gpurun_out/${TAG}_pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mix_a -o a -- $B1 > $R/gpurun_out/${TAG}_mix_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mix_b -o b -- $B1 > $R/gpurun_out/${TAG}_mix_b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mix_c -o c -- $B1 > $R/gpurun_out/${TAG}_mix_c.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mix_d -o d -- $B1 > $R/gpurun_out/${TAG}_mix_d.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mix_e -o e -- $B1 > $R/gpurun_out/${TAG}_mix_e.log 2>&1
cd $R; python3 tools/prof_summarise.py $TAG
