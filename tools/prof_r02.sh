: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT    # the repository root (gpurun exports it; derived from $0 elsewhere)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_trace -o r02 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/r02f_bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_fetch -o f -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/r02_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_write -o w -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/r02_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_mfma -o m -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 > $R/gpurun_out/r02_pmc_mfma.log 2>&1
ls -R $R/gpurun_out/r02_trace $R/gpurun_out/r02_pmc_fetch $R/gpurun_out/r02_pmc_mfma | head -30
