# instruction mix and LDS behaviour of the lean GuSTO kernel over one bench step (run on the GPU box from the repository root):
# three separate --pmc passes (counter limits), summarised into profiles/r03_lean_instruction_mix.json by
# tools/prof_r03_lean_mix_summarise.py
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT    # the repository root (gpurun exports it; derived from $0 elsewhere)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/r03_mix_a -o a -- $B > $R/gpurun_out/r03_mix_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r03_mix_b -o b -- $B > $R/gpurun_out/r03_mix_b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r03_mix_c -o c -- $B > $R/gpurun_out/r03_mix_c.log 2>&1
ls $R/gpurun_out/r03_mix_a $R/gpurun_out/r03_mix_b $R/gpurun_out/r03_mix_c
