# on the GPU box: correctness (product library) + phase clocks (profile library) of the lean kernels
python tools/dev_lean.py c2 > gpurun_out/dev_c2.log 2>&1; tail -8 gpurun_out/dev_c2.log
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT    # the repository root (gpurun exports it; derived from $0 elsewhere)
export SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libsofacontrol_hip_prof.so
SRH_LOCP_TRACE=1 python tools/trace_c2_qp.py c2 > gpurun_out/trace_c2.log 2>&1; grep "lean. laps\|lean. newton\|lean. j0\|^J" gpurun_out/trace_c2.log
SRH_LOCP_TRACE=1 python tools/trace_c2_qp.py c5 > gpurun_out/trace_c5.log 2>&1; grep "lean. laps\|lean. newton\|lean. j0\|^J" gpurun_out/trace_c5.log
python tools/probes/gusto_prof.py > gpurun_out/gusto_prof.log 2>&1; grep -v "^.qp" gpurun_out/gusto_prof.log | tail -3
