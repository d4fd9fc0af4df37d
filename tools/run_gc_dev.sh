# development run on the GPU box: probes, A/B of the dev build of the lean kernels, phase clocks (see tools/build_lean_dev.sh)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT    # the repository root (gpurun exports it; derived from $0 elsewhere)
cd $GRAFT_REPO_ROOT
timeout 60 ./gpurun_variants/lean_probe > gpurun_out/probe4.log 2>&1; echo "probe rc $?"; grep "tile_chol\|chol16\|chol 4\|set sync\|k_solve" gpurun_out/probe4.log
export SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libsofacontrol_hip_dev.so
timeout 300 python tools/lean_ab.py c2 --check > gpurun_out/ab_dev_c2.log 2>&1; echo "rc $?"; tail -n 7 gpurun_out/ab_dev_c2.log
timeout 300 python tools/lean_ab.py c5 > gpurun_out/ab_dev_c5.log 2>&1; tail -n 3 gpurun_out/ab_dev_c5.log
SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libsofacontrol_hip_devprof.so timeout 300 python tools/probes/gusto_prof.py > gpurun_out/prof_lean_dev.log 2>&1; grep -v "^.qp" gpurun_out/prof_lean_dev.log | tail -n 4
