# A/B of dev libraries gpurun_variants/libdev_<tag>.so (tools/build_lean_dev.sh + cp): bash tools/run_ab_variants.sh [c2|c5] tag1 tag2 ...
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT    # the repository root (gpurun exports it; derived from $0 elsewhere)
cd $GRAFT_REPO_ROOT
W=c2; if [ "$1" = c5 ] || [ "$1" = c2 ]; then W=$1; shift; fi
for t in "$@"; do export SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_$t.so; timeout 300 python tools/lean_ab.py $W > gpurun_out/ab_$t.log 2>&1; echo "variant $t: $(grep 'rollouts:' gpurun_out/ab_$t.log | cut -c 1-90) | $(grep 'one rollout' gpurun_out/ab_$t.log | cut -c 1-60) | $(grep fingerprint gpurun_out/ab_$t.log | cut -c 1-60)"; done
