cd $GRAFT_REPO_ROOT
for t in e f g; do export SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_$t.so; timeout 300 python tools/lean_ab.py c2 > gpurun_out/ab_$t.log 2>&1; echo "variant $t: $(grep 'rollouts:' gpurun_out/ab_$t.log | cut -c 1-90) | $(grep 'one rollout' gpurun_out/ab_$t.log | cut -c 1-60)"; done
