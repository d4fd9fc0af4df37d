: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
for t in prev new prev new; do if [ $t = prev ]; then export SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_prev.so; else unset SRH_LIB_PATH; fi; timeout 300 python tools/lean_ab.py c2 > $O/ab_$t.log 2>&1; echo "variant $t: $(grep 'rollouts:' $O/ab_$t.log | cut -c 1-80) | $(grep 'one rollout' $O/ab_$t.log | cut -c 23-60) | $(grep fingerprint $O/ab_$t.log | cut -c 1-100)"; done
unset SRH_LIB_PATH
timeout 300 python tools/lean_ab.py c5 > $O/ab_new_c5.log 2>&1; grep "rollouts:" $O/ab_new_c5.log | cut -c 1-80
SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_prev.so timeout 300 python tools/lean_ab.py c5 > $O/ab_prev_c5.log 2>&1; grep "rollouts:" $O/ab_prev_c5.log | cut -c 1-80
timeout 1200 python -m pytest tests/test_gusto_bench_shapes_gpu.py tests/test_lean_gpu.py tests/test_gusto_gpu.py tests/test_locp_gpu.py tests/test_controllers_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 600 python tools/probes/lean_phase_clocks.py $O/lean_phase_clocks.json > $O/phase.log 2>&1
python tools/probes/summarise_phase_clocks.py profiles/r05a_lean_phase_clocks_raw.json $O/lean_phase_clocks.json $O/r05_lean_phase_clocks.json
