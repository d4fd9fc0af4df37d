: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
for t in base o1 o2 o4 o5 all base; do SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_$t.so timeout 300 python tools/lean_ab.py c2 $([ $t = all ] && echo --check) > $O/ab_$t.log 2>&1; echo "variant $t: $(grep 'rollouts:' $O/ab_$t.log | cut -c 1-80) | $(grep 'one rollout' $O/ab_$t.log | cut -c 23-60) | $(grep fingerprint $O/ab_$t.log | cut -c 1-100)"; done
grep "oracle rollout" $O/ab_all.log
SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_all.so timeout 300 python tools/lean_ab.py c5 > $O/ab_all_c5.log 2>&1; grep "rollouts:\|fingerprint" $O/ab_all_c5.log | cut -c 1-110
SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_base.so timeout 300 python tools/lean_ab.py c5 > $O/ab_base_c5.log 2>&1; grep "rollouts:\|fingerprint" $O/ab_base_c5.log | cut -c 1-110
timeout 900 python -m pytest tests/test_ssm_gpu.py tests/test_lqr_gpu.py tests/test_ssm_controllers_gpu.py "tests/test_gusto_bench_shapes_gpu.py::test_short_horizon_wave_form_matches_box_form_and_oracle" "tests/test_gusto_bench_shapes_gpu.py::test_reference_driver_horizons_match_oracle" -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python tools/time_c3.py > $O/time_c3.log 2>&1; tail -8 $O/time_c3.log | head -4
SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libsofacontrol_hip_lqrprof.so timeout 120 python tools/prof_c3.py 1 > $O/prof_c3_512.log 2>&1; grep "ilqr" $O/prof_c3_512.log | tail -4
timeout 600 python tools/probes/lean_phase_clocks.py $O/lean_phase_clocks.json > $O/phase.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d/lean_phase_clocks.json'))
for k,c in d['cases'].items():
    cl=c.get('clocks_last_solve') or {}
    print(k, c.get('product',{}).get('kernel'), c.get('product',{}).get('ms_per_scp_iteration_median'), cl.get('qp_laps',{}).get('grad+newton'), cl.get('qp_tail'))
PY
