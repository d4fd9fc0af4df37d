: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 900 python tools/probes/lean_phase_clocks.py gpurun_out/r05b/lean_phase_clocks.json > gpurun_out/r05b/phase.log 2>&1; echo "phase rc $?"
for t in 64 128 256 512; do SRH_ILQR_THREADS=$t SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libsofacontrol_hip_lqrprof.so timeout 120 python tools/prof_c3.py 1 > gpurun_out/r05b/prof_c3_$t.log 2>&1; echo "threads $t"; grep "ilqr\|batch" gpurun_out/r05b/prof_c3_$t.log | tail -3; done
timeout 600 python -m pytest tests/test_gusto_bench_shapes_gpu.py tests/test_lean_gpu.py -m gpu -x -q > gpurun_out/r05b/pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r05b/pytest.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05b/lean_phase_clocks.json'))
for k,c in d['cases'].items():
    print(k, c.get('product',{}).get('ms_per_scp_iteration_median'), c.get('profile',{}).get('ms_per_scp_iteration_median'), c.get('profile',{}).get('scp_iterations'))
    print(json.dumps(c.get('clocks_last_solve')))
PY
