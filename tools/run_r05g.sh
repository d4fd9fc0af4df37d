: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 600 python tools/probes/solve_outliers.py > $O/outliers.log 2>&1; grep -v amdgpu.ids $O/outliers.log | head -3
SRH_GUSTO_NO_ZEROCOPY=1 timeout 600 python tools/probes/solve_outliers.py > $O/outliers_copy.log 2>&1; grep -v amdgpu.ids $O/outliers_copy.log | head -2
timeout 900 python -m pytest tests/test_gusto_gpu.py tests/test_gusto_bench_shapes_gpu.py tests/test_controllers_gpu.py tests/test_pipeline_gpu.py tests/test_lean_gpu.py tests/test_dubins_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python tools/bench_pod_shapes.py > $O/pod_base.log 2>&1; grep -v amdgpu $O/pod_base.log
SRH_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libdev_mt3.so timeout 300 python tools/bench_pod_shapes.py > $O/pod_mt3.log 2>&1; echo MT3; grep -v amdgpu $O/pod_mt3.log
timeout 600 python tools/probes/lean_phase_clocks.py $O/lean_phase_clocks.json > $O/phase.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05g/lean_phase_clocks.json'))
for k,c in d['cases'].items():
    print(k, c.get('product',{}).get('kernel'), c.get('product',{}).get('ms_per_scp_iteration_median'), c.get('product',{}).get('ms_per_solve'))
PY
