: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_pod_gpu.py tests/test_edge_cases_gpu.py tests/test_snapshots_gpu.py "tests/test_gusto_bench_shapes_gpu.py::test_forced_hand_over_gives_the_same_solve" tests/test_gusto_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
timeout 300 python tools/bench_pod_shapes.py > $O/pod.log 2>&1; grep -v amdgpu $O/pod.log
