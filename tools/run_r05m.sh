: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -c 300 $O/bench.err
python - <<'PY'
import json
l=open('gpurun_out/r05m/bench.log').read().strip().splitlines()[-1]
d=json.loads(l)
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'vs', d.get('vs_baseline'))
s=d['secondary']
for k,v in s['scp_reference_horizons'].items():
    print(k, v.get('ms_per_scp_iteration_median'), 'max', v.get('ms_per_solve_max'), v.get('max_over_median'), 'keep', v.get('keep_solver_state'), 'batch', v.get('batch_of_8'), 'cpu', (v.get('cpu') or {}).get('ms_per_scp_iteration'))
print('single', s['scp_single_rollout']['ms_per_scp_iteration'], 'c5', s['scp_c5']['ms'], 'c3', s['ilqr_c3']['ms'], s['ilqr_c3']['one_problem_ms'])
PY
grep thrott /sys/fs/cgroup/cpu.stat
