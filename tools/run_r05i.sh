: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -c 300 $O/bench.err
bash tools/prof_round.sh r05 > $O/prof_round.log 2>&1; echo "prof rc $?"; tail -3 $O/prof_round.log
timeout 600 python tools/probes/lean_phase_clocks.py $O/lean_phase_clocks.json > $O/phase.log 2>&1
mkdir -p gpurun_out/profiles_r05; cp profiles/r05_* gpurun_out/profiles_r05/ 2>/dev/null; ls gpurun_out/profiles_r05 | head -30
python - <<'PY'
import json
l=open('gpurun_out/r05i/bench.log').read().strip().splitlines()[-1]
d=json.loads(l)
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'vs', d.get('vs_baseline'))
s=d['secondary']
for k,v in s['scp_reference_horizons'].items():
    print(k, v.get('ms_per_scp_iteration_median'), v.get('ms_per_solve_max'), v.get('max_over_median'), v.get('batch_of_8'), v.get('cpu'))
print('single', s['scp_single_rollout']['ms_per_scp_iteration'])
PY
