: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -c 600 $O/bench.err
python - <<'PY'
import json
l=open('gpurun_out/r05e/bench.log').read().strip().splitlines()[-1]
d=json.loads(l)
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'vs', d.get('vs_baseline'))
print('cpu_baseline', {k: d['cpu_baseline'].get(k) for k in ('value','cores')} if 'cpu_baseline' in d else None)
s=d['secondary']
for k in ('ilqr_c3','ilqr_diamond'):
    print(k, json.dumps(s[k])[:900])
for k,v in s['scp_reference_horizons'].items():
    print(k, json.dumps(v)[:1100])
print('scp_c5', json.dumps(s['scp_c5'])[:900])
print('uncapped', json.dumps(s['scp_uncapped_500'])[:700])
print('gram', json.dumps(s['gramian_c4'])[:500])
print('single', json.dumps(s['scp_single_rollout'])[:600])
PY
