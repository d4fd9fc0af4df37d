# End-of-round measurement pass on the GPU box (gpurun -- bash tools/run_round.sh r05): the full GPU test suite, the default bench line, the rocprofv3 passes
# of tools/prof_round.sh, the phase-clock table of the lean kernel, the determinism probe and the two C3 forms; everything under gpurun_out/<tag>/
# (+ gpurun_out/profiles_<tag>/: the summaries to copy into profiles/).
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
TAG=${1:?usage: bash tools/run_round.sh <tag, e.g. r05>}; export TAG
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"
bash tools/prof_round.sh $TAG > $O/prof_round.log 2>&1; echo "prof rc $?"; tail -2 $O/prof_round.log
timeout 600 python tools/probes/lean_phase_clocks.py $O/lean_phase_clocks.json > $O/phase.log 2>&1
python tools/probes/summarise_phase_clocks.py ${PHASE_RAW:-profiles/${TAG}a_lean_phase_clocks_raw.json} $O/lean_phase_clocks.json $O/${TAG}_lean_phase_clocks.json
timeout 300 python tools/determinism_probe.py c2 > $O/determinism.log 2>&1; tail -3 $O/determinism.log
timeout 300 python tools/time_c3.py > $O/time_c3.log 2>&1; tail -6 $O/time_c3.log
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null; ls gpurun_out/profiles_$TAG
python - <<'PY'
import json
import os
lines=open('gpurun_out/%s/bench.log' % os.environ['TAG']).read().strip().splitlines()
d=json.loads(lines[-1])
assert len(lines[-1]) <= 4096, len(lines[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline'], 'vs', d.get('vs_baseline'), 'line bytes', len(lines[-1]))
full=[l for l in lines if l.startswith('BENCH_DETAIL ')]
if full:
    sec=json.loads(full[-1].split(' ', 1)[1]).get('secondary', {})
    print('c5', {k:v for k,v in sec.get('scp_c5', {}).items() if k in ('ms','ms_all_calls')})
    print('ssm_gusto_rti', {k:v for k,v in sec.get('ssm_gusto_rti', {}).items() if k in ('ms_median','ms_p95','kernel')})
PY
