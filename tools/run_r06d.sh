#!/bin/bash
# round 6, call d: the fused SSM GuSTO kernel (tests + the RTI latency), interior-point iteration counts of the uncapped tail, batched U^T M U
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd $GRAFT_REPO_ROOT
echo "== ssm tests"; timeout 900 python -m pytest tests/test_ssm_gpu.py tests/test_ssm_controllers_gpu.py -q -x 2>&1 | tail -15
echo "== tail with QIT"; SRH_GUSTO_TRACE_QIT=1 timeout 600 python tools/uncapped_tail.py --top 2 2>&1 | grep -v "^library\|did not converge" | tail -12
echo "== reduce"; timeout 300 python tools/bench_reduce.py 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py > gpurun_out/r06d_bench.log 2> gpurun_out/r06d_bench.err; tail -n1 gpurun_out/r06d_bench.log | cut -c1-400
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_secondary.json'))
print('ssm_gusto_rti', d.get('ssm_gusto_rti'))
print('uncapped', {k:v for k,v in d.get('scp_uncapped_500',{}).items() if k!='cpu'})
PY
