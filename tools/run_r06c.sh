#!/bin/bash
# round 6, call c: warm-started full (trust-region-active) QPs A/B on the uncapped tail; K-split sweep of the batched U^T M U
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd $GRAFT_REPO_ROOT
O=gpurun_out
echo "== uncapped tail, warm full QPs (SRH_GUSTO_WARM_FULL=1)"; SRH_GUSTO_WARM_FULL=1 timeout 600 python tools/uncapped_tail.py --top 4 2>&1 | grep -v "^library" | tail -6
echo "== uncapped tail, cold full QPs (default)"; timeout 600 python tools/uncapped_tail.py --top 4 2>&1 | grep -v "^library" | tail -6
echo "== binding trust region test"; timeout 900 python -m pytest tests/test_gusto_bench_shapes_gpu.py tests/test_gusto_gpu.py tests/test_locp_gpu.py -q -x 2>&1 | tail -3
for ks in 1 2 3 4 6 13; do echo "== SRH_UTMU_KSPLIT=$ks"; SRH_UTMU_KSPLIT=$ks timeout 300 python tools/bench_reduce.py 2>&1 | tail -2; done
