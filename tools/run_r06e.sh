#!/bin/bash
# round 6, call e: SSM GuSTO latency (device / device without lean / host loop), the price of streaming G from L2 (SRH_LEAN_J0)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd $GRAFT_REPO_ROOT
echo "== ssm tests"; timeout 900 python -m pytest tests/test_ssm_gpu.py tests/test_ssm_controllers_gpu.py -q -x 2>&1 | tail -5
echo "== ssm rti"; timeout 300 python tools/time_ssm_rti.py 2>&1 | tail -1
SRH_GUSTO_SSM_NO_LEAN=1 timeout 300 python tools/time_ssm_rti.py 2>&1 | tail -1
SRH_GUSTO_SSM_HOST_LOOP=1 timeout 300 python tools/time_ssm_rti.py 2>&1 | tail -1
timeout 300 python tools/time_ssm_rti.py --batch 256 2>&1 | tail -1
echo "== lean: fixed layout (product)"; timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
echo "== lean: run-time layout, j0 = 7 (61 of 82 KB of G in LDS)"; SRH_LEAN_NO_FIXED=1 timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
echo "== lean: run-time layout, j0 = 49 (G streamed from L2)"; SRH_LEAN_NO_FIXED=1 SRH_LEAN_J0=49 timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
echo "== lean: run-time layout, j0 = 25"; SRH_LEAN_NO_FIXED=1 SRH_LEAN_J0=25 timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
