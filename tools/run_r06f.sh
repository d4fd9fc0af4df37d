#!/bin/bash
# round 6, call f: SSM GuSTO kernel with the model tables in LDS
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd $GRAFT_REPO_ROOT
echo "== ssm tests"; timeout 900 python -m pytest tests/test_ssm_gpu.py tests/test_ssm_controllers_gpu.py -q -x 2>&1 | tail -5
echo "== ssm rti"; timeout 300 python tools/time_ssm_rti.py 2>&1 | tail -1
SRH_GUSTO_SSM_NO_TABLES=1 timeout 300 python tools/time_ssm_rti.py 2>&1 | tail -1
timeout 300 python tools/time_ssm_rti.py --batch 256 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06f_trace -o ssm -- python3 $GRAFT_REPO_ROOT/tools/time_ssm_rti.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r06f_trace -name "*kernel_stats.csv" | head -1); head -5 $f
