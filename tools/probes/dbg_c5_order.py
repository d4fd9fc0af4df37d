"""Why does secondary.scp_c5 take 43 ms inside bench.py and 14 ms alone?  The secondary sequence, then scp_c5 three more times."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd')]
import torch; torch.cuda.init()
import bench
from sofacontrol_amd import _lib
L = _lib.lib()
if 'first' in sys.argv:
    for i in range(3):
        r = bench.scp_c5(_lib, 0, 1, None); print('alone', round(r['ms'], 2), r['rollouts_handed_to_fused_kernel'])
sec = bench.secondary(L, _lib, 0, 1, None)
print('in secondary', sec['scp_c5']['ms'], sec['scp_c5']['ms_all_calls'], sec['scp_c5'].get('rollouts_handed_to_fused_kernel'), sec['scp_c5_32_rollouts']['ms'])
for i in range(3):
    r = bench.scp_c5(_lib, 0, 1, None); print('after', round(r['ms'], 2), r['rollouts_handed_to_fused_kernel'])
