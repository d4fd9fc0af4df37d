"""The POD shapes of SURVEY 8(d) alone (bench.py: pod_shapes): projection r = 30 / 36 (q and full-state forms), lift, U^T M U.
SRH_LIB_PATH selects a library build (A/B of pod.hip variants).  Usage (GPU box): python tools/bench_pod_shapes.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd')]
import torch; torch.cuda.init()
import bench
from sofacontrol_amd import _lib
out = bench.pod_shapes(_lib.lib(), _lib)
for k, v in out.items():
    if isinstance(v, dict):
        print('%-16s %s' % (k, '  '.join('%s %.4g' % (a, b) for a, b in v.items() if isinstance(b, float))))
