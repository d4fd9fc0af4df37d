import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd')]
import torch; torch.cuda.init()
import bench
print(json.dumps(bench.scp_reference_horizons()['hardware_open_loop_N200']))
