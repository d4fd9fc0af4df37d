import sys; sys.path.insert(0,'.'); sys.path.insert(0,'soft-robot-control_amd')
import bench
from sofacontrol_amd import _lib
for i in range(4):
    r = bench.scp_c5(_lib, 0, 1, None)
    print(round(r['iterations_per_s']), round(r['ms'],1), r['iterations'])
