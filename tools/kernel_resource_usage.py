"""profiles/<tag>_kernel_resource_usage.txt from the device assembly the build keeps next to every object (csrc/<unit>.s,
tools/hipcc_guarded.sh): the .amdhsa metadata of every kernel -- registers, scratch, spills, LDS -- one line per instantiation.
Usage: python3 tools/kernel_resource_usage.py r06"""
import glob, os, re, subprocess, sys
if len(sys.argv) < 2:
    raise SystemExit('usage: python3 tools/kernel_resource_usage.py <round tag, e.g. r06>  (no default: a default once overwrote an older round\'s file)')
TAG = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = ['# .amdhsa kernel metadata of soft-robot-control_amd/csrc/*.s (hipcc -O3 --offload-arch=gfx950, ROCm 7.2), %s tree; one line per kernel instantiation' % TAG,
       '# lean kernels: <n_u, n_x, GX lanes per stage (0: general rows), fixed horizon (0: run time), first resident stage, state rows>',
       '# unit | kernel | VGPRs | AGPRs | scratch bytes/lane | SGPR spills | VGPR spills | SGPRs | static LDS bytes | max workgroup']
pat = re.compile(r'- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.max_flat_workgroup_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)'
                 r'.*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', re.S)
for f in sorted(glob.glob(os.path.join(R, 'soft-robot-control_amd', 'csrc', '*.s'))):
    s = open(f).read()
    for m in pat.finditer(s):
        agpr, lds, wg, name, scratch, sgpr, sspill, vgpr, vspill = m.groups()
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
        out.append(' | '.join([os.path.basename(f).replace('.s', '.hip'), dem, vgpr, agpr, scratch, sspill, vspill, sgpr, lds, wg]))
open(os.path.join(R, 'profiles', TAG + '_kernel_resource_usage.txt'), 'w').write('\n'.join(out) + '\n')
print(len(out) - 3, 'kernels')
