"""The toolchain guard of the build (tools/check_spill_exec.py): VGPR spills that hipcc places in front of the EXEC restore
of a join block are found, repaired, and absent from the device assembly the shipped objects were assembled from."""
import glob
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'soft-robot-control_amd', 'csrc')

spec = importlib.util.spec_from_file_location('check_spill_exec', os.path.join(ROOT, 'tools', 'check_spill_exec.py'))
guard = importlib.util.module_from_spec(spec)
spec.loader.exec_module(guard)

HEAD = '_Z6kernelv:                             ; @_Z6kernelv\n'
# a value defined before a divergent region (v74), spilled at the top of the join block while only the lanes of the region
# are enabled, reloaded later for every lane: the pattern that produced wrong states in round 3
BAD = HEAD + '''	v_mov_b32_e32 v74, v0
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_u32_e32 v3, 1, v3
.LBB0_2:
	v_writelane_b32 v250, s46, 36
	scratch_store_dword off, v74, off offset:536 ; 4-byte Folded Spill
	v_writelane_b32 v250, s47, 37
	s_or_b64 exec, exec, s[0:1]
	v_mov_b32_e32 v74, 0
	scratch_load_dword v5, off, off offset:536 ; 4-byte Folded Reload
	s_endpgm
'''
# the same placement for a value the region itself produced for its own lanes: legitimate (the other path fills the rest)
OWN = HEAD + '''	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_u32_e32 v74, 1, v3
	scratch_store_dword off, v74, off offset:536 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	scratch_load_dword v5, off, off offset:536 ; 4-byte Folded Reload
	s_endpgm
'''


def write(tmp_path, text):
    p = tmp_path / 'unit.s'
    p.write_text(text)
    return str(p)


def test_live_through_spill_before_exec_restore_is_flagged_and_repaired(tmp_path):
    p = write(tmp_path, BAD)
    hits = guard.scan(p)
    assert [(h[1], h[2]) for h in hits] == [(536, 9)]
    assert guard.main([p]) == 1
    assert guard.fix(p) == (1, 0)
    assert guard.scan(p) == []
    lines = open(p).read().split('\n')
    restore = next(i for i, l in enumerate(lines) if 's_or_b64 exec, exec' in l)
    assert 'scratch_store_dword off, v74, off offset:536' in lines[restore + 1] and lines[restore + 2].strip() == 's_nop 1'
    assert sum('scratch_store_dword' in l for l in lines) == 1
    assert guard.main(['--fix', p]) == 0


def test_value_produced_inside_the_region_is_left_alone(tmp_path):
    p = write(tmp_path, OWN)
    assert guard.scan(p) == [] and guard.fix(p) == (0, 0) and open(p).read() == OWN


def test_store_that_would_pass_a_wait_is_refused(tmp_path):
    p = write(tmp_path, BAD.replace('	v_writelane_b32 v250, s47, 37\n', '	s_waitcnt vmcnt(0)\n'))
    assert guard.fix(p) == (0, 1) and guard.main(['--fix', p]) == 1


def test_shipped_device_assembly_is_clean():
    units = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    asms = [u[:-4] + '.s' for u in units]
    lib = os.path.join(ROOT, 'soft-robot-control_amd', 'sofacontrol_amd', 'libsofacontrol_hip.so')
    if not os.path.exists(lib) or not all(os.path.exists(a) for a in asms):
        pytest.skip('library not built here (the assembly is kept next to the objects by the Makefile)')
    for a in asms:
        assert guard.scan(a) == [], a
        assert guard.audit(a) == [], a          # every spill store has the form the check examines; metadata and comments agree


def test_audit_fails_closed_on_spill_code_it_cannot_read(tmp_path):
    """Spill stores addressed through an SGPR frame register (non-inlined device functions), spilled kernels whose comments
    are worded differently, and wave32 EXEC handling are NOT what the patterns were written for: the tool must say so and exit
    non-zero instead of reporting '0 spills'."""
    framed = HEAD + '\tscratch_store_dword off, v74, s33 offset:536 ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[0:1]\n\ts_endpgm\n'
    assert guard.scan(write(tmp_path, framed)) == [] and len(guard.audit(write(tmp_path, framed))) == 1
    assert guard.main([write(tmp_path, framed)]) == 1
    reworded = HEAD + '\tscratch_store_dword off, v74, off offset:536 ; 4-byte spill\n\ts_endpgm\n    .vgpr_spill_count: 3\n'
    assert any('metadata' in m for m in guard.audit(write(tmp_path, reworded)))
    w32 = HEAD + '\ts_or_b32 exec_lo, exec_lo, s0\n\ts_endpgm\n'
    assert any('wave32' in m for m in guard.audit(write(tmp_path, w32)))
    assert guard.audit(write(tmp_path, BAD)) == [] and guard.audit(write(tmp_path, OWN)) == []


def test_scanner_reads_what_the_installed_compiler_emits(tmp_path):
    """A kernel forced to spill, compiled with the installed hipcc: its assembly must pass the audit (the spill stores have
    the examined form and carry the 'Folded Spill' comment) -- a compiler update that changes either shows up here."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc here')
    src = tmp_path / 'spill.hip'
    src.write_text('''#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void k(double *p, int n) {
    double a[96];
    for (int i = 0; i < 96; ++i) a[i] = p[threadIdx.x + 1024 * i];
    __syncthreads();
    for (int r = 0; r < n; ++r) for (int i = 0; i < 96; ++i) a[i] = fma(a[i], a[(i + r) % 96], p[i]);
    for (int i = 0; i < 96; ++i) p[threadIdx.x + 1024 * i] = a[i];
}
''')
    out = tmp_path / 'spill.s'
    r = subprocess.run([hipcc, '-O3', '--offload-arch=gfx950', '--cuda-device-only', '-S', str(src), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    assert guard.audit(str(out)) == []
    if 'Folded Spill' in text:                      # it did spill: the stores must be in the examined form
        assert any(guard.SPILL.match(l) for l in text.split('\n'))
