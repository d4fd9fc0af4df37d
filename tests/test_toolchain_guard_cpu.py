"""The toolchain guard of the build (tools/check_spill_exec.py): VGPR spills and PHI copies that hipcc places in front of the EXEC
restore of a join block are found, repaired, and absent from the device assembly the shipped objects were assembled from; and
tools/check_dpp_hazard.py: the wait states in front of inline-asm DPP reads that the compiler's hazard recogniser does not see."""
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
    """the spill mover refuses to carry a store over a wait; the whole tool (round 5) moves the EXEC restore up to the join label instead"""
    p = write(tmp_path, BAD.replace('	v_writelane_b32 v250, s47, 37\n', '	s_waitcnt vmcnt(0)\n'))
    assert guard.fix(p) == (0, 1) and guard.main(['--fix', p]) == 0
    lines = open(p).read().split('\n')
    lab = lines.index('.LBB0_2:')
    assert lines[lab + 1].startswith('\ts_or_b64 exec, exec, s[0:1]') and guard.scan(p) == []


# round 5: the copies of values that live across a divergent loop, placed at the join label IN FRONT of the EXEC restore (EXEC is empty
# there: the loop ends when its last lane leaves, the branch around it is taken when no lane enters) -- the registers keep their old
# contents and the code behind the join reads them (locp_kernel<*, 0, 0>: a QP that converged to a worse minimiser)
STRANDED = HEAD + '''	s_and_saveexec_b64 s[2:3], s[12:13]
	s_cbranch_execz .LBB0_3
; %bb.1:
	s_mov_b64 s[20:21], 0
.LBB0_2:
	v_add_u32_e32 v3, s12, v3
	v_cmp_le_i32_e32 vcc, s4, v3
	s_or_b64 s[20:21], vcc, s[20:21]
	s_andn2_b64 exec, exec, s[20:21]
	s_cbranch_execnz .LBB0_2
.LBB0_3:
	v_mov_b64_e32 v[166:167], v[154:155]
	v_readlane_b32 s8, v250, 23
	v_mov_b32_e32 v0, v139
	s_or_b64 exec, exec, s[2:3]
	v_mov_b64_e32 v[130:131], v[108:109]
	s_endpgm
'''


def test_copies_in_front_of_the_exec_restore_of_a_join_are_flagged_and_repaired(tmp_path):
    p = write(tmp_path, STRANDED)
    found = guard.stranded(open(p).read().split('\n'))
    assert [(f[1], f[2], f[3]) for f in found] == [(12, 16, None)]
    assert guard.main([p]) == 1
    assert guard.main(['--fix', p]) == 0
    lines = open(p).read().split('\n')
    lab = lines.index('.LBB0_3:')
    assert lines[lab + 1].startswith('\ts_or_b64 exec, exec, s[2:3]') and 'v_mov_b64_e32 v[166:167]' in lines[lab + 2]
    assert sum('s_or_b64 exec, exec' in l for l in lines) == 1 and guard.stranded(lines) == []
    # lane accesses (SGPR spill code) do not look at EXEC: a join block that only has those in front of its restore is left alone
    only_lanes = STRANDED.replace('\tv_mov_b64_e32 v[166:167], v[154:155]\n', '').replace('\tv_mov_b32_e32 v0, v139\n', '')
    q = write(tmp_path, only_lanes)
    assert guard.stranded(only_lanes.split('\n')) == [] and guard.main(['--fix', q]) == 0 and open(q).read() == only_lanes
    # the saved mask reloaded between the label and the restore: the restore cannot move over it -- refused, the build fails
    reload_mask = STRANDED.replace('v_readlane_b32 s8, v250, 23', 'v_readlane_b32 s2, v250, 23')
    r = write(tmp_path, reload_mask)
    assert guard.stranded(reload_mask.split('\n'))[0][3] is not None and guard.main(['--fix', r]) == 1


def test_spill_stores_addressed_through_an_sgpr_are_read(tmp_path):
    """frames beyond 4 KB per lane (the generic fused kernels): `s_movk_i32 sK, imm` + `scratch_store ... off, v, sK`"""
    big = BAD.replace('	scratch_store_dword off, v74, off offset:536 ; 4-byte Folded Spill',
                      '	s_movk_i32 s3, 0x1010\n	v_readlane_b32 s5, v246, 22\n	scratch_store_dword off, v74, s3 ; 4-byte Folded Spill')
    big = big.replace('	scratch_load_dword v5, off, off offset:536 ; 4-byte Folded Reload', '	s_movk_i32 s0, 0x1010\n	scratch_load_dword v5, off, s0 ; 4-byte Folded Reload')
    p = write(tmp_path, big)
    assert guard.audit(p) == [] and [(h[1]) for h in guard.scan(p)] == [0x1010]
    assert guard.fix(p) == (0, 1)                          # never moved on its own ...
    assert guard.main(['--fix', p]) == 0                   # ... the EXEC restore moves up to the join label instead
    lines = open(p).read().split('\n')
    assert lines[lines.index('.LBB0_2:') + 1].startswith('\ts_or_b64 exec, exec, s[0:1]') and guard.scan(p) == []
    # an address register the tool cannot follow (set in another block): not examined -> the audit says so
    lost = big.replace('	s_movk_i32 s3, 0x1010\n', '')
    assert len(guard.audit(write(tmp_path, lost))) == 1


def _dpp_tool():
    sp = importlib.util.spec_from_file_location('check_dpp_hazard', os.path.join(ROOT, 'tools', 'check_dpp_hazard.py'))
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    return m


def test_wait_states_in_front_of_inline_asm_dpp_reads(tmp_path):
    """VALU write -> DPP read of the same VGPR needs two wait states; the compiler does not see a DPP operand inside inline asm."""
    dpp = _dpp_tool()
    body = HEAD + '''	v_mul_f64 v[32:33], v[36:37], v[4:5]
	;;#ASMSTART
	v_fmac_f64_dpp v[38:39], -v[32:33], v[32:33] row_newbcast:1 row_mask:0xf bank_mask:0xf
	;;#ASMEND
	v_mul_f64 v[30:31], v[94:95], v[4:5]
	;;#ASMSTART
	v_mov_b64_dpp v[2:3], v[38:39] row_newbcast:1 row_mask:0xf bank_mask:0xf
	;;#ASMEND
	v_rsq_f64_e32 v[36:37], v[2:3]
	;;#ASMSTART
	v_mul_f64 v[40:41], v[36:37], v[36:37]
	;;#ASMEND
	v_mov_b32_dpp v7, v30 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
	s_endpgm
'''
    p = write(tmp_path, body)
    found, _ = dpp.scan(open(p).read().splitlines(True), False)
    # the fmac right behind the producer of its DPP operand (2), the mov one instruction behind the fmac that wrote v[38:39] (1), the
    # asm consumer of a transcendental result (1); the compiler's own DPP move outside asm is the compiler's business
    assert [(f[2].split()[0], f[3]) for f in found] == [('v_fmac_f64_dpp', 2), ('v_mov_b64_dpp', 1), ('v_mul_f64', 1)]
    assert dpp.main([p]) == 1 and dpp.main(['--fix', p]) == 0 and dpp.main([p]) == 0
    text = open(p).read()
    assert text.count('s_nop 1') == 1 and text.count('s_nop 0') == 2


def test_exec_write_mfma_result_and_label_rules_of_the_dpp_guard(tmp_path):
    """The three orderings the round-5 advice found unchecked: (1) a VALU write of EXEC (v_cmpx) followed by an asm DPP op needs FIVE
    wait states; (2) an MFMA / DGEMM result read by an asm VALU instruction needs the MAI -> VALU wait states the compiler applies to
    its own instructions only (the tool asks for 19 behind the f64 16x16x4, 8 behind the f64 4x4x4); (3) behind a label the other
    predecessors of the block are unknown: an asm consumer of a possible transcendental result waits inside its own block, and the
    fall-through predecessor in the text is still examined.  --fix pads every one of them; the re-scan is clean."""
    dpp = _dpp_tool()
    body = HEAD + '''	v_cmpx_gt_f64_e32 v[10:11], v[12:13]
	v_mov_b32_e32 v20, v21
	;;#ASMSTART
	v_mov_b64_dpp v[2:3], v[38:39] row_newbcast:1 row_mask:0xf bank_mask:0xf
	;;#ASMEND
	s_nop 7
	s_nop 7
	s_nop 7
	v_mfma_f64_16x16x4_f64 v[40:47], v[50:51], v[52:53], v[40:47]
	v_mov_b32_e32 v22, v23
	v_mov_b32_e32 v24, v25
	;;#ASMSTART
	v_mul_f64 v[60:61], v[42:43], v[42:43]
	;;#ASMEND
	s_nop 7
	s_nop 7
	s_nop 7
	v_mfma_f64_4x4x4_4b_f64 v[70:71], v[50:51], v[52:53], v[70:71]
	;;#ASMSTART
	v_add_f64 v[70:71], v[62:63], v[62:63]
	;;#ASMEND
	s_nop 7
	s_nop 7
	s_nop 7
	v_rsq_f64_e32 v[36:37], v[2:3]
.LBB0_7:
	;;#ASMSTART
	v_mul_f64 v[64:65], v[36:37], v[36:37]
	;;#ASMEND
	s_endpgm
'''
    p = write(tmp_path, body)
    found, _ = dpp.scan(open(p).read().splitlines(True), False)
    got = [(f[2].split()[0], f[3]) for f in found]
    # (1) one instruction stands between the v_cmpx and the DPP op: 4 of the 5 wait states are missing; (2) two instructions behind the
    # 16x16x4: 17 of 19; the asm write of the 4x4x4's destination right behind it: 8; (3) the asm multiply at the label: the fall-through
    # predecessor IS the transcendental (1)
    assert got == [('v_mov_b64_dpp', 4), ('v_mul_f64', 17), ('v_add_f64', 8), ('v_mul_f64', 1)], got
    assert dpp.main([p]) == 1 and dpp.main(['--fix', p]) == 0 and dpp.main([p]) == 0
    # a label alone (no transcendental in the text before it) still makes an asm VALU instruction wait: unknown predecessors
    body2 = HEAD + '''	v_mov_b32_e32 v22, v23
.LBB0_9:
	;;#ASMSTART
	v_mul_f64 v[64:65], v[36:37], v[36:37]
	;;#ASMEND
	s_endpgm
'''
    p2 = write(tmp_path, body2)
    found2, _ = dpp.scan(open(p2).read().splitlines(True), False)
    assert [(f[2].split()[0], f[3]) for f in found2] == [('v_mul_f64', 1)]


def test_shipped_device_assembly_is_clean():
    units = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    asms = [u[:-4] + '.s' for u in units]
    lib = os.path.join(ROOT, 'soft-robot-control_amd', 'sofacontrol_amd', 'libsofacontrol_hip.so')
    if not os.path.exists(lib) or not all(os.path.exists(a) for a in asms):
        pytest.skip('library not built here (the assembly is kept next to the objects by the Makefile)')
    dpp = _dpp_tool()
    for a in asms:
        assert guard.scan(a) == [], a
        assert guard.stranded(open(a).read().split('\n')) == [], a
        assert guard.audit(a) == [], a          # every spill store has the form the check examines; metadata and comments agree
        assert dpp.scan(open(a).read().splitlines(True), False)[0] == [], a


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
