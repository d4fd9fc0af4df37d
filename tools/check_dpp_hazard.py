#!/usr/bin/env python3
"""Toolchain guard: wait states around hand-written DPP / transcendental instructions (inline asm) on gfx950.

The compiler's hazard recogniser does not look inside inline asm: an `asm("v_fmac_f64_dpp ...")` that reads a VGPR through
the DPP network needs TWO wait states after the VALU instruction that wrote that VGPR (gfx9 / CDNA "VALU writes VGPR -> DPP
reads that VGPR"), a non-transcendental VALU instruction that reads the result of a transcendental one (v_rsq_f64 ...)
needs ONE (gfx940+ forwarding hazard), and nothing inserts them when either end of the pair sits in an asm statement.
The kernels use such statements where a broadcast operand folded into the FMA replaces a v_readlane pair + s_nop + FMA
(csrc/locp_cond.h: qpc::chol16_dpp); the asm statements themselves carry no s_nop -- this tool places them where the FINAL
instruction order needs them, and only there.

For every function of a `-S` output, in program order inside a basic block:
  * an instruction with a DPP control (`row_newbcast`, `row_shr`, `quad_perm` ... ) inside an `;;#ASMSTART` region: none of its
    VGPR sources may have been written by a VALU instruction less than 2 wait states before;
  * any VALU instruction that reads a VGPR written by a transcendental instruction less than 1 wait state before, when one of
    the two is inside an asm region (both compiler-made: the compiler has already dealt with it).
  * a VALU instruction that WRITES EXEC (v_cmpx_*; any VALU with `exec` as its destination) followed by an asm DPP instruction:
    FIVE wait states (gfx9 / CDNA "VALU writes EXEC -> VALU DPP op");
  * an MFMA / DGEMM result (v_mfma_*, v_smfmac_*, v_dot*) read -- or overwritten -- by a VALU instruction inside an asm region:
    the MAI -> VALU wait states are applied by the compiler to ITS instructions only.  The tool asks for the largest figure of
    LLVM's hazard recogniser for the instruction class on gfx90a / gfx940 / gfx950 (f64 16x16x4: 19, f64 4x4x4: 8, any other
    MFMA: 19) -- nops in a place that never occurs in the shipped kernels cost nothing, a short count would.
A label does not reset the history: what precedes it in the text is the fall-through predecessor and is still examined; the
OTHER predecessors of a branch target are unknown, so a consumer closer to the label than its rule's distance is given the
missing wait states inside its own block (DPP: 2; EXEC -> DPP: 5, only in functions that contain a VALU write of EXEC at
all; asm VALU behind a possible transcendental: 1; asm VALU behind a possible MFMA: the full distance, but only when some
MFMA of the same function writes a register the asm instruction touches -- otherwise no path can carry the hazard).  A case `--fix` cannot pad (it never gives up on one: an s_nop can always be placed in
front of the consumer) would be reported by the re-scan and fails the build.
`--fix` inserts the missing `s_nop`; without it the tool lists the places and exits 1.  Usage: check_dpp_hazard.py [--fix] file.s [...]
"""
import re
import sys

FUNC = re.compile(r'^(_Z\w+|\w+):\s*; @')
LABEL = re.compile(r'^\.LBB\w+:')
INSTR = re.compile(r'^\s+([a-z][a-z0-9_]+)\s*(.*?)\s*(?:;.*)?$')
DPP_CTRL = re.compile(r'\b(row_newbcast|row_shl|row_shr|row_ror|row_mirror|row_half_mirror|row_bcast|wave_shl|wave_shr|wave_rol|wave_ror|quad_perm|row_share|row_xmask)\b')
TRANS = re.compile(r'^v_(rsq|rcp|sqrt|exp|log|sin|cos|rcp_iflag|rsq_clamp)_')
VREG = re.compile(r'v\[(\d+):(\d+)\]|\bv(\d+)\b')
NO_VDST = re.compile(r'^v_(cmp|cmpx|readlane|readfirstlane|nop)')
MAI = re.compile(r'^v_(mfma|smfmac|dot)')
HIST = 24                           # instructions of history kept: the longest rule looks 19 wait states back


def mai_wait(op):
    if op.startswith('v_mfma_f64_4x4x4'):
        return 8
    return 19


def writes_exec(op, dst_txt):
    return op.startswith('v_cmpx') or (op.startswith('v_') and re.search(r'\bexec(_lo|_hi)?\b', dst_txt) is not None)


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_operands(ops):
    """destination text, source text (first comma at depth 0 splits them)"""
    depth = 0
    for i, ch in enumerate(ops):
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        elif ch == ',' and depth == 0:
            return ops[:i], ops[i + 1:]
    return ops, ''


def scan(lines, fix):
    """returns (findings, new_lines)"""
    findings, out = [], []
    in_asm = False
    # per issued instruction of the function, oldest first: (wait states it stands for, vgprs written by VALU, kind, in asm, opcode)
    # kind: '' plain, 'trans', 'exec' (VALU write of EXEC), 'mai', 'label'
    hist = []
    func = None
    func_has_vexec = {}
    func_mai_dst = {}
    cur = None
    for raw in lines:                 # first pass: which functions contain a VALU write of EXEC at all
        line = raw.rstrip('\n')
        if FUNC.match(line):
            cur = line.split(':')[0]
            func_has_vexec[cur] = False
            func_mai_dst[cur] = set()
        m = INSTR.match(line)
        if m and cur is not None and m.group(1).startswith('v_'):
            d, _ = split_operands(m.group(2))
            if writes_exec(m.group(1), d):
                func_has_vexec[cur] = True
            if MAI.match(m.group(1)):
                func_mai_dst[cur] |= vregs(d)
    for ln, raw in enumerate(lines, 1):
        line = raw.rstrip('\n')
        if FUNC.match(line):
            func, hist, in_asm = line.split(':')[0], [], False
        if ';;#ASMSTART' in line:
            in_asm = True
        elif ';;#ASMEND' in line:
            in_asm = False
        if LABEL.match(line):
            hist.append((0, set(), 'label', False, 'label'))
        m = INSTR.match(line)
        if not m or m.group(1).startswith('.') or func is None:
            out.append(raw)
            continue
        op, ops = m.group(1), m.group(2)
        if op == 's_nop':
            n = int(ops.split()[0], 0) + 1 if ops else 1
            hist.append((n, set(), '', in_asm, op))
            out.append(raw)
            continue
        is_valu = op.startswith('v_')
        dst_txt, src_txt = split_operands(ops)
        if is_valu and op.startswith('v_fmac'):
            src_txt = ops                         # the destination is read as well
        need = 0
        if is_valu:
            srcs = vregs(src_txt)
            touched = srcs | (vregs(dst_txt) if not NO_VDST.match(op) else set())
            dpp_here = in_asm and DPP_CTRL.search(ops) is not None
            dist = 0                              # wait states between a producer and this instruction
            label_dist = None                     # wait states between the nearest label in front and this instruction
            for ws, wr, kind, asm_p, pop in reversed(hist):
                if kind == 'label':
                    if label_dist is None:
                        label_dist = dist
                    continue
                if dpp_here and wr & srcs and dist < 2:
                    need = max(need, 2 - dist)
                if dpp_here and kind == 'exec' and dist < 5:
                    need = max(need, 5 - dist)
                if kind == 'trans' and (asm_p or in_asm) and wr & srcs and dist < 1 and not TRANS.match(op):
                    need = max(need, 1 - dist)
                if kind == 'mai' and in_asm and wr & touched and dist < mai_wait(pop):
                    need = max(need, mai_wait(pop) - dist)
                dist += ws
                if dist >= 19:
                    break
            if label_dist is not None:
                # the other predecessors of a branch target are unknown: wait inside this block
                if dpp_here and label_dist < 2:
                    need = max(need, 2 - label_dist)
                if dpp_here and func_has_vexec.get(func) and label_dist < 5:
                    need = max(need, 5 - label_dist)
                if in_asm and not TRANS.match(op) and label_dist < 1:
                    need = max(need, 1)
                if in_asm and label_dist < 19 and touched & func_mai_dst.get(func, set()):
                    need = max(need, 19 - label_dist)
        if need:
            findings.append((func, ln, op + ' ' + ops, need))
            if fix:
                out.append('\ts_nop %d\n' % (min(need, 16) - 1))
                hist.append((min(need, 16), set(), '', in_asm, 's_nop'))
                if need > 16:
                    out.append('\ts_nop %d\n' % (need - 16 - 1))
                    hist.append((need - 16, set(), '', in_asm, 's_nop'))
        wr = set()
        if is_valu and not NO_VDST.match(op):
            wr = vregs(dst_txt)
        kind = 'trans' if TRANS.match(op) else ('exec' if is_valu and writes_exec(op, dst_txt) else ('mai' if MAI.match(op) else ''))
        hist.append((1, wr, kind, in_asm, op))
        if len(hist) > HIST:
            hist = hist[-HIST:]
        out.append(raw)
    return findings, out


def main(argv):
    fix = '--fix' in argv
    files = [a for a in argv if not a.startswith('--')]
    bad = 0
    for path in files:
        with open(path) as f:
            lines = f.readlines()
        findings, new = scan(lines, fix)
        if findings and fix:
            with open(path, 'w') as f:
                f.writelines(new)
            again, _ = scan(new, False)
            if again:
                print('%s: %d places still short of wait states after --fix' % (path, len(again)))
                bad += len(again)
            print('%s: %d wait-state gaps in front of asm DPP / transcendental consumers filled' % (path, len(findings)))
        elif findings:
            for func, ln, text, need in findings[:20]:
                print('%s:%d: %s: %d wait state(s) missing in front of `%s`' % (path, ln, func, need, text))
            bad += len(findings)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
