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
A label resets the history CONSERVATIVELY: what precedes it in the text counts as the predecessor, and a DPP instruction
closer than two instructions behind a label that is a branch target is given its wait states inside its own block.
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
    hist = []                       # per issued instruction since the block start: (wait states it stands for, vgprs written by VALU, is transcendental, in asm)
    func = None
    for ln, raw in enumerate(lines, 1):
        line = raw.rstrip('\n')
        if FUNC.match(line):
            func, hist, in_asm = line.split(':')[0], [], False
        if ';;#ASMSTART' in line:
            in_asm = True
        elif ';;#ASMEND' in line:
            in_asm = False
        if LABEL.match(line):
            hist = [(0, set(), False, False, 'label')]
        m = INSTR.match(line)
        if not m or m.group(1).startswith('.') or func is None:
            out.append(raw)
            continue
        op, ops = m.group(1), m.group(2)
        if op == 's_nop':
            n = int(ops.split()[0], 0) + 1 if ops else 1
            hist.append((n, set(), False, in_asm, op))
            out.append(raw)
            continue
        is_valu = op.startswith('v_')
        dst_txt, src_txt = split_operands(ops)
        if is_valu and op.startswith('v_fmac'):
            src_txt = ops                         # the destination is read as well
        need = 0
        if is_valu:
            srcs = vregs(src_txt)
            dpp_here = in_asm and DPP_CTRL.search(ops) is not None
            dist = 0                              # wait states between a producer and this instruction
            at_label = False
            for ws, wr, trans, asm_p, pop in reversed(hist):
                if pop == 'label':
                    at_label = True
                    break
                if dpp_here and wr & srcs and dist < 2:
                    need = max(need, 2 - dist)
                if trans and (asm_p or in_asm) and wr & srcs and dist < 1 and not TRANS.match(op):
                    need = max(need, 1 - dist)
                dist += ws
                if dist >= 2:
                    break
            if dpp_here and at_label and dist < 2:
                need = max(need, 2 - dist)        # the block before a branch target is unknown: wait inside this block
        if need:
            findings.append((func, ln, op + ' ' + ops, need))
            if fix:
                out.append('\ts_nop %d\n' % (need - 1))
                hist.append((need, set(), False, in_asm, 's_nop'))
        wr = set()
        if is_valu and not NO_VDST.match(op):
            wr = vregs(dst_txt)
        hist.append((1, wr, bool(TRANS.match(op)), in_asm, op))
        if len(hist) > 8:
            hist = hist[-8:]
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
