#!/usr/bin/env python3
"""Toolchain guard: VGPR spills that are stored only while EXEC is narrowed and reloaded while it is wider.

hipcc (ROCm 7.2 LLVM) places 'Folded Spill' scratch stores of values that live across a divergent `if` at the top of the
join block BEFORE the `s_or_b64 exec, exec, s[..]` that re-enables the lanes, or inside the region; scratch stores honour
EXEC, so the lanes that sat the region out never reach the stack slot, and a reload after the region hands them whatever
the slot held before.  (Seen in round 3: the final rollout of the lean LOCP kernel loaded its stage matrices through
garbage addresses in threads 400..479 after an `if (tid < N n_u)` region -- a wrong result that came and went with
unrelated edits.)

The check looks for the placement itself: inside one basic block of the -S output, a 'Folded Spill' scratch store that
precedes an `s_or_b64 exec, exec, s[..]` (the lanes come back AFTER the store) with no EXEC write in between, and whose
stack slot is reloaded somewhere else in the function.  A store whose register is written between the matching EXEC
narrowing (`s_and_saveexec_b64 s[a:b]` / `s_mov_b64 s[a:b], exec`) and the store is left alone: that is a value the
region itself produced for its own lanes (the other side of the branch fills the rest of the slot).

Other ways EXEC grows were checked on the round-3 tree and are not flagged: in the else sequence (`s_or_saveexec_b64 d, s` ...
`s_xor_b64 exec, exec, d`) and around whole-wave sections (`s_or_saveexec_b64 d, -1` ... `s_mov_b64 exec, d`) hipcc places
its spill stores behind the widening instruction, i.e. with the lanes already enabled.

`--fix` rewrites the file: every flagged store moves to just behind the EXEC restore (the register still holds the
value in every lane there -- nothing between the two writes it, which the tool verifies), followed by the wait states a
wide store wants before its data registers may be overwritten.  The build (csrc/Makefile) compiles device code to
assembly, runs this, and assembles the result, so what ships has been through the check.  Usage: check_spill_exec.py [--fix] file.s [...]
(exit status 1 when something is flagged and not fixed, or when `audit` finds spill code in a form the patterns were not written
for -- the tool fails closed instead of reporting "0 spills" on assembly it cannot read)
"""
import re, sys

FUNC = re.compile(r'^(_Z\w+|\w+):\s*; @')
LABEL = re.compile(r'^\.LBB\w+:')
BRANCH = re.compile(r'^\s*s_(c?branch|endpgm|setpc)')
EXEC_WRITE = re.compile(r'^\s*(s_\w+saveexec_b64|s_\w+ exec,)')
POP = re.compile(r'^\s*s_or_b64 exec, exec,')
SPILL = re.compile(r'^\s*scratch_store_dword\w*\s+off, (v\[?[\d:]+\]?), off(?: offset:(\d+))?\s*; .*Folded Spill')
RELOAD = re.compile(r'^\s*scratch_load_dword\w*\s+(v\[?[\d:]+\]?), off, off(?: offset:(\d+))?\s*; .*Folded Reload')
# frames beyond the 12-bit immediate (> 4 KB of scratch per lane: the generic fused kernels): the slot address comes in an SGPR that an
# `s_movk_i32 / s_mov_b32 sK, imm` a few lines up sets.  Such stores are examined like the others (slot = imm + offset) but never MOVED by
# fix() (the pair would have to move together): moving the EXEC restore up to the join label (fix_stranded) is the repair for them.
SPILL_S = re.compile(r'^\s*scratch_store_dword\w*\s+off, (v\[?[\d:]+\]?), (s\d+)(?: offset:(\d+))?\s*; .*Folded Spill')
RELOAD_S = re.compile(r'^\s*scratch_load_dword\w*\s+(v\[?[\d:]+\]?), off, (s\d+)(?: offset:(\d+))?\s*; .*Folded Reload')


def sgpr_slot(lines, i, sreg, offset):
    """slot address of an SGPR-addressed spill access at lines[i] (0-based): the immediate last moved into sreg, within 48 lines and
    the same basic block; None when it cannot be read"""
    for j in range(i - 1, max(-1, i - 49), -1):
        line = lines[j]
        if LABEL.match(line) or BRANCH.match(line):
            return None
        m = re.match(r'^\s*s_mov(?:k_i32|_b32)\s+%s,\s*(0x[0-9a-fA-F]+|\d+)\b' % sreg, line)
        if m:
            return int(m.group(1), 0) + int(offset or 0)
        if re.match(r'^\s*[sv]_\w+\s+%s\b' % sreg, line):
            return None                              # written by something else
    return None


def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


DEST = re.compile(r'^\s*(?:v_\w+|ds_read\w*|global_load\w*|scratch_load\w*|buffer_load\w*|flat_load\w*)\s+(v\[\d+:\d+\]|v\d+)')


def defined_in_region(lines, store_ln, restore_ln, reg_tok):
    """is the stored register written between the narrowing of EXEC that `restore_ln` undoes and the store?"""
    m = re.search(r's_or_b64 exec, exec, (s\[\d+:\d+\])', lines[restore_ln - 1])
    if not m:
        return False
    saved = re.escape(m.group(1))
    opener = re.compile(r'^\s*(s_\w*saveexec_b64 %s,|s_mov_b64 %s, exec)' % (saved, saved))
    want = regs_of(reg_tok)
    for i in range(store_ln - 2, max(0, store_ln - 6000), -1):
        line = lines[i]
        if opener.match(line):
            return False
        d = DEST.match(line)
        if d and regs_of(d.group(1)) & want:
            return True
    return False


def scan(path):
    """-> [(function, slot, line of the store, line of the exec restore)]"""
    flagged, func, pending, cand, reloaded = [], None, [], [], set()
    lines = open(path).read().split('\n')

    def close():
        for f, off, ln, lr in cand:
            if off in reloaded:
                reg = (SPILL.match(lines[ln - 1]) or SPILL_S.match(lines[ln - 1])).group(1)
                if not defined_in_region(lines, ln, lr, reg):
                    flagged.append((f, off, ln, lr))

    with open(path) as fh:
        for ln, line in enumerate(fh, 1):
            m = FUNC.match(line)
            if m:
                close()
                func, pending, cand, reloaded = m.group(1), [], [], set()
                continue
            if func is None:
                continue
            if LABEL.match(line) or BRANCH.match(line):
                pending = []
                continue
            if POP.match(line):
                cand.extend((func, off, l, ln) for off, l in pending)
                pending = []
                continue
            if EXEC_WRITE.match(line):
                pending = []
                continue
            m = SPILL.match(line)
            if m:
                pending.append((int(m.group(2) or 0), ln))
                continue
            m = SPILL_S.match(line)
            if m:
                slot = sgpr_slot(lines, ln - 1, m.group(2), m.group(3))
                if slot is not None:
                    pending.append((slot, ln))
                continue
            m = RELOAD.match(line)
            if m:
                reloaded.add(int(m.group(2) or 0))
                continue
            m = RELOAD_S.match(line)
            if m:
                slot = sgpr_slot(lines, ln - 1, m.group(2), m.group(3))
                if slot is not None:
                    reloaded.add(slot)
    close()
    return flagged


def fix(path):
    """move the flagged stores behind their EXEC restore, in place; -> (moved, refused)"""
    flagged = scan(path)
    if not flagged:
        return 0, 0
    lines = open(path).read().split('\n')
    moves, refused = {}, 0
    for func, off, ln, lr in flagged:
        if not SPILL.match(lines[ln - 1]):          # SGPR-addressed: not moved (see SPILL_S)
            refused += 1
            continue
        reg = regs_of(SPILL.match(lines[ln - 1]).group(1))
        clobbered = any(DEST.match(lines[i]) and regs_of(DEST.match(lines[i]).group(1)) & reg for i in range(ln, lr))
        # the vmcnt bookkeeping of the surrounding code stays valid only if the store does not pass another vector memory
        # operation or a wait on the way
        crossed = any(re.match(r'^\s*(s_waitcnt|global_|flat_|buffer_|scratch_load)', lines[i]) or
                      (re.match(r'^\s*scratch_store', lines[i]) and not SPILL.match(lines[i])) for i in range(ln, lr - 1))
        if clobbered or crossed:
            refused += 1
            continue
        moves.setdefault(lr, []).append(ln)
    drop = {ln for lns in moves.values() for ln in lns}
    out = []
    for i, line in enumerate(lines, 1):
        if i in drop:
            continue
        out.append(line)
        if i in moves:
            for ln in moves[i]:
                out.append(lines[ln - 1].split(';')[0].rstrip() + ' ; spill moved behind the EXEC restore (check_spill_exec.py)')
            out.append('\ts_nop 1')
    open(path, 'w').write('\n'.join(out))
    return len(drop), refused


# ---- the same placement with other instructions (round 5).  The join block of a divergent region starts with the EXEC restore; hipcc
# also puts PHI copies (`v_mov_b64 v[166:167], v[154:155]` ...) of the values that live across the region IN FRONT of it.  The label is
# the target of the region's `s_cbranch_execz`: EXEC is empty there when the region was skipped or left through its loop exit and
# narrowed otherwise, so the copies reach no lane (or not all of them) and the code behind the join reads stale registers.  Seen in
# locp_kernel<*, 0, 0> after an unrelated change of the Cholesky tile code: a QP that "converged" to a 0.1 % worse minimiser.
# Looked for: at a label that some s_cbranch_execz of the function targets, vector / memory instructions before the block's first EXEC
# write when that write is `s_or_b64 exec, exec, s[a:b]`.  --fix moves the restore up to the label when everything stranded is a register
# copy or a spill access (what a join block can only mean for every lane) and nothing in between reads SCC, names EXEC or writes
# s[a:b]; anything else is refused and the build fails: a person has to look.
EXECZ = re.compile(r'^\s*s_cbranch_execz\s+(\.LBB\w+)')
ANY_INSTR = re.compile(r'^\s+([a-z][a-z0-9_]+)\b(.*)$')
ANY_EXEC_WRITE = re.compile(r'^\s*(s_\w*saveexec\w*|s_\w+\s+exec\b|v_cmpx\w*)')
HARMLESS_SCALAR = re.compile(r'^\s*(s_waitcnt|s_nop)\b')
LANE_ACCESS = re.compile(r'^\s*v_(readlane|writelane|readfirstlane)_b32\b')      # SGPR spill code: does not look at EXEC


def stranded(lines):
    """-> [(function, line of the label, line of the restore, refusal or None)]   (1-based line numbers)"""
    out, func, targets, start = [], None, set(), 0

    def close(end):
        for i in range(start, end):
            m = LABEL.match(lines[i])
            if not m or lines[i].split(':')[0] not in targets:
                continue
            body, j = [], i + 1
            while j < end:
                line = lines[j]
                if LABEL.match(line) or FUNC.match(line):
                    break
                mi = ANY_INSTR.match(line)
                if not mi or mi.group(1).startswith('.'):
                    j += 1
                    continue
                mp = re.match(r'^\s*s_or_b64 exec, exec, (s\[\d+:\d+\])', line)
                if mp:
                    real = [b for b in body if not HARMLESS_SCALAR.match(lines[b]) and not LANE_ACCESS.match(lines[b]) and not re.match(r'^\s*s_', lines[b])]
                    if real:
                        sx = mp.group(1)
                        lo, hi = (int(v) for v in re.match(r's\[(\d+):(\d+)\]', sx).groups())
                        why = None
                        for b in real:                  # only what can only have been meant for every lane is moved under the full mask
                            t = lines[b]
                            if not (re.match(r'^\s*v_mov_b(32|64)_e32\s+v', t) or re.match(r'^\s*v_accvgpr_(read|write)_b32\b', t) or SPILL.match(t) or SPILL_S.match(t) or RELOAD.match(t) or RELOAD_S.match(t)):
                                why = 'not a register copy or a spill access: ' + t.strip()[:60]
                        for b in body:
                            t = lines[b]
                            if re.match(r'^\s*s_(cselect|addc|subb|cmov|cbranch_scc)', t):
                                why = 'an instruction between the label and the restore reads SCC, which the restore writes'
                            ds = re.match(r'^\s*s_\w+\s+s(?:\[(\d+):(\d+)\]|(\d+))', t)
                            if ds:
                                a, b_ = (int(ds.group(1)), int(ds.group(2))) if ds.group(1) else (int(ds.group(3)), int(ds.group(3)))
                                if a <= hi and b_ >= lo:
                                    why = 'the saved mask is written between the label and the restore'
                            if re.search(r'\bexec\b', t.split(';')[0]):
                                why = 'an instruction between the label and the restore names EXEC'
                            d = re.match(r'^\s*v_read(?:first)?lane_b32\s+s(\d+)', t)
                            if d and lo <= int(d.group(1)) <= hi:
                                why = 'the saved mask is written between the label and the restore'
                        out.append((func, i + 1, j + 1, why))
                    break
                if ANY_EXEC_WRITE.match(line) or BRANCH.match(line) or re.match(r'^\s*s_barrier', line):
                    break
                body.append(j)
                j += 1

    for i, line in enumerate(lines):
        m = FUNC.match(line)
        if m:
            if func is not None:
                close(i)
            func, targets, start = m.group(1), set(), i
            continue
        m = EXECZ.match(line)
        if m:
            targets.add(m.group(1))
    if func is not None:
        close(len(lines))
    return out


def fix_stranded(path):
    """move the EXEC restore of every flagged join block up to its label, in place; -> (moved, refused)"""
    lines = open(path).read().split('\n')
    found = stranded(lines)
    todo = [(ll, lr) for _, ll, lr, why in found if why is None]
    if todo:
        take = {lr for _, lr in todo}
        put = {ll: lr for ll, lr in todo}
        out = []
        for i, line in enumerate(lines, 1):
            if i in take:
                continue
            out.append(line)
            if i in put:
                out.append(lines[put[i] - 1].split(';')[0].rstrip() + ' ; EXEC restore moved up to the join label (check_spill_exec.py)')
        open(path, 'w').write('\n'.join(out))
    return len(todo), len(found) - len(todo)


ANY_SPILL_STORE = re.compile(r'^\s*(scratch_store|buffer_store|flat_store|global_store)\w*\s.*;.*Folded Spill')
ANY_SPILL_MARK = re.compile(r';.*Folded (Spill|Reload)')
META_SPILLS = re.compile(r'^\s*\.vgpr_spill_count:\s*(\d+)')
WAVE32_EXEC = re.compile(r'^\s*s_\w+_b32\s+exec_lo')


def audit(path):
    """Fail closed: is this still the assembly the patterns above were written for?  -> list of complaints.
      * every spill STORE the compiler marks ('Folded Spill') must have the one form SPILL recognises (`scratch_store_dword*
        off, vN, off [offset:]`): stores addressed through an SGPR frame register (non-inlined device functions), buffer
        stores or a changed operand order would otherwise pass unexamined as "0 spills";
      * the kernels' own metadata (.vgpr_spill_count) must not report spills in a file without a single 'Folded Spill' /
        'Folded Reload' comment (a compiler that words the comment differently);
      * EXEC must be manipulated as a 64-bit register (wave64): an `s_*_b32 exec_lo` restore is not looked for at all."""
    text = open(path).read().split('\n')
    any_store = [i for i, l in enumerate(text, 1) if ANY_SPILL_STORE.match(l)]
    def known(i):
        if SPILL.match(text[i - 1]):
            return True
        m = SPILL_S.match(text[i - 1])
        return bool(m) and sgpr_slot(text, i - 1, m.group(2), m.group(3)) is not None
    unknown = [i for i in any_store if not known(i)]
    marks = sum(1 for l in text if ANY_SPILL_MARK.search(l))
    meta = sum(int(m.group(1)) for m in (META_SPILLS.match(l) for l in text) if m)
    out = []
    if unknown:
        out.append('%s: %d spill store(s) in a form this tool does not examine (first at line %d: %s)' %
                   (path, len(unknown), unknown[0], text[unknown[0] - 1].strip()[:120]))
    if meta > 0 and marks == 0:
        out.append('%s: the kernel metadata reports %d spilled VGPRs but no instruction carries a "Folded Spill" / "Folded Reload" '
                   'comment -- the compiler words its spill comments differently, nothing was examined' % (path, meta))
    w32 = [i for i, l in enumerate(text, 1) if WAVE32_EXEC.match(l)]
    if w32:
        out.append('%s: EXEC is handled as a 32-bit register (line %d): wave32 code is not examined' % (path, w32[0]))
    return out


def main(argv):
    do_fix = '--fix' in argv
    argv = [a for a in argv if a != '--fix']
    bad = 0
    for path in argv:
        for msg in audit(path):
            bad += 1
            print(msg)
        if do_fix:
            hoisted, refused = fix_stranded(path)
            if hoisted or refused:
                print('%s: %d EXEC restore(s) moved up to their join label, %d refused' % (path, hoisted, refused))
            moved, refused = fix(path)
            if moved or refused:
                print('%s: %d spill(s) moved behind their EXEC restore, %d refused' % (path, moved, refused))
        for func, off, ln, lr in scan(path):
            bad += 1
            print('%s:%d: %s: spill to stack slot %d sits before the EXEC restore at line %d' % (path, ln, func[:100], off, lr))
        for func, ll, lr, why in stranded(open(path).read().split('\n')):
            bad += 1
            print('%s:%d: %s: vector instructions between the join label and its EXEC restore at line %d%s' % (path, ll, func[:100], lr, ' (%s)' % why if why else ''))
    print('%d spill(s) / join block(s) with code under a narrowed EXEC' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
