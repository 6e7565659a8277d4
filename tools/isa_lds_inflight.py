#!/usr/bin/env python3
"""Reads of registers whose LDS load may still be in flight, found in the compiler's ISA.

The conv kernels issue their `ds_read_b128` through inline asm and wait with hand-counted `s_waitcnt lgkmcnt(N)` (LDS reads return in order: after
lgkmcnt(N) everything but the youngest N has landed). The compiler believes an asm output is written where the asm stands, so a count that is too
generous is invisible to it -- and to every single-process test, because an LDS read lands in ~100 cycles. This tool replays the counts: per basic
block (labels and branches clear the state: only what one straight-line stretch proves wrong is reported, no false alarms from control flow) it keeps
the queue of outstanding LGKM operations and reports every instruction that reads, or overwrites, a register a queued `ds_read` has not yet delivered.
Scalar loads sit in the same counter and return out of order: while one is queued, a wait with N > 0 proves nothing and is treated so.

usage: isa_lds_inflight.py file.s [substring-of-kernel-name]   -> one line per finding, then 'findings: n'"""
import re
import sys

_LABEL = re.compile(r'^(\S+):')
_BRANCH = ('s_branch', 's_cbranch', 's_endpgm', 's_setpc', 's_swappc', 's_call')
_NOOPERAND = ('s_barrier', 's_nop', 's_sleep', 's_setprio', 's_waitcnt_', 's_sendmsg', 's_setreg', 's_getreg', 's_memtime', 's_memrealtime')


def _regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def findings(asm, only=None):
    """-> list of (kernel, line number, instruction, line of the ds_read still in flight)."""
    out = []
    fn = None
    pend = []                     # oldest first: (registers or None for a scalar load / LDS op without a register result, line)
    for ln, raw in enumerate(asm.splitlines(), 1):
        s = raw.strip()
        m = _LABEL.match(s)
        if m:
            name = m.group(1)
            if not name.startswith('.'):
                fn = name
            pend = []
            continue
        if not s or s[0] in ';.':
            continue
        s = s.split(';')[0].strip()
        p = s.replace(',', ' ').split()
        if not p or fn is None or (only and only not in fn):
            continue
        op = p[0]
        if op.startswith(_BRANCH):
            pend = []
            continue
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', s)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pend = []
                elif not any(r is None for r, _ in pend):          # a scalar load in the queue: counts prove nothing
                    pend = pend[len(pend) - n:] if n < len(pend) else pend
            continue
        if op.startswith(_NOOPERAND):
            continue
        used = set()
        for t in p[1:]:
            used |= _regs(t)
        for r, l0 in pend:
            if r and used & r:
                out.append((fn, ln, s, l0))
                break
        if op.startswith(('ds_read', 'ds_load')):
            pend.append((_regs(p[1]), ln))
        elif op.startswith('ds_') or op.startswith(('s_load', 's_buffer_load', 's_scratch_load')):
            pend.append((None if op.startswith('s_') else set(), ln))
    return out


if __name__ == '__main__':
    f = findings(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else None)
    for fn, ln, s, l0 in f:
        print('%s:%d: %s   <- ds_read at line %d may be in flight' % (fn[:90], ln, s, l0))
    print('findings: %d' % len(f))
