#!/usr/bin/env python
"""Lint for the hand-counted waits of conv_k16.h (K16_ASYNC_A): walks the ISA of every kernel in a hipcc -S listing in program
order, keeps the queue of outstanding vector-memory instructions (loads with their destination registers, stores), retires the
oldest ones at every s_waitcnt vmcnt(n) (they complete in order), and reports any instruction that reads a register whose load
is still in the queue -- the compiler copying an operand before the data has arrived, or a wait count that is too large.
Straight-line approximation (skip-branches are walked through), which is exact for the steady-state row bodies.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k16.s cartpoleplusplus_amd/csrc/conv_fwd_k16.hip
  python profiles/tools/check_async_loads.py /tmp/k16.s
"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(name, lines):
    queue, bad, in_asm = [], [], False          # queue entries: (line, set of destination registers)
    for i, l in lines:
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
        elif t.startswith(';;#ASMEND'):
            in_asm = False
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith(('buffer_load', 'global_load')) and 'lds' not in t:
            # only loads issued from inline asm are hand-waited; the compiler waits for its own
            queue.append((i, regs_of(t.split()[1].rstrip(',')) if in_asm else set()))
            continue
        if op.startswith(('buffer_store', 'global_store')):
            queue.append((i, set()))
            continue
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
            continue
        if op in ('s_endpgm',):
            queue = []
            continue
        used = regs_of(t)
        for ln, dst in queue:
            if used & dst:
                bad.append((i, t, ln))
                break
    return bad


def main(path):
    text = open(path).read().split('\n')
    kernels, cur, name = [], None, None
    for i, l in enumerate(text, 1):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif cur is not None:
            cur.append((i, l))
            if '.amdhsa_kernel' in l:
                cur = None
    total = 0
    for name, lines in kernels:
        bad = check(name, lines)
        total += len(bad)
        print("%-70s %s" % (name[:70], "ok" if not bad else "%d suspicious reads" % len(bad)))
        for i, t, ln in bad[:6]:
            print("    line %d: %s   (load issued at line %d still outstanding)" % (i, t, ln))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
