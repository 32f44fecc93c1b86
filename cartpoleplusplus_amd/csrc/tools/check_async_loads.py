#!/usr/bin/env python
"""Build-time check of the hand-counted waits of conv_k16.h (K16_ASYNC_A) -- and of every other vector-memory wait in a listing.

conv_fwd_k16_kernel issues its A-operand loads from inline asm and waits for them with hand-counted `s_waitcnt vmcnt(N)`: N must be the
number of vector-memory instructions issued AFTER the load that is needed (they retire in order).  The counts are derived from the
source order of the loads and stores; a compiler that orders them differently, merges two stores, or copies a register whose load is
still in flight (an asm load's destination counts as written at the asm statement) breaks the kernel without a diagnostic -- and
possibly without a failing test: the failure is a race.

This tool reads `hipcc -S --cuda-device-only` output, builds each kernel's control-flow graph (labels, s_branch / s_cbranch_*), and runs
a forward data-flow analysis whose state is the queue of outstanding vector-memory instructions (loads with their destination
registers, stores), youngest last:
  * a load / store appends an entry;  `s_waitcnt vmcnt(n)` keeps the youngest n entries;
  * where paths meet, the states are merged position by position from the YOUNGEST end (a vmcnt(n) is about the youngest n), taking
    the union of the destination registers: a register is "in flight" if it is on ANY path -- no false negatives;
  * loops are iterated to a fixed point (the queue is capped at 64 entries: vmcnt is a 6-bit counter).
An instruction that reads or writes a register that is in flight in its entry state is a violation: a wait that is too small, a wait
that was moved, an extra or a missing vector-memory instruction between a load and its wait, or a compiler copy of an in-flight register
all end here.  Compiler-counted loads are checked the same way (they pass by construction).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k16.s cartpoleplusplus_amd/csrc/conv_fwd_k16.hip
  python cartpoleplusplus_amd/csrc/tools/check_async_loads.py /tmp/k16.s            # exit status 1 on a violation

csrc/Makefile runs it on conv_fwd_k16.hip's listing as part of `all` (target check-waits); tests/test_wait_checker.py feeds it doctored
listings.
"""
import re
import sys

CAP = 64
_REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
_VM_LOAD = ('buffer_load', 'global_load', 'flat_load', 'scratch_load')
_VM_STORE = ('buffer_store', 'global_store', 'flat_store', 'scratch_store', 'buffer_atomic', 'global_atomic', 'flat_atomic')


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_kernels(text):
    """[(name, [(line number, stripped instruction or 'label:')])] for every function symbol of the listing"""
    kernels, cur = [], None
    for i, l in enumerate(text.split('\n'), 1):
        m = re.match(r'^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$', l)
        if m and not l.startswith('.L') and not l.startswith('\t'):
            cur = []
            kernels.append((m.group(1), cur))
            continue
        if cur is None:
            continue
        t = l.split(';')[0].strip() if not l.strip().startswith(';') else ''
        if '.end_amdhsa_kernel' in l or re.match(r'^\s*\.section', l) or re.match(r'^\.Lfunc_end', l):
            cur = None
            continue
        if not t or t.startswith('.') and not re.match(r'^\.L[\w$]+:', t):
            continue
        cur.append((i, t))
    return [(n, ins) for n, ins in kernels if any(t.split()[0].startswith(('s_', 'v_', 'buffer_', 'ds_', 'global_')) for _, t in ins)]


def blocks_of(ins):
    """basic blocks: list of (label or None, [(line, text)], successors as block indexes)"""
    starts = {0}
    label_at = {}
    for k, (_, t) in enumerate(ins):
        m = re.match(r'^(\.L[\w$]+):$', t)
        if m:
            label_at[m.group(1)] = k
            starts.add(k)
        op = t.split()[0]
        if op.startswith(('s_branch', 's_cbranch', 's_endpgm', 's_setpc')) and k + 1 < len(ins):
            starts.add(k + 1)
    order = sorted(starts)
    index_of = {s: b for b, s in enumerate(order)}
    blocks = []
    for b, s in enumerate(order):
        e = order[b + 1] if b + 1 < len(order) else len(ins)
        body = ins[s:e]
        succ = []
        last = body[-1][1] if body else ''
        op = last.split()[0] if last else ''
        tgt = re.search(r'(\.L[\w$]+)', last) if op.startswith(('s_branch', 's_cbranch')) else None
        if tgt and tgt.group(1) in label_at:
            succ.append(index_of[label_at[tgt.group(1)]])
        if not op.startswith(('s_branch', 's_endpgm', 's_setpc')) and e < len(ins):
            succ.append(index_of[e])
        blocks.append((body, succ))
    return blocks


def merge(a, b):
    """position-wise from the youngest end; entries are frozensets of destination registers (empty: a store)"""
    if a is None:
        return b
    if b is None:
        return a
    n = max(len(a), len(b))
    out = []
    for k in range(1, n + 1):
        x = a[-k] if k <= len(a) else frozenset()
        y = b[-k] if k <= len(b) else frozenset()
        out.append(x | y)
    return tuple(reversed(out))


def transfer(body, state, report=None):
    q = list(state)
    for line, t in body:
        if t.endswith(':'):
            continue
        op = t.split()[0]
        operands = t[len(op):]
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                if len(q) > n:
                    q = q[len(q) - n:] if n else []
            continue
        used = regs_of(operands)
        if report is not None and used:
            for pos, dst in enumerate(q):
                if dst and (used & dst):
                    report.append((line, t, len(q) - pos))
                    break
        if op.startswith(_VM_LOAD) and ' lds' not in t:
            first = operands.split(',')[0]
            q.append(frozenset(regs_of(first)))
        elif op.startswith(_VM_STORE) or (op.startswith(_VM_LOAD) and ' lds' in t):
            q.append(frozenset())
        if len(q) > CAP:
            q = q[len(q) - CAP:]
    return tuple(q)


def check_kernel(ins):
    blocks = blocks_of(ins)
    if not blocks:
        return []
    state_in = [None] * len(blocks)
    state_in[0] = ()
    work = [0]
    rounds = 0
    while work and rounds < 200000:
        rounds += 1
        b = work.pop()
        out = transfer(blocks[b][0], state_in[b])
        for s in blocks[b][1]:
            m = merge(state_in[s], out)
            if m != state_in[s]:
                state_in[s] = m
                work.append(s)
    bad = []
    for b, (body, _succ) in enumerate(blocks):
        if state_in[b] is not None:
            transfer(body, state_in[b], bad)
    return bad


def check_listing(text, only=None):
    """{kernel name: [(line, instruction, age of the load: 1 = youngest)]} for every kernel with a violation"""
    res = {}
    for name, ins in parse_kernels(text):
        if only and not any(o in name for o in only):
            continue
        bad = check_kernel(ins)
        res[name] = bad
    return res


def main(argv):
    paths = [a for a in argv if not a.startswith('--')]
    only = [a[len('--only='):] for a in argv if a.startswith('--only=')]
    total = 0
    for path in paths:
        res = check_listing(open(path).read(), only or None)
        for name, bad in res.items():
            total += len(bad)
            print("%-78s %s" % (name[:78], "ok" if not bad else "%d reads / writes of registers whose load is in flight" % len(bad)))
            for line, t, age in bad[:8]:
                print("    %s:%d: %s   (the load is %d vector-memory instruction(s) from the youngest)" % (path, line, t, age))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
