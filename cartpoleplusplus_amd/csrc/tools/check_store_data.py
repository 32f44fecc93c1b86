#!/usr/bin/env python3
"""Build-time check of a gfx950 hazard LLVM's recogniser does not cover (found in round 6, profiles/NOTEBOOK_r06.md 10).

A `buffer_store_dwordx2/x3/x4` reads its DATA registers from the VGPR file some cycles after it issues.  LLVM inserts wait states between
such a store and a VALU write of its data registers only when the store's soffset is NOT a register (GCNHazardRecognizer::createsVALUHazard:
"this hazard only exists if the instruction is not using a register in the soffset field") -- but on gfx950 a `v_max_f32 v202, |v202|, |v202|`
ONE instruction behind `buffer_store_dwordx4 v[200:203], v174, s[0:3], s4 offen` changed what was stored, differently from run to run
(conv_dx_rs.h with its row maximum live at 64-wide rows: conv1's gradient off by 1e-3 in all odd channels; with the data registers
kept untouched for four cycles the same build is bit-reproducible and parity-green).

Measured in isolation (profiles/diag/store_hazard_probe.hip, 4e8 stored words per case): dwordx3 / dwordx4 with an SGPR soffset are poisoned
by a write 0 wait states behind them and clean from 1; with an immediate soffset at 0 and 1, clean from 2 (LLVM's rule); dwordx2 never.

This script walks the listing of a translation unit (`hipcc -S --cuda-device-only`) and reports every multi-dword buffer / global store whose
data registers are overwritten by a VALU instruction fewer than MIN_WAIT wait states later (MIN_WAIT_KNOWN where LLVM applies its own
rule: immediate soffset, global stores) in straight-line code (an `s_nop n` counts
n + 1, every other instruction 1; a v_mfma / load destination lands later than that by itself and is not counted as an overwrite; the
walk stops at a branch or label).  Exit status 1 if there is such a site.
usage: check_store_data.py listing.s [...]"""
import re
import sys

MIN_WAIT = 4          # behind a buffer store whose soffset is an SGPR (LLVM adds nothing there: buffer_store_b128_held() in conv_kyo.h does)
MIN_WAIT_KNOWN = 2    # behind the stores LLVM's recogniser covers itself (immediate soffset, global / flat): its own rule for gfx940+
STORE = re.compile(r'^(buffer_store_dwordx[234]|global_store_dwordx[234]|buffer_store_dword|global_store_dword|buffer_store_short|buffer_store_byte)\b')


def regs(tok):
    tok = tok.strip().strip('|').lstrip('-')
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def check(path, wide_only=True):
    lines = [l.strip() for l in open(path)]
    ins = [(i + 1, l) for i, l in enumerate(lines) if l and not l.startswith(';') and not l.startswith('.') and not l.startswith('//')]
    hits, kernel = [], '?'
    for k, (ln, l) in enumerate(ins):
        if l.endswith(':'):
            if not l.startswith('.L'):
                kernel = l[:-1]
            continue
        m = STORE.match(l)
        if not m:
            continue
        if wide_only and not re.search(r'x[34]$', m.group(1)):      # (64-bit stores: the data leaves with the address -- conv_rs16.h's plane stores are overwritten at +0 by the hundred and are bit-reproducible)
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        data = regs(ops[1] if m.group(1).startswith('global') else ops[0])
        sgpr_soffset = m.group(1).startswith('buffer') and len(ops) > 3 and re.match(r's\d+|s\[', ops[3].split()[0]) is not None
        need = MIN_WAIT if sgpr_soffset else MIN_WAIT_KNOWN
        wait = 0
        for kk in range(k + 1, min(len(ins), k + 1 + need + 2)):
            pl = ins[kk][1]
            if pl.endswith(':') or pl.startswith('s_cbranch') or pl.startswith('s_branch') or pl.startswith('s_endpgm') or pl.startswith('s_setpc'):
                break
            if pl.startswith('s_nop'):
                wait += int(pl.split()[1]) + 1
                continue
            if wait >= need:
                break
            if pl.startswith('v_') and not pl.startswith('v_mfma') and not pl.startswith('v_cmp') and not pl.startswith('v_accvgpr_read') :
                dst = regs(pl.split(None, 1)[1].split(',')[0]) if ' ' in pl else set()
                if dst & data:
                    hits.append((kernel, ln, wait, l, pl))
                    break
            wait += 1
    return hits


def main(argv):
    bad = 0
    for path in argv:
        hits = check(path)
        print("%s: %d wide store(s) whose data registers a VALU instruction overwrites too early (< %d wait states behind an SGPR-soffset buffer store, < %d elsewhere)" % (path, len(hits), MIN_WAIT, MIN_WAIT_KNOWN))
        for kernel, ln, wait, st, ov in hits:
            print("  %s line %d (+%d wait states): %s   <-   %s" % (kernel[:70], ln, wait, st[:70], ov[:70]))
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
