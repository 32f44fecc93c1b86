"""reads the K16SPAN lines of a -DK16_SPAN_PROBE build (start / loop / end of every fourth workgroup of conv_fwd_k16_kernel, 100 MHz
ticks) and prints, for the LAST launch of each instance, how the workgroups' lifetimes lie inside the launch."""
import sys, collections
rows = collections.defaultdict(list)
for l in sys.stdin:
    if "K16SPAN" not in l:
        continue
    f = l.split()
    cin, y, x, st, lo, en = int(f[2]), int(f[4]), int(f[6]), int(f[8]), int(f[10]), int(f[12])
    rows[cin].append((st, lo, en, y, x))
for cin, r in rows.items():
    r.sort()
    # split into launches: a gap of > 20 us between consecutive starts
    launches, cur = [], [r[0]]
    for a in r[1:]:
        if a[0] - cur[-1][0] > 2000:
            launches.append(cur); cur = []
        cur.append(a)
    launches.append(cur)
    L = launches[-2] if len(launches) > 1 else launches[-1]
    t0 = min(a[0] for a in L)
    print("cin %d: %d launches seen; one of the last: %d workgroups sampled, span %.1f us" % (cin, len(launches), len(L), (max(a[2] for a in L) - t0) / 100.0))
    st = sorted((a[0] - t0) / 100.0 for a in L)
    du = sorted((a[2] - a[0]) / 100.0 for a in L)
    se = sorted((a[1] - a[0]) / 100.0 for a in L)
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
    print("  start   min %.1f  p25 %.1f  p50 %.1f  p75 %.1f  max %.1f us" % (st[0], q(st, .25), q(st, .5), q(st, .75), st[-1]))
    print("  setup   min %.1f  p50 %.1f  max %.1f us" % (se[0], q(se, .5), se[-1]))
    print("  life    min %.1f  p25 %.1f  p50 %.1f  p75 %.1f  max %.1f us" % (du[0], q(du, .25), q(du, .5), q(du, .75), du[-1]))
    for y in sorted(set(a[3] for a in L)):
        d = sorted((a[2] - a[0]) / 100.0 for a in L if a[3] == y)
        print("  network %d: life min %.1f  p50 %.1f  max %.1f us (%d sampled)" % (y, d[0], q(d, .5), d[-1], len(d)))
