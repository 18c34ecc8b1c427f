"""What the pipelined step pays beyond its chip-filling kernels: the time in which NO convolution / score-head kernel is executing,
split by the kernels that run there (or idle), from a rocprofv3 rocpd database (steady state = the last MS ms).
usage: rocpd_exposed.py DB [--last-ms MS | --pipelined] [--big 'conv_igemm,conv_winograd,score_head']
(--pipelined: the steady state of bench.py's pipelined steps, scripts/_rocpd.py; the END of a bench trace is its sequential legs)"""
import collections, sqlite3, sys
a = sys.argv
last_ms = float(a[a.index('--last-ms') + 1]) if '--last-ms' in a else 300.0
big = (a[a.index('--big') + 1] if '--big' in a else 'conv_igemm,conv_winograd,score_head').split(',')
cur = sqlite3.connect(a[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
t_end = cur.execute("select max(end) from kernels").fetchone()[0]
t0 = t_end - int(last_ms * 1e6)
if '--pipelined' in a:
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _rocpd import pipelined_window
    w = pipelined_window(cur)
    if w is None:
        raise SystemExit('no pipelined region in this trace')
    t0, t_end = w
rows = list(cur.execute(f"select start, end, {nm} from kernels where end >= {t0} and start <= {t_end} order by start"))
# a dispatch whose record carries a garbage (zero / far too early) start timestamp would count as live from the window's start on --
# r03's file charged __amd_rocclr_copyBuffer with more exposed time than that kernel's total duration that way.  No
# kernel of this workload runs longer than a few ms: records longer than --max-ms (default 20) are dropped and counted
max_ms = float(a[a.index('--max-ms') + 1]) if '--max-ms' in a else 20.0
n_all = len(rows)
rows = [r for r in rows if 0 < r[0] <= r[1] and (r[1] - r[0]) <= max_ms * 1e6]
dropped = n_all - len(rows)


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:48]


# sweep over the event points: at each elementary interval know which kernels are live
ev = []
for i, (s, e, n) in enumerate(rows):
    ev.append((max(s, t0), 0, i)); ev.append((max(min(e, t_end), max(s, t0)), 1, i))     # at equal times a start sorts before its own end
ev.sort()
live, live_big = set(), 0
prev = t0
tot = collections.Counter()
excl = collections.Counter()            # exposed time by the set of (small) kernels live in it
isbig = [any(b in n for b in big) for _, _, n in rows]
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        if live_big:
            tot['big kernel live'] += dt
        elif not live:
            tot['idle'] += dt
        else:
            tot['only small kernels'] += dt
            names = sorted({short(rows[j][2]) for j in live})
            for k in names:
                excl[k] += dt / len(names)
        prev = t
    if kind == 0:
        live.add(i); live_big += isbig[i]
    else:
        live.discard(i); live_big -= isbig[i]
span = t_end - t0
print(f'window {span/1e6:.1f} ms, {len(rows)} dispatches ({dropped} records with an implausible duration > {max_ms:g} ms or a bad start dropped); big = {big}')
by_kernel = collections.Counter()
for s_, e_, n_ in rows:
    by_kernel[short(n_)] += min(e_, t_end) - max(s_, t0)
for k, v in tot.most_common():
    print(f'  {k:22s} {v/1e6:8.2f} ms  {100*v/span:5.1f} %')
print('exposed time (no big kernel live), shared equally among the small kernels live at that moment:')
for k, v in excl.most_common(25):
    assert v <= by_kernel[k] + 1, (k, v, by_kernel[k])           # exposed time of a kernel cannot exceed its own time in the window
    print(f'  {v/1e6:7.3f} ms  {100*v/span:5.2f} %  {k}   (its total time in the window: {by_kernel[k]/1e6:.3f} ms)')
