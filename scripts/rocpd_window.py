"""Timeline around the moments the pipelined step runs no big kernel: for the N longest such intervals of the last MS ms of a rocprofv3 rocpd
database, every kernel live within +- PAD us, one line each (start / end relative to the interval start, queue, workgroups, name).
usage: rocpd_window.py DB [--last-ms MS] [--n N] [--pad US]"""
import sqlite3, sys
a = sys.argv
last_ms = float(a[a.index('--last-ms') + 1]) if '--last-ms' in a else 300.0
N = int(a[a.index('--n') + 1]) if '--n' in a else 3
pad = float(a[a.index('--pad') + 1]) * 1e3 if '--pad' in a else 150e3
big = ('conv_igemm', 'conv_winograd', 'score_head')
cur = sqlite3.connect(a[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
qcol = [c for c in cols if 'queue' in c or 'stream' in c]
gcol = [c for c in cols if c in ('grid_size', 'grid_x', 'grid_size_x', 'workgroup_size', 'workgroup_x', 'workgroup_size_x')]
print('columns:', cols)
t_end = cur.execute("select max(end) from kernels").fetchone()[0]
t0 = t_end - int(last_ms * 1e6)
sel = ', '.join(['start', 'end', nm] + qcol[:1] + gcol)
rows = list(cur.execute(f"select {sel} from kernels where end >= {t0} order by start"))
ev = []
for i, r in enumerate(rows):
    if any(b in r[2] for b in big):
        ev.append((max(r[0], t0), 1)); ev.append((r[1], -1))
ev.sort()
live, prev, gaps = 0, t0, []
for t, d in ev:
    if live == 0 and t > prev:
        gaps.append((t - prev, prev, t))
    live += d
    prev = t if live == 0 else prev
    if live == 0:
        prev = t
gaps.sort(reverse=True)
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
for g, s, e in gaps[:N]:
    print(f'\n==== no big kernel for {g / 1e3:.1f} us')
    for r in rows:
        if r[1] >= s - pad and r[0] <= e + pad:
            extra = ' '.join(str(v) for v in r[3:])
            print(f'  {(r[0] - s) / 1e3:9.1f} .. {(r[1] - s) / 1e3:9.1f} us  [{extra}]  {short(r[2])}')
