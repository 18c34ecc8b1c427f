"""Where the GPU sits idle: gaps between consecutive kernels of a rocprofv3 rocpd database (steady state = the last MS ms).
usage: rocpd_gaps.py DB [--last-ms MS] [--min-us US]   -> total idle, a histogram, and the kernels before / after the largest gaps"""
import collections, sqlite3, sys
a = sys.argv
last_ms = float(a[a.index('--last-ms') + 1]) if '--last-ms' in a else 400.0
min_us = float(a[a.index('--min-us') + 1]) if '--min-us' in a else 20.0
cur = sqlite3.connect(a[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
t_end = cur.execute("select max(end) from kernels").fetchone()[0]
rows = list(cur.execute(f"select start, end, {nm} from kernels where start >= {t_end - int(last_ms * 1e6)} order by start"))
busy_until, idle, gaps = rows[0][1], 0, []
for (s, e, n), prev in zip(rows[1:], rows[:-1]):
    if s > busy_until:
        g = s - busy_until
        idle += g
        if g >= min_us * 1e3:
            gaps.append((g, prev[2][:70], n[:70]))
    busy_until = max(busy_until, e)
span = rows[-1][1] - rows[0][0]
print(f'span {span/1e6:.1f} ms, idle {idle/1e6:.2f} ms ({100*idle/span:.1f} %), {len(rows)} dispatches; gaps >= {min_us:g} us: {len(gaps)} totalling {sum(g[0] for g in gaps)/1e6:.2f} ms')
hist = collections.Counter()
for (s, e, n), prev in zip(rows[1:], rows[:-1]):
    pass
by_pair = collections.defaultdict(lambda: [0, 0])
for g, p, n in gaps:
    by_pair[(p, n)][0] += 1; by_pair[(p, n)][1] += g
for (p, n), (c, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{t/1e6:8.2f} ms in {c:4d} gaps   after  {p}\n{"":27s}before {n}')
