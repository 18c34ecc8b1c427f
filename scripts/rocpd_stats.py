"""Per-kernel summary (calls, total/avg duration, share) from a rocprofv3 rocpd SQLite database -> text table.

usage: rocpd_stats.py DB [top_n] [--last-ms MS | --pipelined]
       --last-ms: only dispatches that start in the final MS of the trace (a sequential run: the steady-state steps without model
       packing / warm-up); --pipelined: only the steady state of bench.py's pipelined steps (scripts/_rocpd.py: the END of a bench
       trace is its sequential legs)
"""
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
last_ms = float(sys.argv[sys.argv.index('--last-ms') + 1]) if '--last-ms' in sys.argv else None
if last_ms is not None:
    args = [a for a in args if a != sys.argv[sys.argv.index('--last-ms') + 1]]
db = sqlite3.connect(args[0])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
where = ''
tag = f" (last {last_ms:g} ms of the trace)" if last_ms else ''
if '--pipelined' in sys.argv:
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _rocpd import pipelined_window
    w = pipelined_window(cur)
    if w is None:
        raise SystemExit('no pipelined region in this trace')
    where = f"where start >= {w[0]} and start <= {w[1]}"
    tag = f" (steady state of the pipelined steps: {(w[1] - w[0]) / 1e6:.1f} ms)"
elif last_ms is not None:
    t_end = cur.execute("select max(end) from kernels").fetchone()[0]
    where = f"where start >= {t_end - int(last_ms * 1e6)}"
rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels {where} group by {name_col} order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for n, c, s, a, mn, mx in rows[: int(args[1]) if len(args) > 1 else 40]:
    print(f"{n[:90]:90s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches" + tag)
