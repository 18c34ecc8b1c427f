"""Text timeline of a pipelined run from a rocprofv3 rocpd database: one column per launching host thread, one row per BIN us of the steady
state (the last SKIP ms of the trace -- the drain of the pipeline -- left out): C = a convolution of that thread is executing in the bin,
H = a score head, p = the pose encoder, s = another sampler kernel, a = anything else, . = nothing.
usage: rocpd_lanes.py DB [--ms 60] [--skip-ms 120 | --from-ms T] [--bin 250]"""
import sqlite3, sys, collections
a = sys.argv
span = float(a[a.index('--ms') + 1]) if '--ms' in a else 60.0
skip = float(a[a.index('--skip-ms') + 1]) if '--skip-ms' in a else 120.0
binus = float(a[a.index('--bin') + 1]) if '--bin' in a else 250.0
cur = sqlite3.connect(a[1]).cursor()
t_end = cur.execute("select max(end) from kernels").fetchone()[0]
t1 = t_end - int(skip * 1e6); t0 = t1 - int(span * 1e6)
if '--from-ms' in a:                                      # window given from the START of the trace instead
    t_begin = cur.execute("select min(start) from kernels").fetchone()[0]
    t0 = t_begin + int(float(a[a.index('--from-ms') + 1]) * 1e6); t1 = t0 + int(span * 1e6)
rows = list(cur.execute(f"select start, end, name, tid, stream_id from kernels where end >= {t0} and start <= {t1} order by start"))
def cls(n):
    if 'conv_igemm' in n or 'conv_winograd' in n: return 'C'
    if 'score_head' in n: return 'H'
    if 'pose_encoder' in n: return 'p'
    if any(k in n for k in ('time_embed', 'norm_', 'dense_', 'rk_', 'stage_input', 'denoise')): return 's'
    return 'a'
rank = {'C': 5, 'H': 4, 'p': 3, 's': 2, 'a': 1, '.': 0}
tids = sorted({r[3] for r in rows})
nb = int(span * 1e3 / binus)
lanes = {t: ['.'] * nb for t in tids}
for s, e, n, tid, st in rows:
    c = cls(n)
    b0, b1 = max(0, int((s - t0) / 1e3 / binus)), min(nb - 1, int((e - t0) / 1e3 / binus))
    for b in range(b0, b1 + 1):
        if rank[c] > rank[lanes[tid][b]]: lanes[tid][b] = c
print('threads:', tids, f'  bin {binus:g} us;  C conv  H head  p pose-enc  s sampler algebra  a other')
big = 0
for b in range(nb):
    line = '  '.join(lanes[t][b] for t in tids)
    anybig = any(lanes[t][b] in 'CH' for t in tids)
    big += anybig
    print(f'{b * binus / 1e3:7.2f} ms  {line}  {"" if anybig else "<-- no big kernel"}')
print(f'bins with a big kernel: {big}/{nb}')
