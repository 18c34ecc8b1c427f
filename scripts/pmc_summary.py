"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads ->
hbm_read_bytes = 2 * FETCH_SIZE * 1024 for such kernels; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import json
import re
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            k = re.sub(r'\(.*$', '', k).replace('void ', '').strip()
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
print(f"{'kernel':48s} {'launches':>8s} {'FETCH KB/launch':>16s} {'WRITE KB/launch':>16s} {'HBM MB/launch (2*F+W)':>22s}")
for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0])[1]))[:60]:
    n, fv = fetch[k]
    nw, wv = write.get(k, [1, 0.0])
    hbm = (2 * fv / n + wv / max(nw, 1)) * 1024
    out[k] = dict(launches=n, fetch_kb_per_launch=fv / n, write_kb_per_launch=wv / max(nw, 1), hbm_bytes_per_launch=hbm)
    print(f'{k[:48]:48s} {n:8d} {fv / n:16.1f} {wv / max(nw, 1):16.1f} {hbm / 1e6:22.2f}')
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
