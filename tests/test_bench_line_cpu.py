"""The bench line the driver parses stays small: round 5's 22-KB line (13.6 KB of it a diagnostic parity block) was not read
(BENCH_r05.parsed = null).  bench.compact_line() is a pure function of the full record; the full record goes to a side file."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (no torch / GPU import at module level)

CANNED = [os.path.join(ROOT, 'profiles', n) for n in ('r05_bench_default.json', 'r05_bench_cfg4.json', 'r04_bench_default.json')]


@pytest.mark.parametrize('path', [p for p in CANNED if os.path.exists(p)])
def test_line_is_compact_and_strict_json(path):
    full = json.load(open(path))
    full['fabric'] = {'world_size_reported': 8, 'backend': 'nccl', 'rccl_version': '2.26.6', 'rank_ms_per_step_min_max': [29.1234567, 29.7654321]}
    full['value_full_maps'] = {'value': 2100.123456, 'unit': 'images/s', 'what': 'x' * 500}
    text = json.dumps(bench.compact_line(full), allow_nan=False, separators=(',', ':'))
    assert len(text) < bench.LINE_LIMIT == 6144
    assert '\n' not in text
    line = json.loads(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline') + (('cpu_baseline', 'parity') if 'cpu_baseline' in full else ()):     # cfg4 was run with --no_cpu_baseline
        assert k in line, k
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline'])
    if 'cpu_baseline' in full:
        assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(line['cpu_baseline'])
        assert all(not isinstance(v, (dict, str)) for v in line['parity'].values())           # the parity summary is numbers only
    assert 'model' not in line['config'] and line['config']['workload']
    assert line['roofline']['dominant_by_time']['kernel'].startswith('score_head_kernel')
    assert abs(line['value'] - full['value']) / full['value'] < 1e-4


def test_line_survives_bloat_and_non_finite_numbers():
    full = json.load(open(CANNED[0]))
    full['parity']['more'] = {'x' * 40: ['y' * 1000] * 100}
    full['roofline']['note'] = 'z' * 10000
    full['roofline']['kernel'] = 'k' * 5000
    full['cpu_baseline']['sample'] = 's' * 5000
    full['roofline']['traffic'] = float('nan')
    full['hbm']['kernels']['mano_fk']['GB/s'] = float('inf')
    text = json.dumps(bench.compact_line(full), allow_nan=False, separators=(',', ':'))
    assert len(text) < bench.LINE_LIMIT
    assert json.loads(text)['roofline']['traffic'] is None


def test_emit_prints_the_result_as_the_last_line_and_writes_the_side_file(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    full = json.load(open(CANNED[0]))
    out = io.StringIO()
    out.write('[some earlier output]\n')
    text = bench.emit(full, stream=out)
    lines = out.getvalue().rstrip('\n').split('\n')
    assert lines[-1] == text and json.loads(lines[-1])['detail'] == bench.DETAIL_PATH
    side = json.load(open(tmp_path / bench.DETAIL_PATH))
    assert side['parity'] == full['parity']               # nothing is lost: the whole record is in the side file


def test_traffic_is_per_workload_and_launch_weighted(tmp_path):
    prof = tmp_path / 'profiles'
    prof.mkdir()
    json.dump({'conv_igemm_glds_kernel<128, 128, 4, 2, false>': {'launches': 100, 'hbm_bytes_per_launch': 100.0},
               'conv_igemm_pers_kernel<128, 128, 4, 2>': {'launches': 300, 'hbm_bytes_per_launch': 200.0},
               'conv_igemm_glds_kernel<128, 64, 4, 2, false>': {'launches': 50, 'hbm_bytes_per_launch': 1e9}}, open(prof / 'r06_pmc_hbm_traffic.json', 'w'))
    assert bench.pmc_traffic(bench.CONV_CLASS_KERNELS, workload='cfg2', root=str(tmp_path)) == pytest.approx(175.0)
    assert bench.pmc_traffic(bench.CONV_CLASS_KERNELS, workload='cfg4', root=str(tmp_path)) is None      # no pass of that workload: null, never another one's
    assert bench.pmc_traffic(bench.CONV_CLASS_KERNELS, workload=None, root=str(tmp_path)) is None
    assert bench.workload_key(64, 100, 50) == 'cfg2' and bench.workload_key(128, 256, 100) == 'cfg4' and bench.workload_key(32, 100, 50) is None
