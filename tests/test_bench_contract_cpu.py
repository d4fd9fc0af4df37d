"""The JSON line `bench.py` prints (the driver's contract): checked on the committed log of the last full run
(profiles/r02h_bench.log) -- no GPU needed -- and on the argument parser's defaults."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line():
    logs = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if re.fullmatch(r'r\d+[a-z]_bench\.log', f))
    line = [l for l in open(os.path.join(ROOT, 'profiles', logs[-1])) if l.startswith('{"metric"')][-1]
    return json.loads(line)


def test_bench_line_schema():
    d = _last_line()
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert d['dtype'] == 'f64' and 'workload' in d['config'] and 'model' not in d['config']
    if isinstance(base.get('metric'), str):
        assert d['metric'] == base['metric'] or base['metric'] in d['metric'] or d['metric'] in base['metric']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and (r['traffic'] is None or r['traffic'] > 0)
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str)
    assert d['value'] > 0 and d['ms_per_step'] > 0
    p = d['parity_sample']
    assert p['iters_equal'] and p['max_rel_traj'] <= p['tolerance']


def test_bench_defaults_are_single_gpu_and_short():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "'--gpus', type=int, default=1" in src
    m = re.search(r"'--steps', type=int, default=(\d+)", src)
    assert m and int(m.group(1)) <= 10
