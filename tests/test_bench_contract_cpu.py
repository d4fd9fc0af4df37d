"""The JSON line `bench.py` prints (the driver's contract): checked on the committed log of the last full run
(profiles/r02h_bench.log) -- no GPU needed -- and on the argument parser's defaults."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_log():
    logs = sorted((f for f in os.listdir(os.path.join(ROOT, 'profiles')) if re.fullmatch(r'r\d+[a-z]?_bench\.log', f)),
                  key=lambda f: (int(re.match(r'r(\d+)', f).group(1)), f))
    return os.path.join(ROOT, 'profiles', logs[-1])


def _last_line():
    line = [l for l in open(_newest_log()) if l.startswith('{"metric"')][-1]
    return json.loads(line)


def test_committed_bench_line_is_small_enough_for_the_driver():
    """Round 5 lost its record to a 20 KB line (the driver keeps a tail of stdout).  From round 6 on the LAST stdout line of
    the committed run must parse alone and stay under bench.MAX_LINE; the full record sits on an earlier line."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    name = _newest_log()
    if int(re.match(r'r(\d+)', os.path.basename(name)).group(1)) < 6:
        import pytest
        pytest.skip('no round-6 bench log committed yet')
    lines = [l.rstrip('\n') for l in open(name) if l.strip()]
    last = [l for l in lines if l.startswith('{')][-1]
    assert last == lines[-1] or not lines[-1].startswith(('{', 'BENCH_DETAIL'))
    assert len(last) <= bench.MAX_LINE == 4096
    d = json.loads(last)
    assert 'secondary' not in d and d['roofline']['frac'] > 0 and d['cpu_baseline']['value'] > 0
    full = [l for l in lines if l.startswith('BENCH_DETAIL ')]
    assert full and 'secondary' in json.loads(full[-1].split(' ', 1)[1])


def test_compact_line_of_a_full_record_survives_a_4k_tail():
    """The printing path of a real run (bench.emit) fed with round 5's 20 KB record through the stub device: the last stdout
    line alone is the driver's line -- every contract key, roofline, cpu_baseline, parity_sample -- in < 4096 characters."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update({'SRH_BENCH_STUB_DEVICE': '1', 'SRH_BENCH_STUB_RECORD': os.path.join(ROOT, 'profiles', 'r05_bench.log')})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    tail = out.stdout[-4096:]                      # what a driver that keeps 4 KB of stdout would hold
    last = tail.strip().splitlines()[-1]
    assert len(last) < 4096 and last.startswith('{"metric"')
    d = json.loads(last)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity_sample'):
        assert k in d, k
    assert 'secondary' not in d
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(d['roofline'])
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(d['cpu_baseline'])
    assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) < 1e-4
    full = [l for l in out.stdout.splitlines() if l.startswith('BENCH_DETAIL ')]
    assert len(full) == 1 and 'secondary' in json.loads(full[0].split(' ', 1)[1])


def test_bench_line_schema():
    d = _last_line()
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    # BASELINE.md holds no published number: null, or (round 3, as the round-2 review asked) the CPU / GPU single-solve ratio
    # with its definition spelled out in the line
    assert d['vs_baseline'] is None or (d['vs_baseline'] > 0 and isinstance(d.get('vs_baseline_definition'), str))
    assert d['dtype'] == 'f64' and 'workload' in d['config'] and 'model' not in d['config']
    if isinstance(base.get('metric'), str):
        assert d['metric'] == base['metric'] or base['metric'] in d['metric'] or d['metric'] in base['metric']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-4 and (r['traffic'] is None or r['traffic'] > 0)
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


def test_gpus_flag_starts_one_rank_per_gpu_dry_launch():
    """`python bench.py --gpus 2` without torch.distributed.run around it must start 2 ranks:
    --dry-launch prints what would be started, without touching a GPU."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--dry-launch'],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d['n_ranks'] == 2 and len(d['env']) == 2
    assert [e['RANK'] for e in d['env']] == ['0', '1'] and [e['LOCAL_RANK'] for e in d['env']] == ['0', '1']
    assert all(e['WORLD_SIZE'] == '2' and e['MASTER_ADDR'] == '127.0.0.1' for e in d['env'])
    assert len({e['MASTER_PORT'] for e in d['env']}) == 1
    assert d['cmd'][1].endswith('bench.py') and '--dry-launch' not in d['cmd'] and d['cmd'][-4:] == ['--gpus', '2', '--steps', '3']


def test_launcher_runs_children_and_propagates_failure():
    """launch_ranks with a stand-in child command: every rank gets its own environment, rank 0's stdout passes through,
    a failing rank makes the launcher's exit code non-zero."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "rc = bench.launch_ranks(3, [], False, cmd=[sys.executable, '-c', "
            "'import os, sys; print(\"rank\", os.environ[\"RANK\"], os.environ[\"WORLD_SIZE\"]); "
            "sys.exit(FAIL if os.environ[\"RANK\"] == \"2\" else 0)']); print('rc', rc)" % ROOT)
    for fail, want in ((0, 0), (7, 7)):
        out = subprocess.run([sys.executable, '-c', code.replace('FAIL', str(fail))], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.strip().splitlines()
        assert 'rank 0 3' in lines and not any(l.startswith('rank 1') or l.startswith('rank 2') for l in lines)
        assert lines[-1] == 'rc %d' % want


def _run_stub(extra_env, n=2, timeout=300):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update({'SRH_BENCH_STUB_DEVICE': '1'}, **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '0'],
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_two_ranks_for_real_up_to_the_first_gpu_call():
    """`python bench.py --gpus 2` run FOR REAL with SRH_BENCH_STUB_DEVICE=1 (gloo instead of RCCL, a stand-in for the solve):
    the launcher starts two rank processes, they rendezvous on 127.0.0.1, time between barriers, take the max over ranks,
    run the reduction step of the sharded rollout batch, and ONLY rank 0 prints the one JSON line."""
    out = _run_stub({})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['stub'] is True and d['value'] is None and d['n_gpus'] == 2 and d['steps'] == 2
    assert len(d['ms_per_step_per_rank']) == 2
    # rank 1 sleeps twice as long per step: the reported time is the slowest rank's
    assert d['ms_per_step'] >= max(d['ms_per_step_per_rank']) - 1e-6 and d['ms_per_step_per_rank'][1] >= 15.0
    assert d['best_is_global_argmin'] and d['costs_gathered'] == 256


def test_bench_two_ranks_exit_code_when_a_rank_dies():
    """A rank that dies before the first collective must not leave the launcher waiting for ever: non-zero exit, no JSON line."""
    out = _run_stub({'SRH_BENCH_STUB_FAIL_RANK': '1'}, timeout=200)
    assert out.returncode != 0
    assert not [l for l in out.stdout.strip().splitlines() if l.startswith('{"stub"')]


def test_compact_line_without_a_cpu_leg_and_with_failed_legs():
    """The driver's line of a multi-GPU run has no cpu_baseline / parity_sample (rank 0 at N = 1 only), a failed CPU leg is reported as a
    short error -- and whatever a secondary grew into never reaches the line."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = json.loads([l for l in open(os.path.join(ROOT, 'profiles', 'r05_bench.log')) if l.startswith('{"metric"')][-1])
    multi = {k: v for k, v in rec.items() if k not in ('cpu_baseline', 'parity_sample')}
    multi.update(n_gpus=8, vs_baseline=None, secondary={'blob': 'x' * 100000})
    d = json.loads(bench.compact_line(multi))
    assert 'cpu_baseline' not in d and 'parity_sample' not in d and 'secondary' not in d and 'vs_baseline_definition' not in d
    assert d['n_gpus'] == 8 and d['roofline']['frac'] > 0 and d['config']['workload']
    failed = dict(rec, cpu_baseline={'error': 'RuntimeError(' + 'y' * 5000 + ')'})
    text = bench.compact_line(failed)
    assert len(text) < bench.MAX_LINE and len(json.loads(text)['cpu_baseline']['error']) <= 300
    # a workload description that outgrows the budget sheds the optional parts, never a contract key
    fat = dict(rec)
    fat['config'] = dict(rec['config'], workload='w' * 2500)
    d = json.loads(bench.compact_line(fat))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
