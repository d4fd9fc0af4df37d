# rocprofv3 --pmc passes over tools/probes/pmc_short_horizon.py (GPU box, repository root): instruction mix and wave cycles of the one-wave
# short-horizon kernels per SCP iteration -> gpurun_out/<tag>_short_horizon_pmc.json
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
TAG=${1:-r05}; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sh_a -o a -- python3 $R/tools/probes/pmc_short_horizon.py > $R/gpurun_out/${TAG}_sh_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sh_b -o b -- python3 $R/tools/probes/pmc_short_horizon.py > $R/gpurun_out/${TAG}_sh_b.log 2>&1
cd $R; python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
work = None
for ln in open('gpurun_out/%s_sh_a.log' % tag):
    if ln.startswith('PMC_WORKLOAD '):
        work = json.loads(ln[len('PMC_WORKLOAD '):])
out = {'what': 'rocprofv3 --pmc over 33 consecutive one-rollout solves per horizon (tools/probes/pmc_short_horizon.py): counters summed over the dispatches of each '
               'short-horizon lean kernel, per SCP iteration; the interior point of these kernels runs on ONE wave, the other seven wait at a barrier', 'workload': work, 'kernels': {}}
for sub in ('a', 'b'):
    f = glob.glob('gpurun_out/%s_sh_%s/**/*counter_collection.csv' % (tag, sub), recursive=True)
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'gusto_lean_kernel' not in k:
            continue
        name = k.split('gusto_lean_kernel')[1].split('(')[0]
        e = out['kernels'].setdefault(name, {'dispatches': set()})
        e['dispatches'].add(r['Dispatch_Id'])
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for name, e in out['kernels'].items():
    nd = len(e.pop("dispatches"))
    e['dispatches_per_pass'] = nd
    N = 'N5' if ', 4, -1' in name else ('N3' if ', 1, -1' in name else None)
    its = work[N]['scp_iterations'] if (work and N) else None
    if its:
        e['per_scp_iteration'] = {k: v / its for k, v in e.items() if k.startswith('SQ_')}
        p = e['per_scp_iteration']
        if 'SQ_INSTS_VALU' in p:
            p['instructions (VALU + SALU + LDS)'] = p['SQ_INSTS_VALU'] + p['SQ_INSTS_SALU'] + p['SQ_INSTS_LDS']
json.dump(out, open('gpurun_out/%s_short_horizon_pmc.json' % tag, 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
