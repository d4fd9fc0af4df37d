cd /tmp; export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2a -o a -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 > $R/gpurun_out/pmc_l2a.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2b -o b -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 > $R/gpurun_out/pmc_l2b.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ('a','b'):
    f=glob.glob('gpurun_out/pmc_l2%s/**/*counter_collection.csv' % tag, recursive=True)
    if not f: print(tag,'no csv', glob.glob('gpurun_out/pmc_l2%s/**/*' % tag, recursive=True)[:5]); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'][:60]
        acc[k][row['Counter_Name']]+=float(row['Counter_Value']); 
    for k,v in acc.items():
        if 'lean' in k or 'proj' in k: print(tag, k, dict(v))
PY
