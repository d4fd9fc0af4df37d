# round 4: where the lean GuSTO kernel's wave cycles go -- parked (s_waitcnt / barrier), issue stalls, active -- and whether
# its 120 KB body misses the instruction cache.  Three separate --pmc passes over one bench step (run from the repo root).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
rocprofv3 -L > $R/gpurun_out/r04_counters_avail.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r04_ic_a -o a -- $B > $R/gpurun_out/r04_ic_a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $R/gpurun_out/r04_ic_b -o b -- $B > $R/gpurun_out/r04_ic_b.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $R/gpurun_out/r04_ic_c -o c -- $B > $R/gpurun_out/r04_ic_c.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for tag in "abc":
    for f in glob.glob(f"gpurun_out/r04_ic_{tag}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gusto_lean_kernel" not in k: continue
            agg[(k, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        # the largest dispatch = the timed 4096-rollout launch
        if agg:
            best = max(agg.items(), key=lambda kv: sum(kv[1].values()))
            out[tag] = {"kernel": best[0][0], "counters": dict(best[1]), "dispatches": len(agg)}
json.dump(out, open("gpurun_out/r04_icache.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
