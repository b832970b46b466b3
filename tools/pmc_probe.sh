#!/bin/bash
# ad-hoc PMC passes on the saturated solve launch (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_probe; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --drones 1048576 --steps 3 --warmup 1 --no-saturated --no-cpu-baseline --no-graph"
i=0
for SET in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ARGS > /dev/null 2> $OUT/p$i.err || { tail -3 $OUT/p$i.err; }
done
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob('gpurun_out/pmc_probe/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'solve_kernel' in r['Kernel_Name'] and int(r['Grid_Size'])>100000:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("mean kernel us under pmc: %.1f" % (sum(dur)/len(dur)/1e3))
for k,v in sorted(agg.items()): print("%-40s %14.6g" % (k, sum(v)/len(v)))
PY
