#!/bin/bash
# Ad-hoc PMC passes on one kernel of one tool (runs on the GPU box):
#   bash tools/pmc_kernel.sh <kernel substring> "<python args>" "<counter set 1>" "<counter set 2>" ...
# One rocprofv3 --pmc pass per counter set (kernel trace only, as the pool requires); prints the mean
# per launch of every counter for kernels whose name contains the substring.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
KERN=$1; shift
ARGS=$1; shift
OUT=gpurun_out/pmc_kernel; rm -rf $OUT; mkdir -p $OUT
i=0
for SET in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.out 2> $OUT/p$i.err || { tail -3 $OUT/p$i.err; }
done
python3 - "$KERN" <<'PY'
import csv, glob, collections, sys
kern = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob('gpurun_out/pmc_kernel/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
if dur:
    print("kernel *%s*: %d launches, mean %.1f us under pmc" % (kern, len(dur), sum(dur) / len(dur) / 1e3))
for k, v in sorted(agg.items()):
    print("%-40s %14.6g" % (k, sum(v) / len(v)))
PY
rm -rf $OUT/p*/
