#!/bin/bash
# GPU box: kernel timeline (start / duration of every launch) of late passes of a formation pipeline.
#   bash tools/pipeline_timeline.sh <config 2|3> <tag>
set -o pipefail
CFG=${1:-3}; TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/timeline_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/trace -- python3 tools/formation_pipeline.py --config $CFG --reps 40 > /dev/null 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
python3 - $(dirname $(find $O/trace -name "*.db" | head -1)) > $O/timeline.txt <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1].rstrip("/") + "/*.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# the pipeline-as-run pass: the fused launch (or the sampler) starts a pipeline; take three late ones before the trailing loops
starts = [i for i, r in enumerate(rows) if "mesh_sweep" in r[0]] or [i for i, r in enumerate(rows) if "eval_groups" in r[0]]
i0 = starts[len(starts) * 3 // 4] - 8
t0 = rows[i0][1]
for n, a, b, q in rows[i0:i0 + 30]:
    print(f"q{q:<3d} {n.split('(')[0].replace('void ', '')[:58]:58s} start {(a - t0) / 1e3:8.1f}  end {(b - t0) / 1e3:8.1f}  dur {(b - a) / 1e3:6.1f}")
PY
cat $O/timeline.txt
rm -rf $O/trace
