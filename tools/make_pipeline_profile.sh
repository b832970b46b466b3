#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the config-3 / config-4 pipelines (solve -> sample ->
# all-gather -> pairwise pass -> mesh sweep) and of the streaming kernels -> gpurun_out/pipeline_<tag>/
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pipeline_${TAG}
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 tools/formation_pipeline.py --reps 30 > $OUT/config3.json 2> $OUT/c3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 tools/formation_pipeline.py --segments 20 --mesh --reps 30 > $OUT/config4.json 2> $OUT/c4.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/aux -- python3 tools/aux_bench.py > $OUT/aux_bench.txt 2> $OUT/aux.err || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
with open(out + "/pipeline_summary.md", "w") as f:
    for tag, title in (("c3", "configs[2]: 4096 drones x 10 segments, 91 samples, pairwise pass"),
                       ("c4", "configs[3]: 4096 drones x 20 segments, 96 samples, pairwise pass + 56-triangle mesh sweep"),
                       ("aux", "streaming kernels at 2^18 drones x 10 segments (tools/aux_bench.py)")):
        path = glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True)[0]
        f.write(f"### {title}\n\n| kernel | calls | mean us | min us | max us |\n|---|---|---|---|---|\n")
        for r in csv.DictReader(open(path)):
            if "msnap::" not in r["Name"]:
                continue
            name = r["Name"].split("(")[0].replace("void ", "")
            f.write("| `%s` | %s | %.1f | %.1f | %.1f |\n" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
        f.write("\n")
PY
cat $OUT/pipeline_summary.md
rm -rf $OUT/c3 $OUT/c4 $OUT/aux
