#!/bin/bash
# usage (GPU box, repo root): tools/gpu_graph_trace.sh <tag>
# The default bench line, then a rocprofv3 kernel trace of the graph-replay run (per-kernel durations inside the replay;
# the tracer serialises the graph's queues, so the step reads longer there than untraced).
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/gt_$tag
mkdir -p $out
timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/graph -o t -- python3 bench.py --steps 100 --no-cpu-baseline --profile-steps 2 "$@" > $out/graph.json 2> $out/graph.err || { tail -5 $out/graph.err; exit 1; }
python3 tools/show_bench.py $out/bench.json
