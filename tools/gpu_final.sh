#!/bin/bash
# usage (GPU box, repo root): tools/gpu_final.sh <tag>
# The round's evidence for the bench workload: the default bench line (with cpu_baseline), rocprofv3
# kernel stats of an eager run and of the graph-replay run, HBM traffic per launch (separate --pmc
# passes), in gpurun_out/final_<tag>/.
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/final_$tag
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/eager -o t -- python3 bench.py --steps 30 --no-cpu-baseline --no-graph --profile-steps 0 > $out/eager.json 2> $out/eager.err || { tail -5 $out/eager.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/graph -o t -- python3 bench.py --steps 400 --no-cpu-baseline --profile-steps 2 > $out/graph.json 2> $out/graph.err || { tail -5 $out/graph.err; exit 1; }
tools/gpu_pmc_traffic.sh final_$tag && echo final done
