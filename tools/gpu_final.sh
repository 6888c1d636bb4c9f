#!/bin/bash
# usage (GPU box, repo root): tools/gpu_final.sh <tag>
# The round's evidence for the bench workload: the default bench line (with cpu_baseline), rocprofv3
# kernel stats of an eager run and of the graph-replay run, HBM traffic per launch (separate --pmc
# passes), the row-sharded and wide-shape lines and the step timelines, in gpurun_out/final_<tag>/.
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/final_$tag
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/eager -o t -- python3 bench.py --steps 30 --no-cpu-baseline --no-graph --profile-steps 0 > $out/eager.json 2> $out/eager.err || { tail -5 $out/eager.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/graph -o t -- python3 bench.py --steps 400 --no-cpu-baseline --profile-steps 2 > $out/graph.json 2> $out/graph.err || { tail -5 $out/graph.err; exit 1; }
tools/gpu_pmc_traffic.sh final_$tag || exit 1
# the row-sharded engine on this one GPU (whole step as one graph per routing generation; exchanges as copies,
# then through RCCL in a group of one rank), the same with the exchanges launched eagerly, and the wide shapes
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sharded > $out/sharded.json 2> $out/sharded.err || { tail -5 $out/sharded.err; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sharded --force-collectives > $out/sharded_rccl.json 2> $out/sharded_rccl.err || { tail -5 $out/sharded_rccl.err; exit 1; }
XDFM_SHARD_STEP_GRAPH=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --sharded > $out/sharded_eager.json 2> $out/sharded_eager.err || { tail -5 $out/sharded_eager.err; exit 1; }
XDFM_SHARD_STEP_GRAPH=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --sharded --force-collectives > $out/sharded_rccl_eager.json 2> $out/sharded_rccl_eager.err || { tail -5 $out/sharded_rccl_eager.err; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload wide > $out/wide.json 2> $out/wide.err || { tail -5 $out/wide.err; exit 1; }
timeout -k 10 200 python3 tools/stamp_timeline.py --repeat 3 > $out/stamp_plain.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/stamp_timeline.py --sharded --repeat 2 > $out/stamp_sharded.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/stamp_timeline.py --sharded --force-collectives --repeat 2 > $out/stamp_sharded_rccl.txt 2>&1 || exit 1
echo final done
