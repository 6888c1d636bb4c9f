#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round3.sh <tag>
# The bench line, the one-GPU sharded lines (without and with every exchange through RCCL) and one run of the
# folded capture under GPU_MAX_HW_QUEUES=2; everything under gpurun_out/r3_<tag>/.
set -o pipefail
tag=$1; shift
out=gpurun_out/r3_$tag
mkdir -p $out
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sharded > $out/sharded.json 2> $out/sharded.err || { tail -5 $out/sharded.err; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sharded --force-collectives > $out/sharded_rccl.json 2> $out/sharded_rccl.err || { tail -5 $out/sharded_rccl.err; exit 1; }
GPU_MAX_HW_QUEUES=2 timeout -k 10 300 python3 bench.py --steps 50 --no-cpu-baseline > $out/q2.json 2> $out/q2.err; echo "exit status $?" >> $out/q2.err
for f in bench sharded sharded_rccl q2; do python3 tools/show_line.py $out/$f.json; done
tail -3 $out/q2.err
