#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_traffic.sh <tag>
# HBM traffic per kernel launch: one rocprofv3 counter pass each for FETCH_SIZE and WRITE_SIZE
# (counters only, no tracing domains) over a short eager run of the bench workload.
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 bench.py --steps 6 --warmup 2 --batches 16 --no-cpu-baseline --no-graph --profile-steps 0 > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE && python3 tools/pmc_summary.py $out "" --json $out/traffic.json > $out/traffic.txt && echo pmc done
