#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> [bench args...]
# GPU parity tests, then the bench, then a kernel trace of a short eager run.
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1 || { tail -30 gpurun_out/pytest_$tag.log; exit 1; }
tail -1 gpurun_out/pytest_$tag.log
timeout -k 10 200 python bench.py --steps 50 --no-cpu-baseline "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -20 gpurun_out/bench_$tag.err; exit 1; }
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o t -- python3 bench.py --steps 30 --no-cpu-baseline --no-graph --profile-steps 0 "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err || { tail -20 gpurun_out/prof_$tag.err; exit 1; }
echo done
