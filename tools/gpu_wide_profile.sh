#!/bin/bash
# usage (GPU box, repo root): tools/gpu_wide_profile.sh <tag>
# The wide workload (BASELINE configs[4] layer sizes, bf16 products): the bench line, a rocprofv3
# kernel trace of an eager run, and one counter pass for matrix-core busy cycles (counters only,
# no tracing domains in that run).
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/wide_$tag
mkdir -p $out
timeout -k 10 300 python3 bench.py --workload wide --steps 30 --warmup 5 --batches 16 --profile-steps 5 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --workload wide --steps 6 --warmup 2 --batches 8 --no-graph --profile-steps 0 > $out/trace.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o p -- python3 bench.py --workload wide --steps 4 --warmup 2 --batches 8 --no-graph --profile-steps 0 > $out/mfma.json 2> $out/mfma.err || { tail -5 $out/mfma.err; exit 1; }
python3 tools/pmc_summary.py $out/mfma "" > $out/mfma.txt && echo wide profile done
