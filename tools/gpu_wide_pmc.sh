#!/bin/bash
# usage (GPU box, repo root): tools/gpu_wide_pmc.sh <tag>
# Hardware-counter passes over a short eager run of the wide workload, one rocprofv3 run per pass
# (counters only, no tracing domains): where the waves of the bf16 CIN kernels wait, and how much of
# what they stream comes from HBM.
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/widepmc_$tag
mkdir -p $out
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 bench.py --workload wide --steps 3 --warmup 1 --batches 4 --no-cpu-baseline --no-graph --profile-steps 0 > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
}
run sqa SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run sqb SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT &&
run tc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum &&
run fetch FETCH_SIZE && run write WRITE_SIZE &&
python3 tools/pmc_summary.py $out "cin_bf16" > $out/summary.txt && echo wide pmc done
