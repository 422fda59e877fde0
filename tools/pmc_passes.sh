#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_passes.sh <outdir> -- counter passes of a short bench run, one --pmc set per run
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 96 --warmup 96 --no-cpu-baseline --no-events"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/p$i --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/p$i.err || echo "pass $i failed: $set"
  echo "pass $i done: $set"
done <<'SETS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SETS
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
rm -rf $OUT/p[0-9]*/
cat $OUT/summary.txt
