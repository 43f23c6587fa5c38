#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   -- separate rocprofv3 --pmc passes (never mixed with tracing)
export TMPDIR=/tmp
tag=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/pmc_${tag}_p$i.log 2>&1
done
