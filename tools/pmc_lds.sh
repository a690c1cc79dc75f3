#!/bin/bash
# LDS counters (one PMC pass) over one tools/conv_bench.py case (diagnostic).
#   tools/pmc_lds.sh <case> <batch> <kernel name pattern>
case=${1:-fwd3x3}; batch=${2:-16}; pat=${3:-halo}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_lds_$case
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p1 -- python3 tools/conv_bench.py --only $case --batch $batch > $out/p1.log 2>&1 || exit 2
python3 tools/pmc_summary.py $out/p1/*/*counter_collection.csv $pat > $out/summary.txt
rm -rf $out/p1
cat $out/summary.txt
