#!/bin/bash
# PMC passes (SQ counters) over one tools/conv_bench.py case (diagnostic; one counter set per run).
#   tools/pmc_case.sh <case> <batch> <kernel name pattern>
case=${1:-slab_multi4_16}; batch=${2:-16}; pat=${3:-dgrad_group}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_$case
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -- python3 tools/conv_bench.py --only $case --batch $batch > $out/p1.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/p2 -- python3 tools/conv_bench.py --only $case --batch $batch > $out/p2.log 2>&1 || exit 3
python3 tools/pmc_summary.py $out/p1/*/*counter_collection.csv $pat > $out/summary.txt
python3 tools/pmc_summary.py $out/p2/*/*counter_collection.csv $pat >> $out/summary.txt
rm -rf $out/p1 $out/p2
cat $out/summary.txt
