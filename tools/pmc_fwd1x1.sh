#!/bin/bash
# PMC passes over the dense-layer 1x1 forward micro-benchmark (diagnostic; one counter set per run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_fwd1x1
mkdir -p $out
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -- python3 tools/conv_bench.py --only fwd1x1 > $out/p1.log 2>&1 || exit 2
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/p2 -- python3 tools/conv_bench.py --only fwd1x1 > $out/p2.log 2>&1 || exit 3
ls $out/p1/*/ $out/p2/*/
