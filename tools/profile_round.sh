#!/bin/bash
# Evidence run for profiles/: bench line, rocprofv3 kernel stats of the same command, two PMC passes (HBM bytes),
# per-role breakdown, traffic summary.  Usage (on the GPU box): tools/profile_round.sh <tag> [batch]
tag=${1:-x}
batch=${2:-32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/prof_$tag
mkdir -p $o
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -- python3 bench.py --batch $batch --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras > $o/fetch.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -- python3 bench.py --batch $batch --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras > $o/write.log 2>&1 || exit 5
cp $o/fetch/*/*counter_collection.csv $o/fetch.csv
cp $o/write/*/*counter_collection.csv $o/write.csv
# bench.py reads profiles/hbm_traffic.json: written first so that the bench line carries this build's traffic
timeout -k 10 100 python3 tools/traffic.py $o/fetch.csv $o/write.csv $o/hbm_traffic.json $batch > /dev/null || exit 7
cp $o/hbm_traffic.json profiles/hbm_traffic.json
timeout -k 10 400 python3 bench.py --batch $batch > $o/bench.json 2> $o/bench.err || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --batch $batch --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $o/under_rocprof.json 2> $o/trace.err || exit 3
timeout -k 10 300 python3 tools/kernel_breakdown.py --batch $batch > $o/role_breakdown.txt 2>/dev/null || exit 6
cp $o/trace/*/*kernel_stats.csv $o/kernel_stats.csv
rm -rf $o/trace $o/fetch $o/write $o/fetch.csv $o/write.csv
ls -la $o
