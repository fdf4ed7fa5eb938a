#!/bin/bash
# Runs on the GPU box (gpurun): bench line, rocprofv3 kernel stats of the same command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, no tracing domains besides --kernel-trace) -> gpurun_out/r01_*
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write
timeout -k 10 300 python bench.py > $O/r01_bench.json 2> $O/r01_bench.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python bench.py --no-cpu-baseline \
    > $O/r01_bench_under_rocprof.json 2> $O/prof_bench.err || exit 1
echo "kernel trace done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 \
    > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
echo "pmc fetch done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 \
    > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
echo "pmc write done"
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r01_pmc_traffic.json > $O/pmc_summary.txt
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/r01_bench_kernel_stats.csv
cat $O/r01_bench.json
