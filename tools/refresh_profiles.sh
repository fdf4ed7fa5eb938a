#!/bin/bash
# Runs on the GPU box (gpurun): everything under profiles/ for one round -> gpurun_out/${R}_*
#   bench line (cfg2), rocprofv3 kernel stats of the same command, the two PMC traffic passes (FETCH_SIZE / WRITE_SIZE,
#   separate runs, kernel trace only), bench lines of the other BASELINE configs at N = 1, small-batch latency with and
#   without hipGraph replay, the L2 / Infinity-Cache / HBM fetch-rate probe with the GEMM kernels' L2 hit/miss counters,
#   and the in-kernel phase breakdown of the GEMM tiles (diagnostic build, if tools/libmmr_hip_stamps.so is present).
R=${R:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write
timeout -k 10 300 python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err || exit 1
echo "bench done"
# kernel stats of the bench command with ONE step in flight (what roofline.avg_launch_us is measured on), then of the default
# command (two steps in flight: a launch's duration includes the time it shares the chip with the other lane's kernels)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python bench.py --lanes 1 --no-cpu-baseline --no-sweep \
    > $O/${R}_bench_under_rocprof.json 2> $O/prof_bench.err || exit 1
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/${R}_bench_kernel_stats.csv
rm -rf $O/prof_bench2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench2 -- python bench.py --no-cpu-baseline --no-sweep \
    > $O/${R}_bench_lanes2_under_rocprof.json 2> $O/prof_bench2.err || exit 1
cp $(ls $O/prof_bench2/*/*kernel_stats.csv | head -1) $O/${R}_bench_lanes2_kernel_stats.csv
echo "kernel trace done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --lanes 1 --no-cpu-baseline --no-sweep --steps 5 --warmup 2 \
    > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
echo "pmc fetch done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --lanes 1 --no-cpu-baseline --no-sweep --steps 5 --warmup 2 \
    > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
echo "pmc write done"
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/${R}_pmc_traffic.json > $O/pmc_summary.txt
for c in cfg3 cfg4 cfg5; do
  timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 3 > $O/${R}_bench_$c.json 2> $O/${R}_bench_$c.err || echo "bench $c failed"
  echo "bench $c done"
done
# the N > 1 code path (process group, RCCL all-gather, pipelined steps) with ONE self-spawned rank, and the gallery build leg
timeout -k 10 300 python bench.py --gpus 1 --spawn --steps 20 --warmup 5 > $O/${R}_bench_spawn1.json 2> $O/${R}_bench_spawn1.err || echo "bench spawn1 failed"
timeout -k 10 300 python bench.py --config build --steps 20 --warmup 3 > $O/${R}_bench_build.json 2> $O/${R}_bench_build.err || echo "bench build failed"
echo "spawn1 + build done"
rm -rf $O/prof_cfg5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python bench.py --lanes 1 --config cfg5 --steps 5 --warmup 2 > /dev/null 2> $O/prof_cfg5.err \
    && cp $(ls $O/prof_cfg5/*/*kernel_stats.csv | head -1) $O/${R}_bench_cfg5_kernel_stats.csv
timeout -k 10 200 python tools/time_small_batch.py > $O/${R}_small_batch_latency.json 2> $O/small_batch.err || echo "small batch failed"
echo "small batch done"
./tools/pmc_gemm_l2.sh > $O/${R}_gemm_l2_hit_miss.txt 2>&1
cp $O/l2_fetch_probe.txt $O/${R}_l2_fetch_probe.txt
if [ -f tools/libmmr_hip_stamps.so ]; then
  echo "=== persistent workgroups, full 256x256 tiles only (MMR_GEMM_HALF_PANELS=0)" > $O/${R}_gemm_phase_times.txt
  MMR_LIB=$PWD/tools/libmmr_hip_stamps.so MMR_GEMM_HALF_PANELS=0 python tools/gemm_phase_times.py >> $O/${R}_gemm_phase_times.txt 2>&1
  echo "=== default launch policy (persistent workgroups over a full + half tile list where chosen)" >> $O/${R}_gemm_phase_times.txt
  MMR_LIB=$PWD/tools/libmmr_hip_stamps.so python tools/gemm_phase_times.py >> $O/${R}_gemm_phase_times.txt 2>&1
fi
(echo "# fp32 gallery through GalleryIndex (split kept, tiered search)"; python tools/time_search_fp32.py; echo "# same, MMR_SPLIT_TIERS=0 (three-product scan for every query)"; MMR_SPLIT_TIERS=0 NS=1000000 python tools/time_search_fp32.py; echo "# bf16 galleries: E = 768, then E = 512"; E=768 QS=1,32,128,256 python tools/time_search.py; QS=1,32,64,128,256,1024 python tools/time_search.py) > $O/${R}_search_timings.txt 2>&1
cat $O/${R}_bench.json
