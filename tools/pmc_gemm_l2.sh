#!/bin/bash
# Runs on the GPU box: L2 hit/miss and fabric bytes of the GEMM kernels on the ViT-B/32 batch-256 shapes
# (rocprofv3 --pmc in its own runs, kernel trace only), plus the L2 / Infinity-Cache / HBM fetch-rate probe.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result tools/micro/l2_fetch_probe.hip -o $O/l2_fetch_probe && $O/l2_fetch_probe > $O/l2_fetch_probe.txt 2>&1
cat $O/l2_fetch_probe.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_')
  rm -rf $O/pmc_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -- python tools/time_gemm.py > $O/pmc_$tag.log 2>&1 || echo "pmc $set failed"
  for k in "persist_kernel<0" "persist_kernel<1" "bf16_kernel<2" "bf16_kernel<3"; do echo "== $set :: kernel ~ $k"; python tools/pmc_kernel.py $O/pmc_$tag "$k"; done
done
