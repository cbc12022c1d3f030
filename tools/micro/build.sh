#!/bin/bash
# builds the stand-alone micro-benchmarks (development tools; binaries are git-ignored and travel with gpurun)
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
$H --offload-arch=gfx950 -O3 -std=c++17 -DMSCNN_WGEMM_DEV -I../../mscnn_amd/csrc -Rpass-analysis=kernel-resource-usage wgemm_bench.hip ../../mscnn_amd/csrc/wgemm.hip ../../mscnn_amd/csrc/common.cpp -x hip -o wgemm_bench > /tmp/build.log 2>&1
grep -E "error" /tmp/build.log | head
grep -E "Function Name|VGPRs:|ScratchSize" /tmp/build.log | grep -A2 wgemm_kernel | grep -v "^--" | sed 's/.*wgemm_kernelINS_4WCfgI//;s/.*remark: *//;s/ \[-Rpass.*//' | paste - - - | cut -c1-110
for f in mfma_clock lds_dma_probe; do [ $f -nt $f.hip ] || $H --offload-arch=gfx950 -O3 $f.hip -o $f 2>/dev/null; done
# round 4: the fused ROI pooling and the chained transform kernels against their unfused references
$H --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mscnn_amd/csrc roipool_wino_check.hip ../../mscnn_amd/csrc/roipool_wino.hip ../../mscnn_amd/csrc/common.cpp -o roipool_wino_check 2>&1 | grep error
$H --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mscnn_amd/csrc wino_outin_check.hip ../../mscnn_amd/csrc/winograd.hip ../../mscnn_amd/csrc/common.cpp -o wino_outin_check 2>&1 | grep error
