#!/bin/bash
# Hough batch (BASELINE config 4) on ONE GPU box: the shipped library (int32 accumulator) against a build with the
# 16-bit accumulator, each over a sweep of scans in flight; then larger batches.  Writes gpurun_out/<tag>/hough_ab.log.
# Usage (on the box): bash tools/hough_ab.sh r03
set -e
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
SRCS="kernels.hip runs.hip stages.hip deskew.hip hough.hip fft.hip fft_mixed.hip engine.cpp oics_host.cpp oics_hough.cpp oics_fft.cpp"
(cd omr-img-corrector_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    -Wno-unused-result -DOMR_PPHT_U16_MAX_EXTENT=16000 -shared -o "$OLDPWD/$OUT/libomrdeskew_u16.so" $SRCS) > "$OUT/hough_ab_build.log" 2>&1
L="$OUT/hough_ab.log"
echo "== int32 accumulator (the shipped library), 256 scans" > "$L"
timeout -k 10 400 python3 tools/bench_hough.py 256 8 2 64,128,192,256 >> "$L" 2>&1
echo "== 16-bit accumulator (-DOMR_PPHT_U16_MAX_EXTENT=16000), 256 scans" >> "$L"
OMR_AB_LIB="$OUT/libomrdeskew_u16.so" timeout -k 10 400 python3 tools/ab_lib.py tools/bench_hough.py 256 8 2 64,128,192,256 >> "$L" 2>&1
echo "== int32, 512 scans" >> "$L"
timeout -k 10 400 python3 tools/bench_hough.py 512 8 1 256,512 >> "$L" 2>&1
echo "== int32, 1024 scans" >> "$L"
timeout -k 10 400 python3 tools/bench_hough.py 1024 8 1 512,1024 >> "$L" 2>&1
rm -f "$OUT/libomrdeskew_u16.so"
