#!/bin/bash
# Build libspeech + the driver with AddressSanitizer on the HOST code only and run it (on the GPU box: gpurun -- bash tools/asan/run.sh).
# usage: tools/asan/run.sh build   (here, cross-compiling)      tools/asan/run.sh run   (on the box)
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/asan/build
if [ "${1:-build}" = build ]; then
  mkdir -p $OUT
  for f in speech_api generic_kernels fast_kernels fast_spectrogram fast_mel fast_mel_xwide post_kernels host_pipeline synth_kernels; do
    extra=""; case $f in fast_spectrogram|fast_mel|fast_mel_xwide) extra="-fno-slp-vectorize -fno-signed-zeros";; esac
    /opt/rocm/bin/hipcc -std=c++17 -O1 -g -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer $extra -c speech_amd/csrc/$f.hip -o $OUT/$f.o
  done
  /opt/rocm/bin/hipcc -std=c++17 -O1 -g --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -x hip -c tools/asan/host_asan.cpp -o $OUT/host_asan.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o $OUT/host_asan $OUT/host_asan.o $OUT/speech_api.o $OUT/generic_kernels.o $OUT/fast_kernels.o $OUT/fast_spectrogram.o $OUT/fast_mel.o $OUT/fast_mel_xwide.o $OUT/post_kernels.o $OUT/host_pipeline.o $OUT/synth_kernels.o
  echo built $OUT/host_asan
else
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 timeout -k 10 300 $OUT/host_asan
  # second pass with the quarantine off: the bounded-device-memory check over 400 option sets (see host_asan.cpp)
  HOST_ASAN_MEMCHECK=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0:quarantine_size_mb=0:thread_local_quarantine_size_kb=0 timeout -k 10 300 $OUT/host_asan
fi
