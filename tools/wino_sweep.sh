#!/bin/bash
# Per-layer times of the nine 3x3 stride-1 shapes at batch 2: the direct split-f16x3 kernel, the Winograd form, and the
# Winograd form's timing-only builds (`make -C geo-trax_amd winoprobe`). Output -> profiles/r05_winograd_probe.txt
cd "$(dirname "$0")/.."
echo "== direct (GTX_WINO=0)"; GTX_WINO=0 python tools/conv_sweep.py 1920 2 k3s1 f32s
echo "== winograd (GTX_WINO=1)"; GTX_WINO=1 python tools/conv_sweep.py 1920 2 k3s1 f32s
for v in ${WINO_PROBES:-1 2 4 6 8 16 24 32 64 88}; do
  f=geo-trax_amd/geotrax_amd/libgtx_winoprobe$v.so
  [ -f $f ] || continue
  echo "== winograd, timing-only build GTXW_PROBE=$v (bits: 1 no transform arithmetic, 2 no MFMAs, 4 no transform at all, 8 no weight loads in the loop, 16 no patch loads / commits in the loop, 32 no epilogue, 64 no barrier in the loop)"
  GTX_LIB=$PWD/$f GTX_WINO=1 python tools/conv_sweep.py 1920 2 k3s1 f32s
done
