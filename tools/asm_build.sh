#!/bin/bash
# Developer aid: device-only assembly listing of the library (what tests/test_isa_hazards.py reads), into $1 (default /tmp/fx.s)
rm -f "${1:-/tmp/fx.s}"
cd "$(dirname "$0")/../effex_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o "${1:-/tmp/fx.s}" fxcorr.hip
