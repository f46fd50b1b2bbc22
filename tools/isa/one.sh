#!/bin/bash
# usage: tools/isa/one.sh [out.s] [extra -D flags]   -- ISA + statistics of one wave-kernel instantiation (default: the headline one)
out=${1:-/tmp/one.s}; shift
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S "$@" \
    -I tardis_amd/csrc -o "$out" tools/isa/one_kernel.hip 2>&1 | grep -v hip-link
python3 tools/isa_stats.py "$out" propagate_wave_kernel
