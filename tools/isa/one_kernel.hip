// Compiles ONE instantiation of the wave kernel so that its ISA can be read in seconds instead of a minute:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S \
//         -DK_FULL=false -DK_TRACK=true -DK_G=16 -DK_VPK=false -DK_LS=true -I tardis_amd/csrc -o /tmp/one.s tools/isa/one_kernel.hip
//   python3 tools/isa_stats.py /tmp/one.s propagate_wave_kernel
#include <hip/hip_runtime.h>
#include "mc_device.hpp"
#include "propagate_group.hpp"
#include "estimator_log.hpp"
#include "propagate_wave.hpp"
#ifndef K_FULL
#define K_FULL false
#define K_TRACK true
#define K_G 16
#define K_VPK false
#define K_LS true
#endif
#ifndef K_XWALK
#define K_XWALK true
#endif
#ifndef K_WPE
#define K_WPE (K_VPK ? 3 : 4)
#endif
#ifndef K_NT
#define K_NT 0
#endif
#ifndef K_SL
#define K_SL false
#endif
template __global__ void mc::propagate_wave_kernel<K_FULL, K_TRACK, K_G, K_VPK, K_LS, K_XWALK, K_WPE, K_NT, K_SL>(mc::WaveHot, const mc::WaveCold *);
