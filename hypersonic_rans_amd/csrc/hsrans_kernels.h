// Launch interface between the C ABI (hsrans_capi.cpp) and the gfx950 kernels (hsrans_kernels.hip).
#ifndef HSRANS_KERNELS_H
#define HSRANS_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hsrans_plan.h"

namespace hsrans
{

struct KParams
{
  const uint8_t *stream; // device, 16-byte aligned
  uint64_t stream_len;
  uint8_t *out; // device, 4-byte aligned
  uint64_t out_cap;
  const uint8_t *plan; // device copy of the plan blob
  uint32_t *status;    // device status word (kStatus* bits)
  // index-build pass only (ckpt_interval != 0): checkpoint g / ckpt_interval receives the S states and the absolute
  // byte position of the read cursor at group boundary g
  uint32_t *ckpt_states;
  uint64_t *ckpt_words;
  uint32_t ckpt_interval;
};

struct LaunchInfo
{
  uint32_t grid, block, lds_bytes, waves_per_block, chains, shared_table, walk, two_level;
};

// one-time per process: raise the dynamic-LDS limit of every kernel variant to the gfx950 maximum (160 KiB)
hipError_t prepare_kernels();
// asynchronous on `stream`; no allocation, no synchronisation (graph-capturable)
hipError_t launch_decode(const KParams &kp, const PlanHeader &h, hipStream_t stream, LaunchInfo *info);

} // namespace hsrans

#endif // HSRANS_KERNELS_H
