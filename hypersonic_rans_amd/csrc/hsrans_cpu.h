// Host SIMD decoders with runtime dispatch (hsrans_cpu.cpp): the CPU counterpart used by the `*_decode_auto_N` drop-in
// entries for single-chain streams, by the host index builder and as the in-run CPU comparator.  Never used by the GPU entries.
#ifndef HSRANS_CPU_H
#define HSRANS_CPU_H

#include <stddef.h>
#include <stdint.h>

namespace hsrans
{
namespace cpu
{

constexpr int kLevelScalar = 0, kLevelAvx2 = 1, kLevelAvx512 = 2;

int best_level();                    // what this host supports (CPUID, once)
const char *level_name(int level);
// every function takes `level` = the widest instruction set to use (clamped to best_level()) and `threads` (>= 1)
size_t decode(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap);
size_t exec_plan(int level, uint32_t threads, const uint8_t *plan, size_t plan_size, const uint8_t *stream, size_t stream_len, uint8_t *out, size_t out_cap);
size_t index_build(int level, uint32_t threads, int container, int states, uint32_t bits, const uint8_t *in, size_t in_len, const uint64_t *groups, size_t n_groups,
                   uint8_t *plan_out, size_t plan_cap, uint32_t uniform_interval = 0);

} // namespace cpu
} // namespace hsrans

#endif // HSRANS_CPU_H
