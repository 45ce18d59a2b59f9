// Host-side format layer: histogram normalisation, scalar encoders and the decode planner.
// The C ABI in include/hsrans_hip.h is a thin veneer over these.
#ifndef HSRANS_HOST_H
#define HSRANS_HOST_H

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/hsrans_hip.h"
#include "hsrans_plan.h"

namespace hsrans
{

constexpr uint32_t kConsumePoint16 = 1u << 15; // reference rans.h:8

inline bool valid_codec(int container, int states, uint32_t bits)
{
  return container >= HSRANS_RAW && container <= HSRANS_MT && (states == 32 || states == 64) && bits >= 10 && bits <= 15;
}

size_t capacity(int container, int states, size_t n);
void make_hist(hsrans_hist *hist, const uint8_t *data, size_t size, uint32_t bits);
void normalize_counts(hsrans_hist *hist, const uint32_t raw[256], size_t data_bytes, uint32_t bits);

size_t encode(int container, int states, uint32_t bits, const uint8_t *in, size_t n, uint8_t *out, size_t cap, const hsrans_hist *hist,
              hsrans_encode_opts *opts);

// plan of a raw stream from its encoder's checkpoints (ascending by group; states n_ck x `states`): what encode() emits
size_t raw_plan_from_checkpoints(int states, uint32_t bits, uint64_t n, uint64_t total, const uint16_t counts[256], const uint32_t *start_states, size_t n_ck,
                                 const uint64_t *ck_group, const uint64_t *ck_words_from_end, const uint32_t *ck_states, uint32_t interval, uint8_t *plan_out,
                                 size_t plan_capacity);

// in-memory plan under construction
struct PlanBuilder
{
  PlanHeader hdr{};
  std::vector<uint32_t> chain_first; // n_chains entries while building (+1 sentinel on serialise)
  std::vector<Piece> pieces;
  std::vector<uint32_t> states; // n_chains * S
  uint16_t hist_counts[256] = {}; // copy of the shared histogram (kPlanHasHist), set with set_hist()
  bool has_hist = false;
  void set_hist(const uint16_t counts[256]);

  void begin(int container, int states, uint32_t bits, uint64_t decoded_len, uint64_t stream_len);
  // starts a new chain whose first piece is `p` with start states `st` (S values; may be null for fills)
  void add_chain(const Piece &p, const uint32_t *st);
  void add_piece(const Piece &p); // continuation piece of the current chain
  void reserve(size_t chains);    // (after begin(): room for that many single-piece chains)
  size_t serialized_size() const;
  size_t serialize(uint8_t *out, size_t cap); // fills shared_hist / aux_off, returns bytes or 0
};

// plan derived from the stream alone (mirrors the control flow of the reference decoders; see hsrans_host.cpp)
size_t plan_build(int container, int states, uint32_t bits, const uint8_t *stream, size_t stream_len, size_t out_cap, uint8_t *plan_out, size_t plan_cap);
bool plan_build_vec(int container, int states, uint32_t bits, const uint8_t *stream, size_t stream_len, size_t out_cap, std::vector<uint8_t> *plan);
bool plan_validate(const uint8_t *plan, size_t plan_size, uint64_t stream_len, uint64_t out_cap);
size_t plan_slice(const uint8_t *plan, size_t plan_size, uint32_t first, uint32_t count, uint8_t *out, size_t cap);
bool plan_chain_range(const uint8_t *plan, size_t plan_size, uint32_t first, uint32_t count, uint64_t *begin, uint64_t *end);
bool plan_stream_ranges(const uint8_t *plan, size_t plan_size, uint32_t first, uint32_t count, uint64_t out[4]);
size_t plan_capacity(int container, int states, size_t decoded_size, uint32_t interval, uint32_t block_size);
size_t plan_capacity_chains(int container, int states, size_t decoded_size, size_t extra_chains, uint32_t block_size);
// hsrans_shard_layout's body (pure host arithmetic; see include/hsrans_hip.h)
int shard_layout(const uint8_t *plan, size_t plan_size, uint32_t world, uint32_t parts, const double *weights, hsrans_shard *shards, uint64_t *windows);
size_t plan_thin(const uint8_t *plan, size_t plan_size, const uint64_t *groups, size_t n_groups, uint8_t *out, size_t cap);

} // namespace hsrans

#endif // HSRANS_HOST_H
