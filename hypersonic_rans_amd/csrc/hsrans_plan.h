// Decode-plan layout shared by the host planner (hsrans_host.cpp) and the gfx950 kernels (hsrans_kernels.hip).
//
// A plan is one little-endian byte blob:
//   PlanHeader (64 B) | uint32 chain_first[n_chains + 1] (padded to 16 B) | Piece pieces[n_pieces] | uint32 states[n_chains * S]
//   | optional uint16 symbolCount[256] (kPlanHasHist)
// A *chain* is what one wavefront decodes: it loads S start states, then runs its pieces in order, carrying the
// states from piece to piece.  A *piece* is a run of whole S-symbol groups (+ an optional final masked group) decoded
// with one histogram from one contiguous span of uint16 words, or a single-symbol fill.
//
// Where the fields come from in the reference (/root/reference/src):
//   states      raw: stream+528 (rANS32x64_16w.cpp:200-206); mt_: block header (mt_rANS32x64_16w_decode.cpp:62-66);
//               block_: stream+16 (block_rANS32x64_16w_decode.cpp:36-40); checkpoints: the encoder's states at a group boundary
//   words_off   position of pReadHead when the piece starts (rANS32x64_16w.cpp:208, mt_…decode.cpp:72)
//   hist_off    position of the 256 uint16 symbol counts (rANS32x64_16w.cpp:191-195, mt_…decode.cpp:68-72, block_…decode.cpp:63-67)
//   steps/tail  loop trip count of decode_section (block_codec64.h:182) / lanes of the final partial group (rANS32x64_16w.cpp:252-280)
#ifndef HSRANS_PLAN_H
#define HSRANS_PLAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsrans
{

constexpr uint32_t kPieceChainStart = 1; // load start states plan.states[state_idx * S ...] before this piece
constexpr uint32_t kPieceFill = 2;       // single-symbol block: fill_len bytes of (hist_off & 0xFF); states untouched

constexpr uint32_t kPlanWalk = 1; // block_ without checkpoints: the kernel parses the inline headers itself
// every chain is one rANS piece of the same histogram and chain i+1 starts exactly where chain i stops (output AND word
// stream): any run of consecutive chains can be decoded as one longer chain from the first one's start states.  The
// kernel uses this to hand every resident wavefront one contiguous, equally long share of the stream.
constexpr uint32_t kPlanMergeable = 2;
// the plan carries a copy of the one histogram all its chains use (uint16 symbolCount[256] after the states): lets the
// host prepare the decode table once per plan instead of every workgroup rebuilding it on every launch.  The kernel
// still compares the copy with the counts in the stream it is given and flags a mismatch.
constexpr uint32_t kPlanHasHist = 4;

struct Piece
{
  uint64_t words_off; // byte offset in the stream of the first uint16 word (even)
  uint64_t out_off;   // byte offset in the output of the first symbol
  uint64_t hist_off;  // byte offset in the stream of uint16 symbolCount[256]; fill: the symbol
  uint64_t fill_len;  // kPieceFill only
  uint32_t steps;     // whole groups of S symbols
  uint16_t tail;      // symbols in the final masked group (0 .. S-1)
  uint16_t flags;
  uint32_t state_idx; // kPieceChainStart only
  uint32_t reserved;
};
static_assert(sizeof(Piece) == 48, "Piece layout");

struct PlanHeader
{
  char magic[8]; // "HSRPLAN1"
  uint32_t container, states, bits, flags;
  uint64_t decoded_len, stream_len;
  uint32_t n_chains, n_pieces;
  uint32_t shared_hist; // 1: every non-fill piece has the same hist_off (one table per workgroup)
  uint32_t interval;    // checkpoint interval in groups (0 = none)
  uint64_t aux_off;     // kPlanWalk: byte offset of the first inline block header; shared_hist: the shared hist_off
};
static_assert(sizeof(PlanHeader) == 64, "PlanHeader layout");

__host__ __device__ inline uint64_t plan_chain_first_off() { return sizeof(PlanHeader); }
__host__ __device__ inline uint64_t plan_pieces_off(uint32_t n_chains) { return sizeof(PlanHeader) + ((uint64_t(n_chains) + 1) * 4 + 15) / 16 * 16; }
__host__ __device__ inline uint64_t plan_states_off(uint32_t n_chains, uint32_t n_pieces) { return plan_pieces_off(n_chains) + uint64_t(n_pieces) * sizeof(Piece); }
__host__ __device__ inline uint64_t plan_hist_off(uint32_t n_chains, uint32_t n_pieces, uint32_t S) { return plan_states_off(n_chains, n_pieces) + uint64_t(n_chains) * S * 4; }
__host__ __device__ inline uint64_t plan_size(uint32_t n_chains, uint32_t n_pieces, uint32_t S, uint32_t flags)
{
  return plan_hist_off(n_chains, n_pieces, S) + ((flags & kPlanHasHist) ? 512 : 0);
}

// device status bits (hsrans_dplan status word)
constexpr uint32_t kStatusBadHist = 1;   // counts do not sum to 1 << bits (hist.cpp:308-324 returns false)
constexpr uint32_t kStatusBadBlock = 2;  // block end not a multiple of S (block_…decode.cpp:76-77)
constexpr uint32_t kStatusOutOfRange = 4; // a header / fill points outside the buffers

} // namespace hsrans

#endif // HSRANS_PLAN_H
