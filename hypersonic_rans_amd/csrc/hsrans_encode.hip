// gfx950 encoder for mt_ streams with independent fixed-size blocks (include/hsrans_hip.h: hsrans_encode_device).
//
// The reference's encoders are scalar CPU loops (src/mt_rANS32x64_16w_encode.cpp:140-356); their only serial dependency
// between blocks is the coder state that is carried across block boundaries.  With HSRANS_ENC_INDEPENDENT_BLOCKS every
// block starts from fresh states, so a block is one self-contained job:
//
//   K_hist   byte counts of every block, a workgroup per block (32 LDS copies laid out [symbol][copy]: no bank conflicts)
//   K_enc    one wavefront per block: normalisation identical to hist.cpp:16-215 (the heap sort's extractions as straight-line
//            scalar code) -> backward rANS pass, lane j = coder state j, a set of four groups as one asm statement, words through an
//            LDS ring into the block's scratch slot top-down, then the block header [size][skip][states][counts] in front of them:
//            the slot ends with the block's image.  Written for its INSTRUCTION COUNT: a wavefront alone on its SIMD pays about
//            2.3 ns per instruction, whatever it is (tools/microbench/lone_wave.hip)
//   K_gather adds up the image sizes in front of its block (K_scan does beyond 4,096 blocks), copies the image to its place; the last
//            workgroup writes the file header [n][total] and the result words (page-locked host memory)
//
// The result is byte-identical to the host encoder (hsrans_host.cpp encode(), same flag).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "hsrans_encode.h"
#include "hsrans_kernels.h"

namespace hsrans
{
namespace
{

constexpr uint32_t kWavesPerWG = 1;       // one wavefront per workgroup: its LDS starts at address 0, so every LDS address of the pass is a constant plus the lane's part
// Input bytes fetched per step of the rANS pass (CHUNK): the staging ring — two chunks — is most of a wavefront's LDS.  4 KiB chunks
// (13.25 KiB a wavefront, 11 wavefronts per CU) while every block of the launch is resident at once anyway: a block is then alone with
// its chain and pays for every chunk change; 2 KiB chunks (9.25 KiB, 17 per CU) beyond that: 100 MB in 32 KiB blocks 442 -> 501 GB/s
// (3,052 blocks: one round instead of two), 256 MiB 615 -> 685, 2^30 bytes 777 -> 822 GB/s; 100 MB in 64 KiB blocks lose 1 % with them.
constexpr uint32_t kChunkFew = 4096, kChunkMany = 2048;
constexpr uint32_t kSubHists = 8;        // histogram copies (lane & 7) of a wavefront that counts its own bytes, laid out [symbol][copy]
// The emitted words of the rANS pass go through an LDS ring and leave it in whole segments (one 8-byte store per lane) instead of
// one masked 2-byte global store per group: a group emits at most 128 bytes, a set of four groups at most one segment, so with a
// check after every set two segments are all the ring needs (byte o of the block's slot sits at ring offset o mod kOutRing)
constexpr uint32_t kOutSeg = 512;
constexpr uint32_t kOutRing = 2 * kOutSeg;

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }
__device__ __forceinline__ uint32_t enc_lane_to_byte(uint32_t j) { return (j & 0x23u) | ((j & 0x04u) << 2) | ((j & 0x18u) >> 1); }
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct __attribute__((packed, aligned(2))) U64a2
{
  uint64_t v;
};
struct __attribute__((packed, aligned(2))) U32a2
{
  uint32_t v;
};

template <uint32_t CHUNK>
struct WaveLdsT
{
  uint32_t order[256];      // count << 8 | symbol in heap-sort order; after the normalisation: the kOutRing bytes of the emitted-word ring
  uint4 table[256];         // {x_max, bias, rcp, cmpl | shift << 24}
  uint8_t stage[2 * CHUNK]; // the input ring (table and stage double as the kSubHists histogram copies before the table exists)
  uint32_t sink[64];        // where the lanes that emit nothing in a group put their write (a select of the address is cheaper than two writes of EXEC)
};
static_assert(sizeof(uint4) * 256 + 2 * kChunkMany >= kSubHists * 256 * 4, "histogram copies must fit");
static_assert(kOutRing == sizeof(uint32_t) * 256, "the emitted-word ring is the sort's order array");

// ---- hist.cpp:16-215: the heap sort ---------------------------------------------------------------------------------
// The reference sorts the symbols by count with a textbook heap sort; the order of EQUAL counts that sort happens to
// produce decides which symbols are adjusted, so the sort is replayed exactly (hsrans_host.cpp heap_sift) — but as
// wave-uniform code: the heap lives in five registers by tree level (entries count << 8 | symbol),
//   A: nodes 0..62 (levels 0-5, lane = node), B: 63..126 (level 6), C: 127..190, D: 191..254 (level 7), E: node 255,
// read with v_readlane and written with a one-lane select, all control flow scalar.  ~4x faster than one lane walking LDS.
struct Heap
{
  uint32_t A, B, C, D, E;
};

// one lane of a register takes a wave-uniform value (v_cmp + v_cndmask: as cheap as v_writelane through M0, and the
// compiler keeps track of the hazards)
__device__ __forceinline__ uint32_t write_lane(uint32_t value, uint32_t lane, uint32_t reg) { return lane_id() == lane ? value : reg; }

template <int LEVEL>
__device__ __forceinline__ uint32_t heap_get(const Heap &h, uint32_t i)
{
  if constexpr (LEVEL <= 5)
    return __builtin_amdgcn_readlane(h.A, i);
  else if constexpr (LEVEL == 6)
    return __builtin_amdgcn_readlane(h.B, i - 63);
  else if constexpr (LEVEL == 7)
    return i < 191 ? __builtin_amdgcn_readlane(h.C, i - 127) : __builtin_amdgcn_readlane(h.D, i - 191);
  else
    return h.E;
}

template <int LEVEL>
__device__ __forceinline__ void heap_set(Heap &h, uint32_t i, uint32_t v)
{
  if constexpr (LEVEL <= 5)
    h.A = write_lane(v, i, h.A);
  else if constexpr (LEVEL == 6)
    h.B = write_lane(v, i - 63, h.B);
  else if constexpr (LEVEL == 7)
  {
    if (i < 191)
      h.C = write_lane(v, i - 127, h.C);
    else
      h.D = write_lane(v, i - 191, h.D);
  }
  else
    h.E = v;
}

// `val` belongs at node `root` (on level LEVEL) or below it; n = heap size
template <int LEVEL>
__device__ __forceinline__ void heap_sift(Heap &h, uint32_t root, uint32_t n, uint32_t val)
{
  if constexpr (LEVEL == 8)
    heap_set<8>(h, root, val);
  else
  {
    const uint32_t l = 2 * root + 1;
    if (l >= n)
    {
      heap_set<LEVEL>(h, root, val);
      return;
    }
    uint32_t big = l;
    uint32_t vb = heap_get<LEVEL + 1>(h, l);
    if constexpr (LEVEL + 1 < 8)
      if (l + 1 < n)
      {
        const uint32_t ar = heap_get<LEVEL + 1>(h, l + 1);
        if ((ar >> 8) > (vb >> 8)) // the right child only wins when strictly larger
        {
          big = l + 1;
          vb = ar;
        }
      }
    if ((vb >> 8) > (val >> 8))
    {
      heap_set<LEVEL>(h, root, vb);
      heap_sift<LEVEL + 1>(h, big, n, val);
    }
    else
      heap_set<LEVEL>(h, root, val);
  }
}

template <int LEVEL>
__device__ __forceinline__ void heap_build_level(Heap &h, uint32_t first, uint32_t last)
{
  for (uint32_t i = last + 1; i-- > first;)
    heap_sift<LEVEL>(h, i, 256, heap_get<LEVEL>(h, i));
}

// ---- the extraction as straight-line scalar code -----------------------------------------------------------------------------
// One extraction is one sift of the heap's last entry down from the root, every step depends on the one before, and the wavefront
// pays about 2.3 ns per instruction whatever it is (tools/microbench/lone_wave.hip) — so this is written for its instruction count
// (the compiler's form of heap_sift<> above spends 22-24 instructions a level).  Bottom-up: walk down the path of the larger
// children to its end WITHOUT looking at the entry to place (8 instructions a level: two v_readlane, "right wins only when its
// count is strictly larger" as one s_or + s_cmp on count << 8 | symbol, two s_cselect), then find from the bottom where the entry
// belongs (it came from the bottom: usually one compare), then move the path up by one (s_mov m0 + v_writelane per level).
// Same heap as heap_sift<0>: the counts along the path never grow, so "the first node whose larger child is not larger than the
// entry" is found from either end.  Lanes: A holds nodes 0..62 (lane = node), B nodes 63..126 (lane = node - 63), C 127..190,
// D 191..254; cK = lane of the path's node on level K (level 7: 0..63 in C, 64..127 in D), bK = that node's entry.
#define HSRANS_HEAP_PICK_AT(REG, L, R, BN, CN) /* children at lanes L, R of REG: the larger one's entry -> BN, its lane -> CN */                    \
  "v_readlane_b32 %[sl], " REG ", " L "\n\t"                                                                                                      \
  "v_readlane_b32 %[sr], " REG ", " R "\n\t"                                                                                                      \
  "s_or_b32 " BN ", %[sl], 0xff\n\t"                                                                                                              \
  "s_cmp_gt_u32 %[sr], " BN "\n\t"                                                                                                                \
  "s_cselect_b32 " BN ", %[sr], %[sl]\n\t"                                                                                                        \
  "s_cselect_b32 " CN ", " R ", " L "\n\t"
#define HSRANS_HEAP_PICK(REG, BN, CN) HSRANS_HEAP_PICK_AT(REG, "%[l]", "%[r]", BN, CN)
#define HSRANS_HEAP_DOWN_A(CK, BN, CN) /* node at lane CK of A (levels 1..4): children at lanes 2 CK + 1, 2 CK + 2 of A */                        \
  "s_lshl1_add_u32 %[l], " CK ", 1\n\t"                                                                                                           \
  "s_add_u32 %[r], %[l], 1\n\t" HSRANS_HEAP_PICK("%[A]", BN, CN)
#define HSRANS_HEAP_HEAD /* the entry's compare key, levels 0..4 */                                                                           \
  "s_or_b32 %[vmax], %[val], 0xff\n\t" HSRANS_HEAP_PICK_AT("%[A]", "1", "2", "%[b1]", "%[c1]") HSRANS_HEAP_DOWN_A("%[c1]", "%[b2]", "%[c2]")         \
      HSRANS_HEAP_DOWN_A("%[c2]", "%[b3]", "%[c3]") HSRANS_HEAP_DOWN_A("%[c3]", "%[b4]", "%[c4]") HSRANS_HEAP_DOWN_A("%[c4]", "%[b5]", "%[c5]")
#define HSRANS_HEAP_CLIMB_FROM(K, BK, BNEXT, LOWER) /* the path ends on level K: does the entry go there? (else try level K - 1) */                \
  "bottom" K "_%=:\n\t"                                                                                                                           \
  "s_cmp_gt_u32 " BK ", %[vmax]\n\t"                                                                                                              \
  "s_cbranch_scc0 " LOWER "_%=\n\t"                                                                                                               \
  "s_mov_b32 " BNEXT ", %[val]\n\t"                                                                                                               \
  "s_branch w" K "_%=\n\t"
#define HSRANS_HEAP_WRITE_A(K, CK, BNEXT) "w" K "_%=:\n\ts_mov_b32 m0, " CK "\n\tv_writelane_b32 %[A], " BNEXT ", m0\n\t"
#define HSRANS_HEAP_WRITES /* levels 5 .. 0 take the entry of the level below (or the entry to place) */                                          \
  HSRANS_HEAP_WRITE_A("5", "%[c5]", "%[b6]") HSRANS_HEAP_WRITE_A("4", "%[c4]", "%[b5]") HSRANS_HEAP_WRITE_A("3", "%[c3]", "%[b4]")                \
  HSRANS_HEAP_WRITE_A("2", "%[c2]", "%[b3]") HSRANS_HEAP_WRITE_A("1", "%[c1]", "%[b2]")                                                           \
  "w0_%=:\n\tv_writelane_b32 %[A], %[b1], 0\n\t"                                                                                                  \
  "s_branch done_%=\n\t"
#define HSRANS_HEAP_CLIMBS /* the rarer ends of the climb */                                                                                       \
  HSRANS_HEAP_CLIMB_FROM("5", "%[b5]", "%[b6]", "bottom4") HSRANS_HEAP_CLIMB_FROM("4", "%[b4]", "%[b5]", "bottom3")                               \
  HSRANS_HEAP_CLIMB_FROM("3", "%[b3]", "%[b4]", "bottom2") HSRANS_HEAP_CLIMB_FROM("2", "%[b2]", "%[b3]", "bottom1")                               \
  HSRANS_HEAP_CLIMB_FROM("1", "%[b1]", "%[b2]", "bottom0")                                                                                        \
  "bottom0_%=:\n\t"                                                                                                                               \
  "s_mov_b32 %[b1], %[val]\n\t"                                                                                                                   \
  "s_branch w0_%=\n\t"                                                                                                                            \
  "done_%=:"
struct HeapScratch // (scalar temporaries of one extraction; the register allocator places them)
{
  uint32_t vmax, l, r, sl, sr, c1, c2, c3, c4, c5, c6, c7, b1, b2, b3, b4, b5, b6, b7;
};
#define HSRANS_HEAP_OPERANDS                                                                                                                       \
  [A] "+v"(h.A), [B] "+v"(h.B), [C] "+v"(h.C), [D] "+v"(h.D), [vmax] "=&s"(t.vmax), [l] "=&s"(t.l), [r] "=&s"(t.r), [sl] "=&s"(t.sl), \
      [sr] "=&s"(t.sr), [c1] "=&s"(t.c1), [c2] "=&s"(t.c2), [c3] "=&s"(t.c3), [c4] "=&s"(t.c4), [c5] "=&s"(t.c5), [c6] "=&s"(t.c6), [c7] "=&s"(t.c7),  \
      [b1] "=&s"(t.b1), [b2] "=&s"(t.b2), [b3] "=&s"(t.b3), [b4] "=&s"(t.b4), [b5] "=&s"(t.b5), [b6] "=&s"(t.b6), [b7] "=&s"(t.b7)

// heap of n entries, 127 <= n <= 255 (levels 0..6 whole, level 7 up to node n - 1): the maximum leaves, `val` (the entry that was at
// node n) is sifted down from the root
__device__ __forceinline__ void heap_extract_high(Heap &h, uint32_t n, uint32_t val)
{
  HeapScratch t;
  const uint32_t n7 = n - 127; // nodes of level 7 in the heap: their lanes in C / D (as 0..127) are below this
  asm volatile(HSRANS_HEAP_HEAD
               // level 5 (A, nodes 31..62) -> level 6 (B): children at lanes 2 c - 62, 2 c - 61
               "s_lshl1_add_u32 %[l], %[c5], -62\n\t"
               "s_add_u32 %[r], %[l], 1\n\t" HSRANS_HEAP_PICK("%[B]", "%[b6]", "%[c6]")
               // level 6 (B, lane c) -> level 7: children are level-7 entries 2 c and 2 c + 1 (C: 0..63, D: 64..127), if still in the heap
               "s_lshl_b32 %[l], %[c6], 1\n\t"
               "s_cmp_ge_u32 %[l], %[n7]\n\t"
               "s_cbranch_scc1 bottom6_%=\n\t"
               "s_add_u32 %[r], %[l], 1\n\t"
               "s_cmp_lt_u32 %[c6], 32\n\t"
               "s_cbranch_scc0 fromD_%=\n\t"
               "v_readlane_b32 %[sl], %[C], %[l]\n\t"
               "v_readlane_b32 %[sr], %[C], %[r]\n\t"
               "s_branch pick7_%=\n\t"
               "fromD_%=:\n\t"
               "s_sub_u32 %[b7], %[l], 64\n\t"
               "v_readlane_b32 %[sl], %[D], %[b7]\n\t"
               "s_add_u32 %[b7], %[b7], 1\n\t"
               "v_readlane_b32 %[sr], %[D], %[b7]\n\t"
               "pick7_%=:\n\t"
               "s_cmp_lt_u32 %[r], %[n7]\n\t" // the right child may already be out of the heap: then it never wins
               "s_cselect_b32 %[sr], %[sr], 0\n\t"
               "s_or_b32 %[b7], %[sl], 0xff\n\t"
               "s_cmp_gt_u32 %[sr], %[b7]\n\t"
               "s_cselect_b32 %[b7], %[sr], %[sl]\n\t"
               "s_cselect_b32 %[c7], %[r], %[l]\n\t"
               // the path ends on level 7: the common case first, straight down through the writes
               "s_cmp_gt_u32 %[b7], %[vmax]\n\t"
               "s_cbranch_scc0 bottom6_%=\n\t"
               "s_cmp_lt_u32 %[c7], 64\n\t"
               "s_cbranch_scc1 w7C_%=\n\t"
               "s_sub_u32 m0, %[c7], 64\n\t"
               "v_writelane_b32 %[D], %[val], m0\n\t"
               "s_branch w6_%=\n\t"
               "w7C_%=:\n\t"
               "s_mov_b32 m0, %[c7]\n\t"
               "v_writelane_b32 %[C], %[val], m0\n\t"
               "w6_%=:\n\t"
               "s_mov_b32 m0, %[c6]\n\t"
               "v_writelane_b32 %[B], %[b7], m0\n\t" HSRANS_HEAP_WRITES HSRANS_HEAP_CLIMB_FROM("6", "%[b6]", "%[b7]", "bottom5") HSRANS_HEAP_CLIMBS
               : HSRANS_HEAP_OPERANDS
               : [val] "s"(val), [n7] "s"(n7)
               : "scc");
}

// heap of n entries, 63 <= n <= 126 (levels 0..5 whole, level 6 up to node n - 1)
__device__ __forceinline__ void heap_extract_mid(Heap &h, uint32_t n, uint32_t val)
{
  HeapScratch t;
  const uint32_t n6 = n - 63; // nodes of level 6 in the heap: their lanes in B are below this
  asm volatile(HSRANS_HEAP_HEAD
               // level 5 (A, nodes 31..62) -> level 6 (B): children at lanes 2 c - 62, 2 c - 61, if still in the heap
               "s_lshl1_add_u32 %[l], %[c5], -62\n\t"
               "s_cmp_ge_u32 %[l], %[n6]\n\t"
               "s_cbranch_scc1 bottom5_%=\n\t"
               "s_add_u32 %[r], %[l], 1\n\t"
               "v_readlane_b32 %[sl], %[B], %[l]\n\t"
               "v_readlane_b32 %[sr], %[B], %[r]\n\t"
               "s_cmp_lt_u32 %[r], %[n6]\n\t"
               "s_cselect_b32 %[sr], %[sr], 0\n\t"
               "s_or_b32 %[b6], %[sl], 0xff\n\t"
               "s_cmp_gt_u32 %[sr], %[b6]\n\t"
               "s_cselect_b32 %[b6], %[sr], %[sl]\n\t"
               "s_cselect_b32 %[c6], %[r], %[l]\n\t"
               "s_cmp_gt_u32 %[b6], %[vmax]\n\t"
               "s_cbranch_scc0 bottom5_%=\n\t"
               "s_mov_b32 %[b7], %[val]\n\t"
               "s_mov_b32 m0, %[c6]\n\t"
               "v_writelane_b32 %[B], %[b7], m0\n\t" HSRANS_HEAP_WRITES HSRANS_HEAP_CLIMBS
               : HSRANS_HEAP_OPERANDS
               : [val] "s"(val), [n6] "s"(n6)
               : "scc");
}
// heap of n entries, 1 <= n <= 62 (every node in A; more than 192 of the 256 symbols extracted: near-uniform bytes): every level
// looks whether its children are still in the heap
#define HSRANS_HEAP_DOWN_CHECKED(K, CK, BN, CN) /* node at lane CK of A on level K: children 2 CK + 1, 2 CK + 2 if below n */                    \
  "s_lshl1_add_u32 %[l], " CK ", 1\n\t"                                                                                                           \
  "s_cmp_ge_u32 %[l], %[n]\n\t"                                                                                                                   \
  "s_cbranch_scc1 bottom" K "_%=\n\t"                                                                                                             \
  "s_add_u32 %[r], %[l], 1\n\t"                                                                                                                   \
  "v_readlane_b32 %[sl], %[A], %[l]\n\t"                                                                                                          \
  "v_readlane_b32 %[sr], %[A], %[r]\n\t"                                                                                                          \
  "s_cmp_lt_u32 %[r], %[n]\n\t"                                                                                                                   \
  "s_cselect_b32 %[sr], %[sr], 0\n\t"                                                                                                             \
  "s_or_b32 " BN ", %[sl], 0xff\n\t"                                                                                                              \
  "s_cmp_gt_u32 %[sr], " BN "\n\t"                                                                                                                \
  "s_cselect_b32 " BN ", %[sr], %[sl]\n\t"                                                                                                        \
  "s_cselect_b32 " CN ", %[r], %[l]\n\t"
__device__ __forceinline__ void heap_extract_low(Heap &h, uint32_t n, uint32_t val)
{
  HeapScratch t;
  asm volatile("s_or_b32 %[vmax], %[val], 0xff\n\t"
               "s_mov_b32 %[c1], 0\n\t" // (the root, as the lane the first step starts from)
               HSRANS_HEAP_DOWN_CHECKED("0", "%[c1]", "%[b1]", "%[c1]") HSRANS_HEAP_DOWN_CHECKED("1", "%[c1]", "%[b2]", "%[c2]")
               HSRANS_HEAP_DOWN_CHECKED("2", "%[c2]", "%[b3]", "%[c3]") HSRANS_HEAP_DOWN_CHECKED("3", "%[c3]", "%[b4]", "%[c4]")
               HSRANS_HEAP_DOWN_CHECKED("4", "%[c4]", "%[b5]", "%[c5]")
               // level 5 (nodes 31..62) has no children below 62: the path ends here at the latest
               "s_cmp_gt_u32 %[b5], %[vmax]\n\t"
               "s_cbranch_scc0 bottom4_%=\n\t"
               "s_mov_b32 %[b6], %[val]\n\t" HSRANS_HEAP_WRITES HSRANS_HEAP_CLIMBS
               : HSRANS_HEAP_OPERANDS
               : [val] "s"(val), [n] "s"(n)
               : "scc");
}
#undef HSRANS_HEAP_DOWN_CHECKED
#undef HSRANS_HEAP_PICK_AT
#undef HSRANS_HEAP_PICK
#undef HSRANS_HEAP_DOWN_A
#undef HSRANS_HEAP_HEAD
#undef HSRANS_HEAP_CLIMB_FROM
#undef HSRANS_HEAP_WRITE_A
#undef HSRANS_HEAP_WRITES
#undef HSRANS_HEAP_CLIMBS
#undef HSRANS_HEAP_OPERANDS

__device__ __forceinline__ uint32_t ballot_count(bool pred) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pred)); }

// Heap sort of L.order (count << 8 | symbol) as far as the adjustment needs it: the `take` largest entries in the order
// the reference's sort puts them at the top of its array.  Returns, per lane, a 4-bit mask of its symbols (4 * lane + k)
// that are among them.
template <bool PARALLEL_BUILD, class LDS> // (the raw format's kernel keeps the register form: its one wavefront does this once per 100 MB, and the LDS form costs its pass 6 % through the register allocation)
__device__ __forceinline__ uint32_t heap_take_largest(LDS &L, uint32_t lane, uint32_t take)
{
  Heap h;
  if constexpr (PARALLEL_BUILD)
  // The BUILD of the heap (the sort's first phase: sift nodes 127 ... 0 down, in that order) has a level's nodes working on
  // disjoint subtrees, and a node only needs the levels below it finished: all nodes of a level sift at once, one per lane, in
  // LDS — 2 + 3 + ... + 8 dependent steps instead of 128 sifts one after the other — and give the heap the serial order gives.
  // The extractions that follow depend on one another and stay in the registers.
  {
    uint32_t *ord = L.order;
    auto sift = [&](uint32_t root) { // (per lane; comparisons as heap_sift below: by count, the right child only wins when strictly larger)
      const uint32_t val = ord[root];
      uint32_t i = root;
      while (true)
      {
        const uint32_t l = 2 * i + 1;
        if (l >= 256)
          break;
        uint32_t big = l, vb = ord[l];
        if (l + 1 < 256)
        {
          const uint32_t ar = ord[l + 1];
          if ((ar >> 8) > (vb >> 8))
            big = l + 1, vb = ar;
        }
        if ((vb >> 8) <= (val >> 8))
          break;
        ord[i] = vb;
        i = big;
      }
      ord[i] = val;
    };
    if (lane == 0)
      sift(127); // the only level-7 node with a child
    wave_sync();
    sift(63 + lane); // level 6: nodes 63 .. 126
    wave_sync();
    for (uint32_t first = 31; ; first = (first - 1) / 2) // levels 5 .. 0: nodes first .. 2 * first
    {
      if (lane <= first)
        sift(first + lane);
      wave_sync();
      if (first == 0)
        break;
    }
  }
  h.A = L.order[lane < 63 ? lane : 62];
  h.B = L.order[63 + lane];
  h.C = L.order[127 + lane];
  h.D = L.order[191 + lane < 255 ? 191 + lane : 254];
  h.E = __builtin_amdgcn_readfirstlane(L.order[255]);
  if constexpr (!PARALLEL_BUILD)
  {
    heap_sift<7>(h, 127, 256, heap_get<7>(h, 127)); // the only level-7 node with a child
    heap_build_level<6>(h, 63, 126);
    heap_build_level<5>(h, 31, 62);
    heap_build_level<4>(h, 15, 30);
    heap_build_level<3>(h, 7, 14);
    heap_build_level<2>(h, 3, 6);
    heap_build_level<1>(h, 1, 2);
    heap_build_level<0>(h, 0, 0);
  }
  if (take == 0)
    return 0;
  // `take` extractions; i + 1 = the heap's size before the next one, whose last entry (node i) is the one sifted down.  Nothing is
  // recorded per extraction: what was taken is what is no longer among the heap's first i + 1 nodes (every symbol is in it once).
  uint32_t i = 255;
  heap_extract_high(h, i, h.E);
  i--, take--;
  auto run = [&](uint32_t lowest, auto &&extract) { // nodes i ... lowest leave the heap, as long as extractions are wanted
    uint32_t todo = take < i + 1 - lowest ? take : i + 1 - lowest;
    take -= todo;
    for (; todo != 0; todo--, i--)
      extract(i);
  };
  run(191, [&](uint32_t k) { heap_extract_high(h, k, __builtin_amdgcn_readlane(h.D, k - 191)); });
  run(127, [&](uint32_t k) { heap_extract_high(h, k, __builtin_amdgcn_readlane(h.C, k - 127)); });
  run(63, [&](uint32_t k) { heap_extract_mid(h, k, __builtin_amdgcn_readlane(h.B, k - 63)); });
  run(1, [&](uint32_t k) { heap_extract_low(h, k, __builtin_amdgcn_readlane(h.A, k)); }); // (more than 192 of the 256 symbols: near-uniform bytes)
  const uint32_t left = take != 0 ? 0 : i + 1; // (take still wanted with one node left: the root goes too)
  uint8_t *rem = (uint8_t *)L.table;         // (the table's space, not yet in use)
  ((uint32_t *)rem)[lane] = 0;
  wave_sync();
  if (lane < 63 && lane < left)
    rem[h.A & 0xFF] = 1;
  if (63 + lane < left)
    rem[h.B & 0xFF] = 1;
  if (127 + lane < left)
    rem[h.C & 0xFF] = 1;
  if (191 + lane < left) // (nodes 191 .. 254; node 255 left with the first extraction)
    rem[h.D & 0xFF] = 1;
  wave_sync();
  const uint32_t m4 = ((const uint32_t *)rem)[lane];
  const uint32_t taken = ~((m4 & 1) | ((m4 >> 7) & 2) | ((m4 >> 14) & 4) | ((m4 >> 21) & 8)) & 0xFu;
  return taken;
}

// The reference's fix-up of the scaled counts (hist.cpp:103-199; hsrans_host.cpp normalize_counts).  It sorts the symbols
// by count (heap sort) and then runs "steal" passes (sum too large: every symbol with count >= 2 gives one, smallest
// first, until the sum fits) or "charity" passes (sum too small: every symbol with count >= 2 gets one, largest first).
// Only the LAST pass depends on the order, and only through which symbols are on which side of one cut in the sorted
// array; everything else follows from the counts alone:
//   steal:   pass j takes one from every symbol with original count >= j + 1 (m_j of them) while more than m_j are still
//            missing; the final pass J takes from the e_J smallest of those m_J, i.e. from all of them except the
//            m_J - e_J largest of the sort;
//   charity: the symbols with count >= 2 (m of them, their number never changes) each get F = (e - 1) / m, then the
//            e - F * m largest of the sort get one more.
// So the heap sort only has to deliver its largest few entries (heap_take_largest), in exactly the order the reference's
// sort would put them (ties!).  Counts stay in registers: sc[k] is symbol 4 * lane + k.
template <bool PARALLEL_BUILD, class LDS>
__device__ __forceinline__ void adjust_counts(LDS &L, uint32_t lane, uint32_t (&sc)[4], uint32_t sum, uint32_t target)
{
  auto count_ge = [&](uint32_t t) {
    uint32_t n = 0;
    for (uint32_t k = 0; k < 4; k++)
      n += ballot_count(sc[k] >= t);
    return n;
  };
  // (skipping the sort when the cut between taken and spared symbols does not fall inside a run of equal counts — a bisection over
  // ballots finds that out — was measured: no gain, the sort is no longer what a block waits for)
  const uint32_t m = count_ge(2); // >= 1: 256 counts <= 1 cannot come from counts that scale to 2^bits >= 1024
  if (sum > target)
  {
    uint32_t e = sum - target, j = 1, mj = m;
    while (e > mj) // terminates: the symbols can give sum - 256 > e in total
    {
      e -= mj;
      j++;
      mj = count_ge(j + 1);
    }
    const uint32_t spared = heap_take_largest<PARALLEL_BUILD>(L, lane, mj - e);
    for (uint32_t k = 0; k < 4; k++)
    {
      const uint32_t c = sc[k];
      uint32_t give = c >= 2 ? (c - 1 < j - 1 ? c - 1 : j - 1) : 0;
      give += c >= j + 1 && !((spared >> k) & 1) ? 1 : 0;
      sc[k] = c - give;
    }
  }
  else
  {
    const uint32_t e = target - sum;
    const uint32_t full = (e - 1) / m;
    const uint32_t lucky = heap_take_largest<PARALLEL_BUILD>(L, lane, e - full * m);
    for (uint32_t k = 0; k < 4; k++)
      sc[k] += (sc[k] >= 2 ? full : 0) + ((lucky >> k) & 1);
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
  for (int d = 32; d >= 1; d >>= 1)
    v += __shfl_xor(v, d, 64);
  return __builtin_amdgcn_readfirstlane(v); // (every lane holds the total; said so, what is computed from it stays in scalar registers)
}

__device__ __forceinline__ uint4 load16_guarded(const uint8_t *in, uint64_t pos, uint64_t n)
{
  if (pos + 16 <= n)
    return *(const uint4 *)(in + pos);
  uint32_t w[4] = {0, 0, 0, 0};
  for (uint32_t k = 0; k < 16; k++)
    if (pos + k < n)
      w[k >> 2] |= (uint32_t)in[pos + k] << (8 * (k & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <uint32_t CHUNK>
struct ChunkT
{
  uint4 q[CHUNK / 1024];
};
template <uint32_t CHUNK>
__device__ __forceinline__ ChunkT<CHUNK> chunk_load(const uint8_t *in, uint64_t begin, uint64_t end, uint32_t c, uint32_t lane)
{
  ChunkT<CHUNK> r;
  const uint64_t at = begin + (uint64_t)c * CHUNK;
  if (at + CHUNK <= end) // (wave-uniform; all but a block's last chunk: plain loads instead of guarded ones)
  {
#pragma unroll
    for (uint32_t k = 0; k < CHUNK / 1024; k++)
      r.q[k] = *(const uint4 *)(in + at + k * 1024 + lane * 16);
    return r;
  }
#pragma unroll
  for (uint32_t k = 0; k < CHUNK / 1024; k++)
    r.q[k] = load16_guarded(in, at + k * 1024 + lane * 16, end);
  return r;
}
template <uint32_t CHUNK>
__device__ __forceinline__ void chunk_to_lds(WaveLdsT<CHUNK> &L, const ChunkT<CHUNK> &r, uint32_t c, uint32_t lane)
{
#pragma unroll
  for (uint32_t k = 0; k < CHUNK / 1024; k++)
    *(uint4 *)(L.stage + (c & 1) * CHUNK + k * 1024 + lane * 16) = r.q[k];
}

// one group, general form: lanes whose byte does not exist (the file's last, partial group) keep their state
template <uint32_t S, uint32_t CHUNK>
__device__ __forceinline__ void encode_group_slow(uint32_t &x, WaveLdsT<CHUNK> &L, uint32_t group_off, uint32_t valid, uint32_t &p, uint32_t lane, uint32_t byte_in_group)
{
  constexpr uint32_t kRing = 2 * CHUNK;
  const bool active = lane < S && byte_in_group < valid;
  const uint32_t sym = L.stage[(group_off + byte_in_group) & (kRing - 1)];
  const uint4 e = L.table[sym];
  const bool emit = active && x >= e.x;
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(emit);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
  p -= 2 * (uint32_t)__builtin_popcountll(mask);
  uint32_t v = x;
  if (emit)
  {
    *(uint16_t *)((uint8_t *)L.order + ((p + 2 * rank) & (kOutRing - 1))) = (uint16_t)x; // lane S-1's word goes last in memory (rANS32x64_16w.cpp:65-99)
    v = x >> 16;
  }
  const uint32_t q = __umulhi(v, e.z) >> (e.w >> 24);
  const uint32_t nx = __umul24(q, e.w) + v + e.y;
  x = active ? nx : x;
}

// Four whole groups (a set: groups 4t+3 ... 4t, coded in that order) with their table entries already in registers.
//
// ONE wavefront codes a block and every state is one dependent chain, so the wavefront is alone on its issue port and pays about
// 2.1-2.4 ns for EVERY instruction, whatever it is (tools/microbench/lone_wave.hip: dependent or independent VALU, SALU, s_nop and
// s_waitcnt alike) — the pass costs its instruction count.  Hence one asm statement per set (nothing of the compiler's between the
// groups) and per group the 14 instructions below: the cursor counts words (no shift of the popcount), the ring address is an
// add-shift and an and-or, lanes that emit nothing write to their own sink word through a select of the ADDRESS (two writes of
// EXEC around the store cost three instructions and their hazards), the renormalisation shift is the select itself
// (v_cndmask_b32_sdwa with src1_sel:WORD_1), the table's shift count is read out of byte 3 of its word by the shift.
// `pw`: word offset (from the slot) of the lowest word written; `ring`: LDS address of the emitted-word ring (kOutRing-aligned);
// `sink`: this lane's sink address.  S == 32: only lanes 0..31 hold states; EXEC is narrowed to them for the set.
#define HSRANS_ENC_GROUP(EX, EY, EZ, EW)                                                                                                  \
  "v_cmp_ge_u32 vcc, %[x], " EX "\n\t"                                                                                                    \
  "s_bcnt1_i32_b64 %[n], vcc\n\t" /* (also the wait state a VALU read of VCC as an operand needs after a VALU write of it) */            \
  "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"                                                                                                \
  "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"                                                                                             \
  "s_sub_u32 %[pw], %[pw], %[n]\n\t"                                                                                                      \
  "v_add_lshl_u32 %[r], %[r], %[pw], 1\n\t"                                                                                               \
  "v_and_or_b32 %[r], %[r], %[mask], %[ring]\n\t"                                                                                         \
  "v_cndmask_b32 %[r], %[sink], %[r], vcc\n\t"                                                                                            \
  "ds_write_b16 %[r], %[x]\n\t"                                                                                                           \
  "v_cndmask_b32_sdwa %[x], %[x], %[x], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"                       \
  "v_mul_hi_u32 %[q], %[x], " EZ "\n\t"                                                                                                   \
  "v_lshrrev_b32_sdwa %[q], " EW ", %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"                          \
  "v_mul_u32_u24 %[q], %[q], " EW "\n\t"                                                                                                  \
  "v_add3_u32 %[x], %[x], " EY ", %[q]\n\t"
template <uint32_t S>
__device__ __forceinline__ void encode_set_fast(uint32_t &x, const uint4 (&e)[4], uint32_t ring, uint32_t sink, uint32_t &pw)
{
  uint32_t r, q, n;
  unsigned long long saved_exec = 0;
  if constexpr (S == 64)
    asm volatile(HSRANS_ENC_GROUP("%[x3]", "%[y3]", "%[z3]", "%[w3]") HSRANS_ENC_GROUP("%[x2]", "%[y2]", "%[z2]", "%[w2]")
                 HSRANS_ENC_GROUP("%[x1]", "%[y1]", "%[z1]", "%[w1]") HSRANS_ENC_GROUP("%[x0]", "%[y0]", "%[z0]", "%[w0]")
                 : [x] "+v"(x), [pw] "+s"(pw), [r] "=&v"(r), [q] "=&v"(q), [n] "=&s"(n)
                 : [x0] "v"(e[0].x), [y0] "v"(e[0].y), [z0] "v"(e[0].z), [w0] "v"(e[0].w), [x1] "v"(e[1].x), [y1] "v"(e[1].y), [z1] "v"(e[1].z), [w1] "v"(e[1].w),
                   [x2] "v"(e[2].x), [y2] "v"(e[2].y), [z2] "v"(e[2].z), [w2] "v"(e[2].w), [x3] "v"(e[3].x), [y3] "v"(e[3].y), [z3] "v"(e[3].z), [w3] "v"(e[3].w),
                   [ring] "v"(ring), [sink] "v"(sink), [mask] "s"(kOutRing - 1)
                 : "vcc", "scc", "memory");
  else
    asm volatile("s_mov_b64 %[ex], exec\n\ts_and_b64 exec, exec, %[lanes]\n\t" //
                 HSRANS_ENC_GROUP("%[x3]", "%[y3]", "%[z3]", "%[w3]") HSRANS_ENC_GROUP("%[x2]", "%[y2]", "%[z2]", "%[w2]")
                 HSRANS_ENC_GROUP("%[x1]", "%[y1]", "%[z1]", "%[w1]") HSRANS_ENC_GROUP("%[x0]", "%[y0]", "%[z0]", "%[w0]") //
                 "s_mov_b64 exec, %[ex]"
                 : [x] "+v"(x), [pw] "+s"(pw), [r] "=&v"(r), [q] "=&v"(q), [n] "=&s"(n), [ex] "=&s"(saved_exec)
                 : [x0] "v"(e[0].x), [y0] "v"(e[0].y), [z0] "v"(e[0].z), [w0] "v"(e[0].w), [x1] "v"(e[1].x), [y1] "v"(e[1].y), [z1] "v"(e[1].z), [w1] "v"(e[1].w),
                   [x2] "v"(e[2].x), [y2] "v"(e[2].y), [z2] "v"(e[2].z), [w2] "v"(e[2].w), [x3] "v"(e[3].x), [y3] "v"(e[3].y), [z3] "v"(e[3].z), [w3] "v"(e[3].w),
                   [ring] "v"(ring), [sink] "v"(sink), [mask] "s"(kOutRing - 1), [lanes] "s"(0xFFFFFFFFull)
                 : "vcc", "scc", "memory");
  // (an asm statement with several outputs returns a struct; when a member of it lives across basic blocks the instruction selector
  // exports the WHOLE struct in registers of one kind — scalar ones here, and the states cannot be copied into those.  These two empty
  // statements, one output each, give the states and the cursor values of their own inside the block.  No instructions.)
  asm volatile("" : "+v"(x));
  asm volatile("" : "+s"(pw));
}
#undef HSRANS_ENC_GROUP

// One wavefront's job.  RAW = false: block b of an mt_ stream with independent blocks (own histogram, own header).
// RAW = true: a whole raw stream (rANS32x64_16w.cpp:34-166 is ONE dependent chain per coder state, so one wavefront is all the
// format has work for): the counts come from the caller or from k_raw_histogram, checkpoints may sit at listed groups, and the
// image in the slot is the finished stream [n][total][counts][states][words].
template <uint32_t S, bool RAW, uint32_t CHUNK>
__device__ __forceinline__ void encode_body(const EncParams &ep, const uint32_t b, WaveLdsT<CHUNK> &L, const uint32_t lane)
{
  constexpr uint32_t kChunk = CHUNK, kRing = 2 * CHUNK;
  using Chunk = ChunkT<CHUNK>;
  const uint64_t begin = RAW ? 0 : (uint64_t)b * ep.block;
  const uint64_t end = RAW || b + 1 == ep.n_blocks ? ep.n : begin + ep.block;
  const uint32_t size = (uint32_t)(end - begin);
  const uint8_t *in = ep.in;
  uint8_t *slot_end = ep.scratch + (uint64_t)(b + 1) * ep.slot_bytes;

  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t raw[4] = {0, 0, 0, 0};
  if constexpr (RAW)
  {
    bool missing = false;
    for (uint32_t k = 0; k < 4; k++) // byte counts of the whole input (k_raw_histogram); with given_counts they only say which symbols occur
    {
      raw[k] = ep.given_counts ? ep.given_counts[lane * 4 + k] : ep.raw_counts[lane * 4 + k];
      missing |= raw[k] == 0 && ep.raw_counts[lane * 4 + k] != 0;
    }
    if (__builtin_amdgcn_ballot_w64(missing) != 0) // a symbol of the input has no slot in the caller's histogram: not encodable (hsrans_host.cpp put_group returns 0 too)
    {
      if (lane == 0)
      {
        ep.image_bytes[0] = 0;
        ep.result[0] = 0;
        ep.result[1] = 0;
        ep.result[2] = 0;
      }
      return;
    }
  }
  else
  {
  if (ep.raw_counts != nullptr) // counted by k_block_histograms, a workgroup per block, before this kernel
  {
    const uint4 v = ((const uint4 *)(ep.raw_counts + (uint64_t)b * 256))[lane];
    raw[0] = v.x, raw[1] = v.y, raw[2] = v.z, raw[3] = v.w;
  }
  else
  {
  // ---- byte histogram by the coding wavefront itself: kSubHists copies laid out [symbol][copy], copy = lane & 7 ----
  // (a lane's counter sits in bank 8 * (symbol % 4) + copy: lanes of different copies never meet in a bank)
  uint32_t *sub = (uint32_t *)L.table;
  for (uint32_t k = lane; k < kSubHists * 256 / 4; k += 64)
    ((uint4 *)sub)[k] = make_uint4(0, 0, 0, 0);
  wave_sync();
  uint32_t *mine = sub + (lane & (kSubHists - 1));
  auto count16 = [&](const uint4 &d) {
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (uint32_t k = 0; k < 16; k++)
      atomicAdd(&mine[((w[k >> 2] >> (8 * (k & 3))) & 0xFF) * kSubHists], 1u);
  };
  uint32_t off = lane * 16;
  for (; off + 3 * 1024 + 16 <= size; off += 4096) // four loads in flight
  {
    uint4 d[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      d[u] = *(const uint4 *)(in + begin + off + u * 1024);
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      count16(d[u]);
  }
  for (; off < size; off += 1024)
  {
    const uint4 d = load16_guarded(in, begin + off, end);
    const uint32_t have = size - off < 16 ? size - off : 16;
    if (have == 16)
      count16(d);
    else
    {
      const uint32_t w[4] = {d.x, d.y, d.z, d.w};
      for (uint32_t k = 0; k < have; k++)
        atomicAdd(&mine[((w[k >> 2] >> (8 * (k & 3))) & 0xFF) * kSubHists], 1u);
    }
  }
  wave_sync();
  for (uint32_t k = 0; k < 4; k++) // symbol 4 * lane + k: its kSubHists copies are 32 bytes in a row
  {
    const uint4 lo = ((const uint4 *)sub)[(lane * 4 + k) * 2], hi = ((const uint4 *)sub)[(lane * 4 + k) * 2 + 1];
    raw[k] = lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
  }
  wave_sync();
  }
  uint32_t present = 0;
  for (uint32_t k = 0; k < 4; k++)
    present += raw[k] != 0;
  const uint32_t distinct = wave_sum(present);
  if (distinct == 1) // single-symbol block: only the marker word (mt_rANS32x64_16w_encode.cpp:289-295)
  {
    if (present)
    {
      uint32_t sym = lane * 4;
      for (uint32_t k = 0; k < 4; k++)
        if (raw[k])
          sym = lane * 4 + k;
      const uint64_t marker = (uint64_t)size | ((uint64_t)1 << 63) | ((uint64_t)sym << 54);
      ((U64a2 *)(slot_end - 8))->v = marker;
      ep.image_bytes[b] = 8;
      ep.chain_count[b] = 1;
    }
    return;
  }
  } // !RAW

  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  // ---- normalisation (hist.cpp:16-215; hsrans_host.cpp normalize_counts) ----
  const uint32_t target = 1u << ep.bits;
  const float factor = (float)target / (float)(uint64_t)size;
  uint32_t sc[4];
  uint32_t part = 0;
  const bool normalised = RAW && ep.given_counts != nullptr; // the caller's hist_t: already sums to 2^bits (checked on the host)
  for (uint32_t k = 0; k < 4; k++)
  {
    const float v = __fmul_rn((float)raw[k], factor); // one rounding per operation, like the reference's build
    uint32_t c = (uint32_t)(uint16_t)__fadd_rn(v, 0.5f);
    if (c == 0 && raw[k] != 0)
      c = 1;
    if (normalised)
      c = raw[k];
    sc[k] = c;
    part += c;
    L.order[lane * 4 + k] = (c << 8) | (lane * 4 + k);
  }
  const uint32_t sum = wave_sum(part);
  wave_sync();
  // (a block whose scaled counts sum above the target replays ~130 extractions of the heap sort, the others a handful; raising those
  // wavefronts' priority — s_setprio 3 — where two share a SIMD was measured: their normalisation 56 -> 49 us at the 90th percentile,
  // the kernel's length unchanged, 138 us: not kept)
  if (sum != target && !normalised)
  {
    adjust_counts<!RAW>(L, lane, sc, sum, target);
    part = sc[0] + sc[1] + sc[2] + sc[3];
  }
  // exclusive prefix over the 256 counts: lane-local then across lanes
  uint32_t incl = part;
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t o = __shfl_up(incl, d, 64);
    if ((int)lane >= d)
      incl += o;
  }
  uint32_t cum = incl - part;
  for (uint32_t k = 0; k < 4; k++)
  {
    // x' = x + bias + (x / freq) * (2^bits - freq), the division as multiply-high by a rounded-up reciprocal (exact for
    // x < 2^31, which renormalisation guarantees)
    const uint32_t freq = sc[k];
    uint4 e;
    e.x = freq << (31 - ep.bits); // emit when x >= ((2^15 >> bits) << 16) * freq
    if (freq < 2)
    {
      e.y = cum + target - 1; // rcp = 2^32 - 1 gives x - 1 for freq == 1
      e.z = 0xFFFFFFFFu;
      e.w = target - freq;
      if (freq == 0)
        e.x = 0xFFFFFFFFu;
    }
    else
    {
      const uint32_t shift = 32 - __clz(freq - 1); // smallest shift with freq <= 1 << shift
      e.y = cum;
      // ceil(2^(shift + 31) / freq), in [2^31, 2^32): by one correctly rounded double division — the quotient is an integer or at least
      // 2^-15 away from one (freq <= 2^15), the rounding error below 2^-21 — instead of a 64-bit integer division in software
      e.z = (uint32_t)__builtin_ceil(__builtin_ldexp(1.0, (int)shift + 31) / (double)freq);
      e.w = (target - freq) | ((shift - 1) << 24);
    }
    L.table[lane * 4 + k] = e;
    cum += freq;
  }

  const uint64_t t2 = __builtin_amdgcn_s_memrealtime();
  // ---- backward rANS pass over the block (rANS32x64_16w.cpp:34-166) ----
  // The block's bytes pass through an LDS ring of two chunks (byte r of the block at ring offset r mod kRing) with a third
  // chunk on its way in registers; symbols and their table entries are fetched one and two sets (of four groups) ahead of
  // the set being coded, so that the state update is the only dependent chain.
  const uint32_t byte_in_group = enc_lane_to_byte(lane) & (S - 1);
  uint32_t x = 1u << 15;
  uint8_t *slot = slot_end - ep.slot_bytes; // the block's scratch slot; words are written from its end downwards
  uint32_t p = (uint32_t)ep.slot_bytes;     // byte offset (from `slot`) of the lowest word written so far
  // (asking for this first input ahead of the normalisation, which uses neither the ring nor these registers, was measured: nothing)
  const uint32_t n_chunks = (size + kChunk - 1) / kChunk;
  chunk_to_lds(L, chunk_load<CHUNK>(in, begin, end, n_chunks - 1, lane), n_chunks - 1, lane);
  if (n_chunks >= 2)
    chunk_to_lds(L, chunk_load<CHUNK>(in, begin, end, n_chunks - 2, lane), n_chunks - 2, lane);
  Chunk pre{};
  if (n_chunks >= 3)
    pre = chunk_load<CHUNK>(in, begin, end, n_chunks - 3, lane);
  wave_sync();

  // sidecar checkpoints (hsrans_host.cpp encode(): after group gr of the block is coded, gr % interval == 0, gr != 0, the
  // group whole): the decoder's states and read cursor when it is about to start group gr
  const uint32_t whole_groups = size / S;
  auto checkpoint_at = [&](uint64_t ck) {
    if (lane < S)
      ep.ck_states[ck * S + lane] = x;
    if (lane == 0)
      ep.ck_pos[ck] = (uint32_t)ep.slot_bytes - p;
  };
  auto checkpoint = [&](uint32_t gr) { checkpoint_at((uint64_t)b * ep.max_ck + (gr / ep.interval - 1)); };
  if (!RAW && lane == 0)
    ep.chain_count[b] = 1 + (ep.interval != 0 && whole_groups >= 1 ? (whole_groups - 1) / ep.interval : 0);
  // RAW with listed checkpoints (ascending group indices, multiples of 4; hsrans_host.cpp encode(): `wanted`): the list is walked
  // from its end; entries at or behind the last whole group are never checkpoints
  uint32_t ck_left = RAW && ep.ck_groups ? ep.n_ck_groups : 0;
  while (ck_left != 0 && ep.ck_groups[ck_left - 1] >= whole_groups)
    ck_left--;
  uint32_t ck_want = ck_left ? ep.ck_groups[ck_left - 1] : 0; // (0 is never asked for)
  auto listed = [&](uint32_t gr) {
    if (!RAW || gr != ck_want || gr == 0)
      return;
    checkpoint_at(ck_left - 1);
    ck_left--;
    ck_want = ck_left ? ep.ck_groups[ck_left - 1] : 0;
  };

  // emitted words: [p, flushed_to) of the slot is still in the LDS ring only; a whole segment leaves with one 8-byte store per lane
  const uint32_t out_ring = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)L.order;
  const uint32_t sink = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)L.sink + 4 * lane;
  uint32_t flushed_to = (uint32_t)ep.slot_bytes; // (a multiple of kOutSeg: encode_slot_bytes)
  auto flush_segment = [&]() {
    flushed_to -= kOutSeg;
    const uint2 v = *(const uint2 *)((const uint8_t *)L.order + ((flushed_to & (kOutRing - 1)) + lane * 8));
    *(uint2 *)(slot + flushed_to + lane * 8) = v;
  };

  uint32_t g = (size + S - 1) / S; // groups of the block still to code; group i covers bytes [i*S, i*S+S)
  if (size % S != 0)               // only the file's last group can be partial
  {
    g--;
    encode_group_slow<S, CHUNK>(x, L, g * S, size - g * S, p, lane, byte_in_group);
  }
  while (g % 4 != 0)
  {
    g--;
    encode_group_slow<S, CHUNK>(x, L, g * S, S, p, lane, byte_in_group);
  }
  wave_sync();
  if (p + kOutSeg <= flushed_to) // (at most four groups so far: at most one segment)
    flush_segment();

  if (ep.interval != 0 && g != 0 && g % ep.interval == 0 && g < whole_groups)
    checkpoint(g);
  if (g < whole_groups)
    listed(g);

  constexpr uint32_t kSetBytes = 4 * S;
  constexpr uint32_t kSetsPerChunk = kChunk / kSetBytes;
  const uint8_t *my_byte = L.stage + byte_in_group;
  auto read_syms = [&](int32_t t, uint32_t(&s)[4]) { // t < 0 wraps inside the ring: harmless, never coded
    const uint32_t off = (uint32_t)t * kSetBytes;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      s[k] = my_byte[(off + k * S) & (kRing - 1)];
  };
  auto read_entries = [&](const uint32_t(&s)[4], uint4(&e)[4]) {
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      e[k] = L.table[s[k]];
  };
  auto chunk_finished = [&](uint32_t c) { // every group of chunk c is coded: its ring half takes the chunk two further down
    if (c < 2)
      return;
    chunk_to_lds(L, pre, c - 2, lane);
    if (c >= 3)
      pre = chunk_load<CHUNK>(in, begin, end, c - 3, lane);
    wave_sync();
  };
  if ((uint64_t)g * S <= (uint64_t)(n_chunks - 1) * kChunk) // the groups coded one by one above were all of the last chunk
    chunk_finished(n_chunks - 1);

  // What else happens after a set — a checkpoint, a chunk of input used up — happens at sets known in advance: `event` is the next
  // such set below, and the loop pays one compare per set for all of it (the wavefront pays for every instruction, see above)
  const uint32_t ck_sets = ep.interval != 0 ? ep.interval / (ep.interval % 4 == 0 ? 4 : ep.interval % 2 == 0 ? 2 : 1) : 0; // 4 t % interval == 0  <=>  t % ck_sets == 0
  uint32_t pw = __builtin_amdgcn_readfirstlane(p >> 1); // the cursor in words
  int32_t t = (int32_t)(g / 4) - 1;
  // the next set (<= t) that ends a chunk whose ring half is wanted / that a checkpoint follows / that a listed checkpoint follows; -1: none
  int32_t ev_chunk = t >= (int32_t)(2 * kSetsPerChunk) ? (int32_t)((uint32_t)t / kSetsPerChunk * kSetsPerChunk) : -1;
  int32_t ev_ck = ck_sets != 0 && t >= (int32_t)ck_sets ? (int32_t)((uint32_t)t / ck_sets * ck_sets) : -1;
  auto listed_set = [&]() { return RAW && ck_want != 0 && (int32_t)(ck_want / 4) <= t ? (int32_t)(ck_want / 4) : -1; };
  int32_t ev_listed = listed_set();
  auto max3 = [](int32_t a, int32_t b, int32_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); };
  int32_t event = max3(ev_chunk, ev_ck, ev_listed);
  auto set_events = [&]() { // set t (groups 4t .. 4t+3) is coded and something is due after it
    p = pw << 1;
    if (t == ev_ck)
    {
      checkpoint((uint32_t)(4 * t));
      ev_ck = t > (int32_t)ck_sets ? t - (int32_t)ck_sets : -1; // (never after set 0: group 0 starts no chain of its own)
    }
    if (RAW && t == ev_listed)
    {
      listed((uint32_t)(4 * t));
      ev_listed = listed_set();
      if (ev_listed == t)
        ev_listed = -1;
    }
    if (t == ev_chunk)
    {
      chunk_finished((uint32_t)t / kSetsPerChunk);
      ev_chunk = t >= (int32_t)(3 * kSetsPerChunk) ? t - (int32_t)kSetsPerChunk : -1;
    }
    event = max3(ev_chunk, ev_ck, ev_listed);
  };

  uint32_t flush_at = (flushed_to - kOutSeg) >> 1; // a segment is complete when the cursor is at or below this word
  uint32_t sa[4], sb[4];
  uint4 ea[4], eb[4];
  read_syms(t, sa);
  read_entries(sa, ea);
  read_syms(t - 1, sb);
  // one wait per set: everything fetched during the last set (this set's entries, the next set's symbols) has had a whole set to arrive;
  // said here, the compiler drops the five finer waits it would spread over the set (each a slot of the wavefront's issue port)
  auto lds_arrived = [] {
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);  // (nothing moves above it: a use scheduled ahead of it would get a wait of its own)
  };
  auto after_set = [&]() { // (both unlikely: a taken branch costs the lone wavefront a refill of its instruction buffer, the straight path must be the common one)
    if (__builtin_expect(pw <= flush_at, 0))
    {
      flush_segment();
      flush_at -= kOutSeg >> 1;
    }
    if (__builtin_expect(t == event, 0))
      set_events();
  };
  if (t >= 0 && (t & 1) == 0) // an odd number of sets: one set ahead of the loop, which then runs two sets a turn and never leaves in the middle
  {
    lds_arrived();
    read_entries(sb, eb);
    read_syms(t - 2, sa);
    encode_set_fast<S>(x, ea, out_ring, sink, pw);
    after_set();
    --t;
    for (uint32_t k = 0; k < 4; k++)
      ea[k] = eb[k], sb[k] = sa[k];
  }
  while (t >= 0)
  {
    lds_arrived();
    read_entries(sb, eb);
    read_syms(t - 2, sa);
    encode_set_fast<S>(x, ea, out_ring, sink, pw);
    after_set();
    --t;
    lds_arrived();
    read_entries(sa, ea);
    read_syms(t - 2, sb);
    encode_set_fast<S>(x, eb, out_ring, sink, pw);
    after_set();
    --t;
  }
  p = pw << 1;

  const uint64_t t3 = __builtin_amdgcn_s_memrealtime();
  if (ep.stamps && lane == 0)
  {
    ep.stamps[b * 4 + 0] = t0;
    ep.stamps[b * 4 + 1] = t1;
    ep.stamps[b * 4 + 2] = t2;
    ep.stamps[b * 4 + 3] = t3;
  }
  const uint32_t words_bytes = (uint32_t)ep.slot_bytes - p;
  wave_sync();
  for (uint32_t o = p + 2 * lane; o < flushed_to; o += 128) // the ring's rest (less than two segments), word by word
    *(uint16_t *)(slot + o) = *(const uint16_t *)((const uint8_t *)L.order + (o & (kOutRing - 1)));
  if constexpr (RAW)
  {
    // ---- the stream's header in front of the words: [n u64][total u64][counts 256 x u16][states S x u32] (rANS32x64_16w.cpp:150-166) ----
    constexpr uint32_t kRawHeader = 16 + 512 + 4 * S;
    uint8_t *h = slot + p - kRawHeader;
    const uint64_t total = (uint64_t)kRawHeader + words_bytes;
    if (lane == 0)
    {
      ((U64a2 *)h)->v = ep.n;
      ((U64a2 *)(h + 8))->v = total;
      ep.image_bytes[0] = total;
      ep.image_off[0] = 0;
      ep.result[0] = total;
      ep.result[1] = total <= ep.out_cap ? 1 : 0;
      ep.result[2] = ck_left; // (listed checkpoints that were not met: 0 unless the list was not what the host validated)
    }
    for (uint32_t k = 0; k < 4; k++)
      *(uint16_t *)(h + 16 + 2 * (lane * 4 + k)) = (uint16_t)sc[k];
    if (lane < S)
      ((U32a2 *)(h + 16 + 512 + 4 * lane))->v = x;
    return;
  }
  // ---- block header in front of the words: [size u64][skip u64][states S x u32][counts 256 x u16] ----
  constexpr uint32_t kHeader = 16 + 4 * S + 512;
  uint8_t *h = slot + p - kHeader;
  // skip: uint16 units from the state array to the next block header, minus one; the last block's is one less
  // (hsrans_host.cpp encode(); mt_rANS32x64_16w_encode.cpp:149,277)
  const uint64_t skip = (uint64_t)(4 * S + 512 + words_bytes) / 2 - 1 - (b + 1 == ep.n_blocks ? 1 : 0);
  if (lane == 0)
  {
    ((U64a2 *)h)->v = (uint64_t)size;
    ((U64a2 *)(h + 8))->v = skip;
    ep.image_bytes[b] = (uint64_t)(slot_end - h);
  }
  if (lane < S)
    ((U32a2 *)(h + 16 + 4 * lane))->v = x;
  for (uint32_t k = 0; k < 4; k++)
    *(uint16_t *)(h + 16 + 4 * S + 2 * (lane * 4 + k)) = (uint16_t)sc[k];
}

template <uint32_t S, uint32_t CHUNK>
__global__ void __launch_bounds__(64 * kWavesPerWG) k_encode_blocks(EncParams ep)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const uint32_t lane = lane_id();
  const uint32_t wave = kWavesPerWG == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t b = blockIdx.x * kWavesPerWG + wave;
  if (b >= ep.n_blocks)
    return;
  using WaveLds = WaveLdsT<CHUNK>;
  encode_body<S, false, CHUNK>(ep, b, *(WaveLds *)(lds_raw + (size_t)wave * sizeof(WaveLds)), lane);
}

// ---- raw streams: K_hist (wide) -> K_raw (one wavefront) -> K_copy (wide) -----------------------------------------------
// byte counts of the whole input into counts[256] (zeroed by the launcher): per-workgroup LDS histograms (the sub-histogram layout
// of the block encoder), one global atomic per symbol and workgroup
__global__ void __launch_bounds__(256) k_raw_histogram(const uint8_t *in, uint64_t n, uint32_t *counts)
{
  constexpr uint32_t kSubStride = 257; // dwords between copies: copy c of symbol s sits in bank (c + s) % 32, not all in bank s % 32
  __shared__ uint32_t sub[kSubHists * kSubStride];
  for (uint32_t k = threadIdx.x; k < kSubHists * kSubStride; k += 256)
    sub[k] = 0;
  __syncthreads();
  uint32_t *mine = sub + (threadIdx.x & (kSubHists - 1)) * kSubStride;
  for (uint64_t off = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; off < n; off += (uint64_t)gridDim.x * 4096)
  {
    const uint4 d = load16_guarded(in, off, n);
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    const uint32_t have = n - off < 16 ? (uint32_t)(n - off) : 16;
    for (uint32_t k = 0; k < have; k++)
      atomicAdd(&mine[(w[k >> 2] >> (8 * (k & 3))) & 0xFF], 1u);
  }
  __syncthreads();
  uint32_t v = 0;
  for (uint32_t c = 0; c < kSubHists; c++)
    v += sub[c * kSubStride + threadIdx.x];
  if (v)
    atomicAdd(&counts[threadIdx.x], v);
}

// byte counts of every block of an mt_ encode, a workgroup per block: counts[b][256].  The coding wavefront used to count its own
// block (33 us of its ~270 at 64 KiB, 130 us at 256 KiB: 64 lanes of LDS atomics); 256 threads per block and every CU at it take
// a few microseconds for the whole input, and the block is in L2 when the coding wavefront reads it again.
__global__ void __launch_bounds__(256) k_block_histograms(EncParams ep, uint32_t *counts)
{
  // 32 copies of the histogram, copy = lane & 31, laid out [symbol][copy]: a lane's counter is ALWAYS in bank (lane & 31), whatever the
  // symbol, so the 32 lanes the LDS serves at a time never meet in a bank (the [copy][symbol] layout of round 4, copies 257 dwords apart,
  // put a counter in bank (copy + symbol) % 32: 64 random symbols per instruction, about five to a bank — 31.7 us per 100 MB); equal
  // bytes only meet from lanes 32 apart, which the LDS serves one after the other anyway
  // (23.7 us per 100 MB.  Fewer copies to fit more workgroups on a CU — 24: 6 per CU, all 1,526 blocks in one round — measured slower:
  // 27.3 us with 24, 26.5 us with 16: it is the atomics, not the tail of the grid)
  constexpr uint32_t kCopies = 32;
  __shared__ uint32_t sub[256 * kCopies];
  const uint32_t b = blockIdx.x;
  const uint64_t begin = (uint64_t)b * ep.block;
  const uint64_t end = b + 1 == ep.n_blocks ? ep.n : begin + ep.block;
  for (uint32_t k = threadIdx.x; k < 256 * kCopies; k += 256)
    sub[k] = 0;
  __syncthreads();
  uint32_t *mine = sub + (threadIdx.x & (kCopies - 1));
  auto count16 = [&](const uint4 &d) {
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (uint32_t k = 0; k < 16; k++)
      atomicAdd(&mine[((w[k >> 2] >> (8 * (k & 3))) & 0xFF) * kCopies], 1u);
  };
  uint64_t off = begin + threadIdx.x * 16;
  for (; off + 3 * 4096 + 16 <= end; off += 4 * 4096) // four loads in flight per thread
  {
    uint4 d[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      d[u] = *(const uint4 *)(ep.in + off + u * 4096);
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      count16(d[u]);
  }
  for (; off < end; off += 4096)
  {
    const uint4 d = load16_guarded(ep.in, off, end);
    const uint32_t have = end - off < 16 ? (uint32_t)(end - off) : 16;
    if (have == 16)
      count16(d);
    else
    {
      const uint32_t w[4] = {d.x, d.y, d.z, d.w};
      for (uint32_t k = 0; k < have; k++)
        atomicAdd(&mine[((w[k >> 2] >> (8 * (k & 3))) & 0xFF) * kCopies], 1u);
    }
  }
  __syncthreads();
  uint32_t v = 0;
  for (uint32_t c = 0; c < kCopies; c++) // (thread t sums symbol t's copies starting at copy t: the 32 lanes served together read 32 banks)
    v += sub[threadIdx.x * kCopies + ((threadIdx.x + c) & (kCopies - 1))];
  counts[(uint64_t)b * 256 + threadIdx.x] = v;
}

template <uint32_t S>
__global__ void __launch_bounds__(64) k_encode_raw(EncParams ep)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  encode_body<S, true, kChunkFew>(ep, 0, *(WaveLdsT<kChunkFew> *)lds_raw, lane_id());
}

// (Round 4 also built single-pass placement — the coding wavefront learns its image's position by a decoupled look-back over status
// words (memory-side atomics: the XCDs' L2s are not coherent) and copies the image itself, no scan and gather kernels — and took it
// out again: the blocks of one launch all finish at about the same time, so nearly every look-back walks the whole chain of own
// sizes, 64 per round trip: 0.302 ms against 0.283 for 100 MB in 64 KiB blocks, 0.348 against 0.277 in 32 KiB blocks.)
// ---- K_scan: one workgroup ----------------------------------------------------------------------------------------
// exclusive prefix of `v` over the 1024 threads of the workgroup, plus the workgroup total in *total
__device__ __forceinline__ uint64_t wg_exclusive_scan(uint64_t v, uint64_t *wave_tot, uint64_t *total)
{
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t incl = v;
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint64_t o = __shfl_up(incl, d, 64);
    if ((int)lane >= d)
      incl += o;
  }
  __syncthreads(); // wave_tot may still be read by the previous call
  if (lane == 63)
    wave_tot[wave] = incl;
  __syncthreads();
  uint64_t before = 0, all = 0;
  for (uint32_t w = 0; w < 16; w++)
  {
    before += w < wave ? wave_tot[w] : 0;
    all += wave_tot[w];
  }
  *total = all;
  return before + incl - v;
}

__global__ void __launch_bounds__(1024) k_scan_images(EncParams ep)
{
  __shared__ uint64_t wave_tot[16];
  uint64_t bytes_before = 16; // file header
  uint64_t chains_before = 0, coded_blocks = 0, last_hist = 0;
  constexpr uint64_t kNone = ~(uint64_t)0;
  uint64_t my_last = kNone; // highest non-single block this thread has seen
  for (uint32_t base = 0; base < ep.n_blocks; base += 1024)
  {
    const uint32_t i = base + threadIdx.x;
    const bool have = i < ep.n_blocks;
    const uint64_t bytes = have ? ep.image_bytes[i] : 0;
    const uint64_t chains = have ? ep.chain_count[i] : 0;
    uint64_t t_bytes, t_chains, t_coded;
    const uint64_t off = wg_exclusive_scan(bytes, wave_tot, &t_bytes);
    const uint64_t coff = wg_exclusive_scan(chains, wave_tot, &t_chains);
    (void)wg_exclusive_scan(have && bytes != 8 ? 1 : 0, wave_tot, &t_coded);
    if (have)
    {
      ep.image_off[i] = bytes_before + off;
      ep.chain_off[i] = (uint32_t)(chains_before + coff);
      if (bytes != 8)
        my_last = bytes_before + off + 16 + 4 * (uint64_t)ep.S; // where this block's counts will be
    }
    bytes_before += t_bytes;
    chains_before += t_chains;
    coded_blocks += t_coded;
  }
  // any thread's candidate will do when there is exactly one coded block (the only case the value is used in)
  __shared__ uint64_t hist_s;
  if (threadIdx.x == 0)
    hist_s = 0;
  __syncthreads();
  if (my_last != kNone)
    atomicMax((unsigned long long *)&hist_s, (unsigned long long)my_last);
  __syncthreads();
  last_hist = hist_s;
  if (threadIdx.x == 0)
  {
    const uint64_t total = bytes_before;
    ep.result[0] = total;
    ep.result[1] = total <= ep.out_cap ? 1 : 0;
    if (ep.fits)
      *ep.fits = total <= ep.out_cap ? 1 : 0;
    ep.result[2] = chains_before;
    ep.result[3] = coded_blocks;
    ep.result[4] = last_hist;
    if (total <= ep.out_cap)
    {
      ((uint64_t *)ep.out)[0] = ep.n;
      ((uint64_t *)ep.out)[1] = total;
    }
  }
}

// ---- K_plan: the sidecar plan of the stream (hsrans_plan.h), one wavefront per block --------------------------------
// Writes exactly what the host encoder emits for the same layout (hsrans_host.cpp encode(), "sidecar plan"): single-piece
// chains in output order — per coded block one chain from the block header's states plus one per checkpoint, per
// single-symbol block one fill chain — and the Group records of the grouped decode launch (one group per block).
__global__ void __launch_bounds__(64) k_plan_blocks(EncParams ep)
{
  const uint32_t b = blockIdx.x, lane = threadIdx.x, S = ep.S;
  const uint32_t nc = ep.n_chains;
  uint32_t *chain_first = (uint32_t *)(ep.plan + plan_chain_first_off());
  Piece *pieces = (Piece *)(ep.plan + plan_pieces_off(nc));
  uint32_t *states = (uint32_t *)(ep.plan + plan_states_off(nc, nc));
  const uint64_t begin = (uint64_t)b * ep.block;
  const uint64_t end = b + 1 == ep.n_blocks ? ep.n : begin + ep.block;
  const uint64_t bytes = ep.image_bytes[b], at = ep.image_off[b];
  const uint32_t c0 = ep.chain_off[b], count = ep.chain_count[b];
  const bool single = bytes == 8;
  const uint64_t header = 16 + 4 * (uint64_t)S + 512;
  const uint64_t hist_off = at + 16 + 4 * (uint64_t)S;
  const uint64_t whole_file = ep.n / S; // whole groups of the file (rANS32x64_16w.cpp:220)
  const uint8_t *image = ep.scratch + (uint64_t)(b + 1) * ep.slot_bytes - bytes;
  if (b == 0 && lane == 0)
    chain_first[nc] = nc;
  for (uint32_t k = lane; k < count; k += 64)
  {
    Piece p{};
    p.flags = kPieceChainStart;
    p.state_idx = c0 + k;
    if (single)
    {
      p.out_off = begin;
      p.hist_off = (image[6] >> 6) | ((uint64_t)(image[7] & 0x3F) << 2); // bits 54..61 of the marker
      p.fill_len = end - begin;
      p.flags |= kPieceFill;
    }
    else
    {
      const uint64_t g_first = begin / S, g_end = (end - 1) / S + 1;
      const uint64_t g0 = g_first + (uint64_t)k * ep.interval;
      const uint64_t g1 = k + 1 < count ? g0 + ep.interval : g_end;
      p.words_off = k == 0 ? at + header : at + bytes - ep.ck_pos[(uint64_t)b * ep.max_ck + (k - 1)];
      p.out_off = g0 * S;
      p.hist_off = hist_off;
      const uint64_t stop = g1 < whole_file ? g1 : whole_file;
      p.steps = (uint32_t)(stop > g0 ? stop - g0 : 0);
      p.tail = (uint16_t)(k + 1 == count && end == ep.n ? ep.n - whole_file * S : 0);
    }
    pieces[c0 + k] = p;
    chain_first[c0 + k] = c0 + k;
  }
  // start states: chain 0 from the block header (2-byte aligned in the image), the others from the checkpoint slots
  for (uint32_t k = 0; k < count; k++)
    if (lane < S)
    {
      uint32_t v = 0;
      if (!single)
        v = k == 0 ? ((const U32a2 *)(image + 16 + 4 * lane))->v : ep.ck_states[((uint64_t)b * ep.max_ck + (k - 1)) * S + lane];
      states[(uint64_t)(c0 + k) * S + lane] = v;
    }
  if (ep.groups != nullptr && lane < ep.group_split)
  {
    // one Group per part: a block is cut into up to group_split parts of >= 128 chains; the parts that are not needed
    // (short last block, single-symbol block) stay empty
    const uint32_t by_size = group_parts_of(count, ep.group_split); // (hsrans_kernels.h)
    const uint32_t k = single || by_size < 1 ? 1 : by_size;
    Group g{};
    g.flags = single ? kGroupFill : kGroupMergeable;
    g.hist_off = single ? 0 : hist_off;
    if (lane < k)
    {
      const uint32_t lo = (uint32_t)((uint64_t)count * lane / k), hi = (uint32_t)((uint64_t)count * (lane + 1) / k);
      g.begin = c0 + lo;
      g.piece0 = c0 + lo; // (k_plan_blocks: chain c is piece c, state set c)
      g.count = hi - lo;
      g.words_end = hi < count ? at + bytes - ep.ck_pos[(uint64_t)b * ep.max_ck + (hi - 1)] : at + bytes;
    }
    else
    {
      g.begin = c0;
      g.piece0 = c0;
      g.count = 0;
      g.flags = kGroupFill;
      g.words_end = at + bytes;
    }
    ((Group *)ep.groups)[(uint64_t)b * ep.group_split + lane] = g;
  }
}

// ---- K_gather: one workgroup per block image; source and destination are only 2-byte aligned -------------------------
struct __attribute__((packed, aligned(2))) U128a2
{
  uint32_t v[4];
};

// sum of `v` over the 256 threads of the workgroup (every thread gets it); `slot`: 4 words of LDS of the caller's
__device__ __forceinline__ uint64_t wg_sum(uint64_t v, uint64_t *slot)
{
  for (int d = 32; d >= 1; d >>= 1)
    v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0)
    slot[threadIdx.x >> 6] = v;
  __syncthreads();
  return slot[0] + slot[1] + slot[2] + slot[3];
}

// Up to this many blocks, every workgroup of K_gather adds up the image sizes in front of its own block itself (n / 256 loads per
// thread out of L2, two reductions) and the scan kernel with its launch is not run at all: K_scan is one workgroup's worth of latency
// (13.5 us at 1,526 blocks) on the path of every encode.  Beyond it the reads grow with the square of the block count: K_scan again.
constexpr uint32_t kSelfScanBlocks = 4096;

__global__ void __launch_bounds__(256) k_gather_images(EncParams ep)
{
  const uint32_t b = blockIdx.x;
  const uint64_t bytes = ep.image_bytes[b];
  uint64_t off;
  if (ep.n_blocks <= kSelfScanBlocks)
  {
    __shared__ uint64_t red[5][4];
    uint64_t before = 0, total = 0, chains_before = 0, chains = 0, coded = 0;
    for (uint32_t i = threadIdx.x; i < ep.n_blocks; i += 256)
    {
      const uint64_t by = ep.image_bytes[i];
      const uint64_t ch = ep.chain_count[i];
      total += by;
      chains += ch;
      coded += by != 8 ? 1 : 0;
      before += i < b ? by : 0;
      chains_before += i < b ? ch : 0;
    }
    before = wg_sum(before, red[0]);
    total = 16 + wg_sum(total, red[1]); // (16: the file header)
    chains_before = wg_sum(chains_before, red[2]);
    chains = wg_sum(chains, red[3]);
    coded = wg_sum(coded, red[4]);
    off = 16 + before;
    const bool fits = total <= ep.out_cap;
    if (threadIdx.x == 0)
    {
      ep.image_off[b] = off;
      ep.chain_off[b] = (uint32_t)chains_before;
      if (bytes != 8 && coded == 1) // the one coded block of the stream says where its counts are (read in that case only)
        ep.result[4] = off + 16 + 4 * (uint64_t)ep.S;
      if (b + 1 == ep.n_blocks)
      {
        ep.result[0] = total;
        ep.result[1] = fits ? 1 : 0;
        ep.result[2] = chains;
        ep.result[3] = coded;
        if (fits)
        {
          ((uint64_t *)ep.out)[0] = ep.n;
          ((uint64_t *)ep.out)[1] = total;
        }
      }
    }
    if (!fits)
      return;
  }
  else
  {
    if (*ep.fits == 0)
      return;
    off = ep.image_off[b];
  }
  const uint8_t *src = ep.scratch + (uint64_t)(b + 1) * ep.slot_bytes - bytes;
  uint8_t *dst = ep.out + off;
  // head: up to the first 16-byte boundary of dst
  uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
  if (head > bytes)
    head = bytes;
  for (uint64_t i = threadIdx.x * 2; i < head; i += 512)
    *(uint16_t *)(dst + i) = *(const uint16_t *)(src + i);
  const uint64_t body = (bytes - head) / 16;
  uint64_t i = threadIdx.x;
  for (; i + 3 * 256 < body; i += 4 * 256) // four loads in flight per thread
  {
    U128a2 v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      v[u] = *(const U128a2 *)(src + head + (i + u * 256) * 16);
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
      *(uint4 *)(dst + head + (i + u * 256) * 16) = make_uint4(v[u].v[0], v[u].v[1], v[u].v[2], v[u].v[3]);
  }
  for (; i < body; i += 256)
  {
    const U128a2 v = *(const U128a2 *)(src + head + i * 16);
    *(uint4 *)(dst + head + i * 16) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
  }
  for (uint64_t k = head + body * 16 + threadIdx.x * 2; k < bytes; k += 512)
    *(uint16_t *)(dst + k) = *(const uint16_t *)(src + k);
}

// the one image of a raw encode (the whole stream), copied by the whole grid
__global__ void __launch_bounds__(256) k_copy_image(EncParams ep)
{
  if (ep.result[1] == 0)
    return;
  const uint64_t bytes = ep.image_bytes[0];
  const uint8_t *src = ep.scratch + ep.slot_bytes - bytes; // 2-byte aligned; ep.out is 16-byte aligned
  const uint64_t body = bytes / 16;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < body; i += (uint64_t)gridDim.x * 256)
  {
    const U128a2 v = *(const U128a2 *)(src + i * 16);
    *(uint4 *)(ep.out + i * 16) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
  }
  if (blockIdx.x == 0)
    for (uint64_t i = body * 16 + threadIdx.x * 2; i < bytes; i += 512)
      *(uint16_t *)(ep.out + i) = *(const uint16_t *)(src + i);
}

} // namespace

uint32_t encode_block_count(uint64_t n, uint64_t block, uint32_t S)
{
  // hsrans_host.cpp split_blocks: the last block absorbs a remainder shorter than one group
  uint64_t count = (n + block - 1) / block;
  if (count > 1 && n - (count - 1) * block < S)
    count--;
  return count > 0xFFFFFFFFull ? 0 : (uint32_t)count;
}

uint64_t encode_slot_bytes(uint64_t block, uint32_t S)
{
  // every symbol emits at most one 16-bit word; the last block can be up to S-1 symbols longer
  const uint64_t need = 2 * (block + S) + 16 + 4 * (uint64_t)S + 512;
  return (need + 511) / 512 * 512; // (a multiple of the encoder's flush segment)
}

hipError_t launch_encode(const EncParams &ep, hipStream_t stream, bool *prepared_flag)
{
  bool local = false;
  bool &prepared = prepared_flag ? *prepared_flag : local;
  const size_t lds_few = sizeof(WaveLdsT<kChunkFew>) * kWavesPerWG, lds_many = sizeof(WaveLdsT<kChunkMany>) * kWavesPerWG;
  if (!prepared)
  {
    hipError_t e = hipFuncSetAttribute((const void *)k_encode_blocks<64, kChunkFew>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_few);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_encode_blocks<32, kChunkFew>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_few);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_encode_blocks<64, kChunkMany>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_many);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_encode_blocks<32, kChunkMany>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_many);
    if (e != hipSuccess)
      return e;
    prepared = true;
  }
  // few blocks: all resident at once with the large chunks too (11 workgroups of 13.25 KiB per CU: the LDS is handed out in pieces)
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 1;
  const bool few = ep.n_blocks <= 11u * (uint32_t)cus * kWavesPerWG;
  const size_t lds = few ? lds_few : lds_many;
  const uint32_t grid = (ep.n_blocks + kWavesPerWG - 1) / kWavesPerWG;
  (void)hipGetLastError(); // (sticky per thread)
  if (ep.raw_counts != nullptr)
    hipLaunchKernelGGL(k_block_histograms, dim3(ep.n_blocks), dim3(256), 0, stream, ep, const_cast<uint32_t *>(ep.raw_counts));
  if (ep.S == 64 && few)
    hipLaunchKernelGGL((k_encode_blocks<64, kChunkFew>), dim3(grid), dim3(64 * kWavesPerWG), lds, stream, ep);
  else if (ep.S == 64)
    hipLaunchKernelGGL((k_encode_blocks<64, kChunkMany>), dim3(grid), dim3(64 * kWavesPerWG), lds, stream, ep);
  else if (few)
    hipLaunchKernelGGL((k_encode_blocks<32, kChunkFew>), dim3(grid), dim3(64 * kWavesPerWG), lds, stream, ep);
  else
    hipLaunchKernelGGL((k_encode_blocks<32, kChunkMany>), dim3(grid), dim3(64 * kWavesPerWG), lds, stream, ep);
  if (ep.n_blocks > kSelfScanBlocks)
    hipLaunchKernelGGL(k_scan_images, dim3(1), dim3(1024), 0, stream, ep);
  hipLaunchKernelGGL(k_gather_images, dim3(ep.n_blocks), dim3(256), 0, stream, ep);
  return hipGetLastError();
}

hipError_t launch_encode_raw(const EncParams &ep, uint32_t *d_counts, hipStream_t stream, bool *prepared_flag)
{
  bool local = false;
  bool &prepared = prepared_flag ? *prepared_flag : local;
  const size_t lds = sizeof(WaveLdsT<kChunkFew>);
  if (!prepared)
  {
    hipError_t e = hipFuncSetAttribute((const void *)k_encode_raw<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)k_encode_raw<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
      return e;
    prepared = true;
  }
  (void)hipGetLastError();
  {
    // (also with the caller's histogram: the coding wavefront checks that every symbol that occurs has a slot in it)
    hipError_t e = hipMemsetAsync(d_counts, 0, 256 * 4, stream);
    if (e != hipSuccess)
      return e;
    const uint64_t per_wg = 1 << 16; // 16 passes of a workgroup's 4 KiB
    const uint32_t grid = (uint32_t)std::min<uint64_t>((ep.n + per_wg - 1) / per_wg, 4096);
    hipLaunchKernelGGL(k_raw_histogram, dim3(grid ? grid : 1), dim3(256), 0, stream, ep.in, ep.n, d_counts);
  }
  if (ep.S == 64)
    hipLaunchKernelGGL(k_encode_raw<64>, dim3(1), dim3(64), lds, stream, ep);
  else
    hipLaunchKernelGGL(k_encode_raw<32>, dim3(1), dim3(64), lds, stream, ep);
  hipLaunchKernelGGL(k_copy_image, dim3(1024), dim3(256), 0, stream, ep);
  return hipGetLastError();
}

hipError_t launch_encode_plan(const EncParams &ep, hipStream_t stream)
{
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_plan_blocks, dim3(ep.n_blocks), dim3(64), 0, stream, ep);
  return hipGetLastError();
}

} // namespace hsrans
