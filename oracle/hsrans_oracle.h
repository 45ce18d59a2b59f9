/* TEST INFRASTRUCTURE ONLY — the CPU oracle for the hypersonic-rANS 32-bit-state / 16-bit-word decode path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * (hypersonic_rans_amd/) never includes, links or calls anything in oracle/.
 *
 * Every function is a plain-C restatement of one reference function; the citation (file:line under
 * /root/reference/src) is given next to each.  Parity of this restatement is PINNED: tests/test_oracle_vs_ref.py
 * compares it against the real reference compiled into oracle/_ref/libhsrans_ref.so (when present) and
 * tests/test_oracle_golden.py against the committed golden vectors in tests/golden/ that were generated
 * from that same real reference by tests/golden/make_golden.py.
 */
#ifndef HSRANS_ORACLE_H
#define HSRANS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hist.h:16-20 */
typedef struct
{
  uint16_t symbolCount[256];
  uint16_t cumul[256];
} orc_hist_t;

enum { ORC_RAW = 0, ORC_BLOCK = 1, ORC_MT = 2 };

/* lane -> byte-in-group permutation; rANS32x64_16w.cpp:210-216 (32-state table = first 32 entries, rANS32x32_16w.cpp:203) */
uint8_t orc_idx2idx(unsigned j);

/* hist.cpp:8-14 */
void orc_observe_hist(uint32_t hist[256], const uint8_t *data, size_t size);
/* hist.cpp:16-215 (FloatingPointHistLimit = true, NewHistModeling = false branch) */
void orc_normalize_hist(orc_hist_t *out, const uint32_t hist[256], size_t dataBytes, unsigned bits);
/* hist.cpp:217-222 */
void orc_make_hist(orc_hist_t *out, const uint8_t *data, size_t size, unsigned bits);
/* hist.cpp:326-354 (inplace_make_hist_dec): fills cumul[256] and cumulInv[1<<bits]; returns 0 when the counts do
 * not sum to 1<<bits.  The reference sums in uint16_t (hist.cpp:332); `wide_sum` != 0 selects the uint32_t sum of
 * inplace_complete_hist (hist.cpp:308-324), which is what the SIMD / block_ / mt_ dispatch paths use. */
int orc_make_dec_table(unsigned bits, const uint16_t counts[256], uint16_t cumul[256], uint8_t *cumulInv, int wide_sum);

/* rANS32x64_16w.cpp:10-13, rANS32x32_16w.cpp:10-13, block_rANS32x64_16w_encode.cpp:47-54, mt_rANS32x64_16w_encode.cpp:50-57 */
size_t orc_capacity(int container, int states, size_t n);

/* rANS32x64_16w.cpp:34-166 / rANS32x32_16w.cpp:34-159 */
size_t orc_raw_encode(int states, unsigned bits, const uint8_t *in, size_t n, uint8_t *out, size_t cap, const orc_hist_t *hist);
/* rANS32x64_16w.cpp:168-283 / rANS32x32_16w.cpp:161-269  — THE bit-exactness oracle */
size_t orc_raw_decode(int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap);
/* block_rANS32x64_16w_decode.cpp:12-126 (scalar section coder block_codec64.h:173-217) and the 32-state twin */
size_t orc_block_decode(int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap);
/* mt_rANS32x64_16w_decode.cpp:12-133 and the 32-state twin */
size_t orc_mt_decode(int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap);

/* dispatch helper for ctypes */
size_t orc_decode(int container, int states, unsigned bits, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap);

/* test-only interpreter of the product's decode plans (see hsrans_oracle.c) */
size_t orc_exec_plan(const uint8_t *plan, size_t planLen, const uint8_t *in, size_t inLen, uint8_t *out, size_t outCap);

#ifdef __cplusplus
}
#endif

#endif /* HSRANS_ORACLE_H */
