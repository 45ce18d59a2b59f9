"""The C ABI used from plain C: a caller written in C99 against include/hsrans_hip.h alone — no C++, no Python, no torch — that does what
a maintainer of the reference would do with the entries this round added: encodes three files' worth of data with
sidecar indexes, decodes them by ONE launch (hsrans_dplan_batch_create / hsrans_decode_device_batch), asks hsrans_shard_layout how one
mt_ stream splits over 4 ranks x 2 sub-runs, and runs that stream through hsrans_comm_create / hsrans_sharded_create /
hsrans_decode_sharded in a world of one rank (RCCL bound at run time).  Compiled with -std=c99 -Wall -Wextra -Werror -pedantic: the
header must be valid C; validated with memcmp like src/main.cpp:891-897; exit code != 0 on any mismatch.
The reference's counterpart of these entries: the per-file loop src/main.cpp:841-898 and the thread-pool fan-out behind
mt_rANS32x64_16w_decode_mt (src/mt_rANS32x64_16w.h:23-28)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLER = r"""
#include "hsrans_hip.h"
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond, code) do { if (!(cond)) { fprintf(stderr, "c caller: line %d: %s\n", __LINE__, #cond); return code; } } while (0)

static void fill(uint8_t *d, size_t n, unsigned seed)
{
  size_t i;
  unsigned long long x = 0x9E3779B97F4A7C15ull * (seed + 1);
  for (i = 0; i < n; i++)
  {
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    d[i] = (uint8_t)(((x * 0x2545F4914F6CDD1Dull) >> 56) % ((i / 50000) % 2 ? 7 : 200)); /* skewed, two regimes */
  }
}

int main(void)
{
  enum { K = 3 };
  const size_t sizes[K] = {1500000, 700001, 2200000};
  hsrans_ctx *ctx = NULL;
  hsrans_dplan *plans[K];
  hsrans_batch *batch = NULL;
  hsrans_batch_info info;
  uint8_t *data[K], *stream[K];
  size_t stream_len[K];
  const void *d_in[K];
  void *d_out[K];
  size_t out_cap[K];
  int status[K], k;

  if (hsrans_ctx_create(0, &ctx) != HSRANS_OK)
    return 4; /* no gfx950 device: there is no CPU path behind these entries */
  for (k = 0; k < K; k++)
  {
    hsrans_encode_opts o;
    uint8_t *plan;
    void *di = NULL, *dout = NULL;
    size_t cap = hsrans_capacity(HSRANS_RAW, 64, sizes[k]), pcap = hsrans_plan_capacity(HSRANS_RAW, 64, sizes[k], 16, 0);
    data[k] = (uint8_t *)malloc(sizes[k]);
    stream[k] = (uint8_t *)malloc(cap);
    plan = (uint8_t *)malloc(pcap);
    CHECK(data[k] && stream[k] && plan, 2);
    fill(data[k], sizes[k], (unsigned)k);
    memset(&o, 0, sizeof(o));
    o.index_interval = 16;
    o.plan_out = plan;
    o.plan_capacity = pcap;
    stream_len[k] = hsrans_encode_ex(HSRANS_RAW, 64, 11, data[k], sizes[k], stream[k], cap, NULL, &o);
    CHECK(stream_len[k] != 0 && o.plan_size != 0, 2);
    CHECK(hsrans_dplan_create(ctx, plan, o.plan_size, &plans[k]) == HSRANS_OK, 2);
    CHECK(hipMalloc(&di, stream_len[k] + 64) == hipSuccess && hipMalloc(&dout, sizes[k] + 16) == hipSuccess, 2);
    CHECK(hipMemcpy(di, stream[k], stream_len[k], hipMemcpyHostToDevice) == hipSuccess, 2);
    d_in[k] = di;
    d_out[k] = dout;
    out_cap[k] = sizes[k];
    free(plan);
  }
  /* K independent streams, ONE launch */
  CHECK(hsrans_dplan_batch_create(ctx, plans, K, &batch) == HSRANS_OK, 3);
  CHECK(hsrans_dplan_batch_info(batch, &info) == HSRANS_OK && info.members == K && info.launches == 1 && info.direct_members == K, 3);
  CHECK(hsrans_decode_device_batch(ctx, batch, d_in, stream_len, d_out, out_cap, NULL) == HSRANS_OK, 3);
  CHECK(hsrans_dplan_batch_status(ctx, batch, NULL, status) == HSRANS_OK, 3);
  for (k = 0; k < K; k++)
  {
    uint8_t *back = (uint8_t *)malloc(sizes[k]);
    CHECK(back && status[k] == HSRANS_OK, 3);
    CHECK(hipMemcpy(back, d_out[k], sizes[k], hipMemcpyDeviceToHost) == hipSuccess, 3);
    CHECK(memcmp(back, data[k], sizes[k]) == 0, 5); /* decoded size == file size && memcmp, as src/main.cpp:891-897 */
    free(back);
  }
  /* a wrong capacity for one member: nothing is launched */
  out_cap[1] = 10;
  CHECK(hsrans_decode_device_batch(ctx, batch, d_in, stream_len, d_out, out_cap, NULL) == HSRANS_E_FORMAT, 3);
  out_cap[1] = sizes[1];
  hsrans_dplan_batch_destroy(batch);

  /* one mt_ stream over the ranks of a communicator: the layout alone, then a world of one rank */
  {
    const size_t n = sizes[2];
    size_t cap = hsrans_capacity(HSRANS_MT, 64, n), pcap = hsrans_plan_capacity(HSRANS_MT, 64, n, 32, 65536), m;
    uint8_t *s = (uint8_t *)malloc(cap), *plan = (uint8_t *)malloc(pcap), *back = (uint8_t *)malloc(n);
    hsrans_encode_opts o;
    hsrans_shard shards[4 * 2];
    uint64_t windows[2 * 4], covered = 0;
    const double weights[4] = {2.0, 1.0, 1.0, 1.0};
    uint8_t id[HSRANS_COMM_ID_BYTES];
    hsrans_comm *comm = NULL;
    hsrans_sharded *sh = NULL;
    hsrans_sharded_info_t si;
    void *d_window = NULL, *d_full = NULL;
    int r;
    CHECK(s && plan && back, 2);
    memset(&o, 0, sizeof(o));
    o.block_size = 65536;
    o.index_interval = 32;
    o.plan_out = plan;
    o.plan_capacity = pcap;
    m = hsrans_encode_ex(HSRANS_MT, 64, 11, data[2], n, s, cap, NULL, &o);
    CHECK(m != 0, 2);
    CHECK(hsrans_shard_layout(plan, o.plan_size, 4, 2, weights, shards, windows) == HSRANS_OK, 6);
    for (r = 0; r < 8; r++)
      covered += shards[r].out_end - shards[r].out_begin;
    CHECK(covered == n && shards[0].out_begin == 0 && shards[7].out_end == n && windows[1] <= m, 6);
    CHECK(shards[1].out_end - shards[0].out_begin > (n * 2) / 5 - 70000 && shards[1].out_end - shards[0].out_begin < (n * 2) / 5 + 70000, 6); /* rank 0: 2/5 of the bytes */
    CHECK(hsrans_comm_unique_id(id) == HSRANS_OK && hsrans_comm_rccl_version() > 0, 7);
    CHECK(hsrans_comm_create(ctx, id, 0, 1, &comm) == HSRANS_OK && hsrans_comm_world(comm) == 1 && hsrans_comm_rank(comm) == 0, 7);
    CHECK(hsrans_sharded_create(ctx, comm, plan, o.plan_size, 3, NULL, -1, &sh) == HSRANS_OK, 7);
    CHECK(hsrans_sharded_info(sh, &si, NULL, 0) == HSRANS_OK && si.world == 1 && si.parts == 3 && si.out_base == 0 && si.out_length == n && si.window_end <= m, 7);
    CHECK(hipMalloc(&d_window, (size_t)(si.window_end - si.window_begin) + 64) == hipSuccess && hipMalloc(&d_full, n + 16) == hipSuccess, 2);
    CHECK(hipMemcpy(d_window, s + si.window_begin, (size_t)(si.window_end - si.window_begin), hipMemcpyHostToDevice) == hipSuccess, 2);
    CHECK(hsrans_decode_sharded(sh, d_window, d_full, HSRANS_SHARD_DECODE_AND_EXCHANGE, NULL) == HSRANS_OK, 7);
    CHECK(hsrans_sharded_status(sh, NULL) == HSRANS_OK, 7);
    CHECK(hipMemcpy(back, d_full, n, hipMemcpyDeviceToHost) == hipSuccess && memcmp(back, data[2], n) == 0, 5);
    hsrans_sharded_destroy(sh);
    hsrans_comm_destroy(comm);
    free(s); free(plan); free(back);
  }
  for (k = 0; k < K; k++)
    hsrans_dplan_destroy(plans[k]);
  hsrans_ctx_destroy(ctx);
  printf("c caller: 3 streams in one launch, shard layout, sharded decode over RCCL (world 1): all validated\n");
  return 0;
}
"""


def _build(tmp_path):
    src = tmp_path / "caller.c"
    src.write_text(CALLER)
    exe = tmp_path / "caller"
    lib = os.path.join(ROOT, "hypersonic_rans_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-isystem", "/opt/rocm/include", str(src), "-o", str(exe),
           "-L" + lib, "-lhsrans_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    return str(exe)


def test_the_header_is_valid_c99_and_a_c_caller_links(tmp_path):
    """CPU side: compiles (-std=c99 -pedantic -Werror), links, and — there being no GPU here — the binary says so with exit code 4."""
    import torch

    exe = _build(tmp_path)
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 4, (r.returncode, r.stderr[-500:])


@pytest.mark.gpu
def test_c_caller_batch_layout_and_sharded_decode_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "all validated" in r.stdout
