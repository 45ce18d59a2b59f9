// hsrans_harness — the benchmark / validation loop of the reference's harness (src/main.cpp:841-898: one dry run, N timed
// runs, min and mean MiB/s over the decoded size with MiB = 2^20, memcmp validation, nonzero exit on mismatch, Validate()
// :949) around THIS library's functions, called through include/hsrans_dropin.hpp exactly as main.cpp would call them
// from its _Codecs[] table (src/main.cpp:172-236).  Encoders are the library's scalar host encoders, decoders are the
// gfx950 kernels: "host buffers" is the drop-in decodeFunc signature (PCIe both ways inside the timed call), "device
// resident" is the C-ABI device entry with a sidecar plan, timed with HIP events (what bench.py reports).  For the mt_
// codecs the GPU encoder is timed as well (wall clock around hsrans_encode_device, which synchronises).
//
//   hsrans_harness <file> [--runs N] [--decode-runs N] [--only <substring>] [--bits B] [--interval G] [--test]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/hsrans_dropin.hpp"

namespace hh = hsrans_hip;

typedef size_t (*EncodeWithHist)(const uint8_t *, const size_t, uint8_t *, const size_t, const hh::hist_t *);
typedef size_t (*EncodeNoHist)(const uint8_t *, const size_t, uint8_t *, const size_t);
typedef size_t (*DecodeFunc)(const uint8_t *, const size_t, uint8_t *, const size_t);
typedef size_t (*IndexCapacity)(const size_t);
typedef size_t (*EncodeIndexed)(const uint8_t *, const size_t, uint8_t *, const size_t, uint8_t *, const size_t, size_t *);
typedef size_t (*DecodeIndexed)(const uint8_t *, const size_t, uint8_t *, const size_t, const uint8_t *, const size_t);

struct Codec
{
  const char *name; // the reference's codec names (main.cpp:181-214)
  int container, states;
  uint32_t bits;
  EncodeWithHist enc_hist; // raw
  EncodeNoHist enc;        // block_ / mt_
  DecodeFunc dec;
  DecodeFunc dec_auto;       // runtime dispatch: one dependent chain -> host SIMD decoder, independent blocks -> GPU
  IndexCapacity index_cap;   // the sidecar index through the drop-in names
  EncodeIndexed enc_indexed;
  DecodeIndexed dec_indexed, dec_indexed_pipelined;
};

#define IDX(codec, N) hh::codec##_decode_auto_##N, hh::codec##_index_capacity_##N, hh::codec##_encode_with_index_##N, hh::codec##_decode_hip_with_index_##N, hh::codec##_decode_hip_pipelined_with_index_##N
#define RAW(S, N) {"rANS32x" #S " 16w (raw)", HSRANS_RAW, S, N, hh::rANS32x##S##_16w_encode_scalar_##N, nullptr, hh::rANS32x##S##_16w_decode_hip_##N, IDX(rANS32x##S##_16w, N)}
#define BLK(S, N) {"rANS32x" #S " 16w (variable block size)", HSRANS_BLOCK, S, N, nullptr, hh::block_rANS32x##S##_16w_encode_##N, hh::block_rANS32x##S##_16w_decode_hip_##N, IDX(block_rANS32x##S##_16w, N)}
#define MTB(S, N) {"rANS32x" #S " 16w (independent blocks)", HSRANS_MT, S, N, nullptr, hh::mt_rANS32x##S##_16w_encode_##N, hh::mt_rANS32x##S##_16w_decode_hip_##N, IDX(mt_rANS32x##S##_16w, N)}
#define ALL_BITS(M, S) M(S, 15), M(S, 14), M(S, 13), M(S, 12), M(S, 11), M(S, 10)
static const Codec kCodecs[] = {ALL_BITS(MTB, 64), ALL_BITS(MTB, 32), ALL_BITS(BLK, 64), ALL_BITS(BLK, 32), ALL_BITS(RAW, 64), ALL_BITS(RAW, 32)};

static double now_s()
{
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct Stats
{
  double min_s = 1e30, sum_s = 0;
  int n = 0;
  void add(double s)
  {
    min_s = std::min(min_s, s);
    sum_s += s;
    n++;
  }
  void print(size_t bytes) const
  {
    const double mib = (double)bytes / (1024.0 * 1024.0);
    printf("| min %11.2f MiB/s | mean %11.2f MiB/s | %d runs", mib / min_s, mib / (sum_s / n), n);
  }
};

static bool validate(const uint8_t *got, const uint8_t *want, size_t size)
{
  if (memcmp(got, want, size) == 0)
    return true;
  for (size_t i = 0; i < size; i++)
    if (got[i] != want[i])
    {
      printf("\n  Validation failed: first invalid byte at %zu (0x%02X != 0x%02X)\n", i, got[i], want[i]);
      break;
    }
  return false;
}

int main(int argc, char **argv)
{
  const char *filename = nullptr, *only = nullptr;
  int runs = 4, decode_runs = 16, only_bits = 0;
  uint32_t interval = 32;
  bool test = false;
  for (int i = 1; i < argc; i++)
  {
    if (!strcmp(argv[i], "--runs") && i + 1 < argc)
      runs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--decode-runs") && i + 1 < argc)
      decode_runs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--only") && i + 1 < argc)
      only = argv[++i];
    else if (!strcmp(argv[i], "--bits") && i + 1 < argc)
      only_bits = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--interval") && i + 1 < argc)
      interval = (uint32_t)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--test"))
      test = true;
    else if (argv[i][0] != '-' && filename == nullptr)
      filename = argv[i];
    else
    {
      printf("Invalid Parameter '%s'. Aborting.\n", argv[i]);
      return 1;
    }
  }
  if (filename == nullptr || runs < 1 || decode_runs < 1)
  {
    puts("Usage: hsrans_harness <file> [--runs N] [--decode-runs N] [--only <codec name substring>] [--bits B] [--interval G] [--test]");
    return 1;
  }

  std::vector<uint8_t> input;
  {
    FILE *f = fopen(filename, "rb");
    if (!f)
    {
      puts("Failed to read file.");
      return 1;
    }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size <= 0)
    {
      puts("Invalid File size / failed to read file.");
      fclose(f);
      return 1;
    }
    input.resize((size_t)size);
    if (fread(input.data(), 1, input.size(), f) != input.size())
    {
      puts("Failed to read file.");
      fclose(f);
      return 1;
    }
    fclose(f);
  }
  const size_t n = input.size();

  hsrans_ctx *ctx = hh::default_context();
  if (ctx == nullptr)
  {
    puts("No usable gfx950 device: this library has no CPU decode path.");
    return 3;
  }
  printf("File: '%s' (%zu Bytes)\nDevice: %s  (%s)\n", filename, n, hsrans_ctx_device_name(ctx), hsrans_version());

  size_t cap = 0;
  for (int c = HSRANS_RAW; c <= HSRANS_MT; c++)
    for (int s = 32; s <= 64; s += 32)
      cap = std::max(cap, hsrans_capacity(c, s, n));
  std::vector<uint8_t> compressed(cap), decoded(n);
  std::vector<uint8_t> plan;
  bool all_ok = true;
  hipStream_t stream = nullptr;
  if (hipStreamCreate(&stream) != hipSuccess)
    return 3;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  uint8_t *d_in = nullptr, *d_out = nullptr;
  if (hipMalloc((void **)&d_in, cap + 16) != hipSuccess || hipMalloc((void **)&d_out, n + 16) != hipSuccess)
  {
    puts("Memory allocation failure.");
    return 1;
  }

  for (const Codec &codec : kCodecs)
  {
    if (only != nullptr && strstr(codec.name, only) == nullptr)
      continue;
    if (only_bits != 0 && (uint32_t)only_bits != codec.bits)
      continue;
    printf("\nCodec: %s, %u bit histogram\n", codec.name, codec.bits);

    // ---- encoder: dry run + timed runs ----
    hh::hist_t hist;
    hh::make_hist(&hist, input.data(), n, codec.bits);
    size_t encoded = 0;
    Stats es;
    for (int run = -1; run < runs; run++)
    {
      memset(compressed.data(), 0xCC, cap);
      const double t0 = now_s();
      encoded = codec.enc_hist ? codec.enc_hist(input.data(), n, compressed.data(), cap, &hist) : codec.enc(input.data(), n, compressed.data(), cap);
      const double t1 = now_s();
      if (run >= 0)
        es.add(t1 - t0);
    }
    printf("  %-44s | %6.2f %% ", codec.container == HSRANS_RAW ? "enc scalar" : "encode", 100.0 * (double)encoded / (double)n);
    es.print(n);
    puts("");
    if (encoded == 0)
    {
      puts("  Failed to encode.");
      all_ok = false;
      continue;
    }

    // ---- decoder through the reference's decodeFunc signature (host buffers) ----
    size_t got = 0;
    Stats ds;
    for (int run = -1; run < decode_runs; run++)
    {
      if (run == -1)
        memset(decoded.data(), 0xCC, n);
      const double t0 = now_s();
      got = codec.dec(compressed.data(), encoded, decoded.data(), n);
      const double t1 = now_s();
      if (run >= 0)
        ds.add(t1 - t0);
    }
    printf("  %-44s |          ", "dec MI355X (hip), host buffers");
    ds.print(n);
    const bool ok_host = got == n && validate(decoded.data(), input.data(), n);
    puts(ok_host ? " | valid" : " | FAILED TO VALIDATE");
    all_ok = all_ok && ok_host;

    // ---- the runtime-dispatched entry (one dependent chain -> this library's host SIMD decoder; mt_ -> GPU) ----
    {
      Stats as;
      size_t got_a = 0;
      for (int run = -1; run < decode_runs; run++)
      {
        if (run == -1)
          memset(decoded.data(), 0xCC, n);
        const double t0 = now_s();
        got_a = codec.dec_auto(compressed.data(), encoded, decoded.data(), n);
        const double t1 = now_s();
        if (run >= 0)
          as.add(t1 - t0);
      }
      char label[64];
      snprintf(label, sizeof(label), "dec auto (%s)", codec.container == HSRANS_MT ? "-> MI355X" : hsrans_cpu_level() == 2 ? "-> host avx512" : hsrans_cpu_level() == 1 ? "-> host avx2" : "-> host scalar");
      printf("  %-44s |          ", label);
      as.print(n);
      const bool ok_auto = got_a == n && validate(decoded.data(), input.data(), n);
      puts(ok_auto ? " | valid" : " | FAILED TO VALIDATE");
      all_ok = all_ok && ok_auto;
    }

    // ---- the sidecar index through the drop-in names: encode_with_index, then host buffers + index (one launch, and with the
    //      PCIe legs overlapped over page-locked buffers) ----
    {
      std::vector<uint8_t> index(codec.index_cap(n));
      size_t index_len = 0;
      const size_t enc_i = codec.enc_indexed(input.data(), n, compressed.data(), cap, index.data(), index.size(), &index_len);
      bool ok_i = enc_i == encoded && index_len != 0;
      const bool pinned = ok_i && hsrans_host_register(ctx, compressed.data(), cap) == HSRANS_OK && hsrans_host_register(ctx, decoded.data(), n) == HSRANS_OK;
      for (int variant = 0; variant < 2 && ok_i; variant++)
      {
        Stats is;
        size_t got_i = 0;
        for (int run = -1; run < decode_runs; run++)
        {
          if (run == -1)
            memset(decoded.data(), 0xCC, n);
          const double t0 = now_s();
          got_i = (variant ? codec.dec_indexed_pipelined : codec.dec_indexed)(compressed.data(), enc_i, decoded.data(), n, index.data(), index_len);
          const double t1 = now_s();
          if (run >= 0)
            is.add(t1 - t0);
        }
        char label[80];
        snprintf(label, sizeof(label), variant ? "dec MI355X, host buffers + index, pipelined" : "dec MI355X, host buffers + index (%.1f%%)", 100.0 * (double)index_len / (double)enc_i);
        printf("  %-44s |          ", label);
        is.print(n);
        ok_i = got_i == n && validate(decoded.data(), input.data(), n);
        puts(ok_i ? (pinned ? " | valid (page-locked)" : " | valid") : " | FAILED TO VALIDATE");
      }
      if (pinned)
      {
        (void)hsrans_host_unregister(ctx, compressed.data());
        (void)hsrans_host_unregister(ctx, decoded.data());
      }
      all_ok = all_ok && ok_i;
    }

    // ---- decoder with device-resident stream and output, sidecar plan with checkpoints every `interval` groups ----
    hsrans_encode_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.index_interval = interval;
    plan.resize(hsrans_plan_capacity(codec.container, codec.states, n, interval, 0));
    opts.plan_out = plan.data();
    opts.plan_capacity = plan.size();
    const size_t encoded2 = hsrans_encode_ex(codec.container, codec.states, codec.bits, input.data(), n, compressed.data(), cap, codec.container == HSRANS_RAW ? &hist : nullptr, &opts);
    hsrans_dplan *dp = nullptr;
    bool ok_dev = encoded2 == encoded && hipMemcpy(d_in, compressed.data(), encoded2, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemset(d_out, 0xCC, n) == hipSuccess && hsrans_dplan_create(ctx, plan.data(), opts.plan_size, &dp) == HSRANS_OK;
    Stats vs;
    for (int run = -1; ok_dev && run < decode_runs; run++)
    {
      (void)hipEventRecord(e0, stream);
      ok_dev = hsrans_decode_device(ctx, dp, d_in, encoded2, d_out, n, stream) == HSRANS_OK;
      (void)hipEventRecord(e1, stream);
      ok_dev = ok_dev && hipEventSynchronize(e1) == hipSuccess;
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (run >= 0)
        vs.add(ms * 1e-3);
    }
    if (ok_dev)
    {
      ok_dev = hsrans_dplan_status(ctx, dp, stream) == HSRANS_OK && hipMemcpy(decoded.data(), d_out, n, hipMemcpyDeviceToHost) == hipSuccess;
      char label[64];
      snprintf(label, sizeof(label), "dec MI355X (hip), device resident, G=%u", interval);
      printf("  %-44s |          ", label);
      vs.print(n);
      ok_dev = ok_dev && validate(decoded.data(), input.data(), n);
      puts(ok_dev ? " | valid" : " | FAILED TO VALIDATE");
    }
    else
      puts("  device-resident decode failed.");
    if (dp)
      hsrans_dplan_destroy(dp);
    all_ok = all_ok && ok_dev;

    // ---- the GPU encoder (mt_ container only): input and stream device-resident, plan built on the device; validated by
    //      decoding its stream with the plan it returned ----
    if (codec.container == HSRANS_MT)
    {
      uint8_t *d_src = d_out; // the decoded bytes of the leg above are the input again
      bool ok_enc = hipMemcpy(d_src, input.data(), n, hipMemcpyHostToDevice) == hipSuccess;
      Stats gs;
      size_t stream_bytes = 0;
      hsrans_dplan *dpe = nullptr;
      for (int run = -1; ok_enc && run < decode_runs; run++)
      {
        if (dpe)
          hsrans_dplan_destroy(dpe);
        dpe = nullptr;
        const double t0 = now_s();
        stream_bytes = hsrans_encode_device(ctx, HSRANS_MT, codec.states, codec.bits, d_src, n, d_in, cap, 65536, interval, stream, &dpe); // synchronises
        const double t1 = now_s();
        ok_enc = stream_bytes != 0;
        if (run >= 0)
          gs.add(t1 - t0);
      }
      uint8_t *d_back = nullptr;
      ok_enc = ok_enc && hipMalloc((void **)&d_back, n + 16) == hipSuccess && hsrans_decode_device(ctx, dpe, d_in, stream_bytes, d_back, n, stream) == HSRANS_OK &&
               hsrans_dplan_status(ctx, dpe, stream) == HSRANS_OK && hipMemcpy(decoded.data(), d_back, n, hipMemcpyDeviceToHost) == hipSuccess;
      if (ok_enc)
      {
        printf("  %-44s | %6.2f %% ", "enc MI355X (hip), 64 KiB blocks + plan", 100.0 * (double)stream_bytes / (double)n);
        gs.print(n);
        ok_enc = validate(decoded.data(), input.data(), n);
        puts(ok_enc ? " | valid" : " | FAILED TO VALIDATE");
      }
      else
        puts("  GPU encode failed.");
      if (dpe)
        hsrans_dplan_destroy(dpe);
      if (d_back)
        (void)hipFree(d_back);
      all_ok = all_ok && ok_enc;
    }
    if (test && !all_ok)
      break;
  }

  (void)hipFree(d_in);
  (void)hipFree(d_out);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(stream);
  puts(all_ok ? "\nAll codecs validated." : "\nFailed to validate.");
  return all_ok ? 0 : 1;
}
