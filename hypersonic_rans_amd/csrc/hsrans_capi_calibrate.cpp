// hsrans_capi_calibrate.cpp — per-device fit of the one-chain-per-wave index: hsrans_ctx_calibrate, hsrans_ctx_calibrate_runs.
// Part of the C ABI of libhsrans_hip.so (include/hsrans_hip.h); split out of hsrans_capi.cpp in round 5 by concern.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/hsrans_hip.h"
#include "hsrans_host.h"
#include "hsrans_cpu.h"
#include "hsrans_encode.h"
#include "hsrans_kernels.h"

using namespace hsrans;

#include "hsrans_internal.h"
#include "hsrans_batch.h"


extern "C"
{

// ---- per-device fit of the one-chain-per-wave index ----------------------------------------------------------------
// The SIMDs serve their oldest wave first and the decode loop is issue-bound, so the 8 wave classes of the one-chain-per-wave
// launch (workgroup in the grid's first / second half x wave / 4) decode at different rates and the index gives them chains of
// different lengths (hsrans_index_boundaries).  The lengths compiled in were fitted on one box; how early the second workgroup
// of a CU becomes resident, and with it the right lengths, differs from box to box by a few per cent (r03: classes of the
// second half done 2 us before the first half's on another box).  This fits them to the context's own device: synthetic
// enwik8-shaped bytes, encoded once on the host; per iteration an index at the current lengths (one host decode pass that
// records the checkpoints), a few launches whose waves leave their finish time, and every class length moved towards
// length x (mean finish / class finish) ^ 0.8.  The best lengths seen stay in the context: hsrans_index_boundaries(ctx, ...)
// and the launch info (class_weights) use them from then on.
static int calibrate_impl(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, uint32_t copies, hsrans_calibration *report)
{
  if (ctx == nullptr || bits < 10 || bits > 12 || copies < 1 || copies > 16) // (the fitted kernel is k_decode_direct<3>: 64 states, 8-byte table, one chain per wave)
    return HSRANS_E_ARG;
  if (iterations == 0)
    iterations = copies == 1 ? 4 : 7; // (a longer run starts from lengths fitted for another run length: further to go)
  if (iterations > 16)
    iterations = 16;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return HSRANS_E_HIP;
  const size_t n = (size_t)48 << 20;
  std::vector<uint8_t> data(n), stream(capacity(HSRANS_RAW, 64, n));
  {
    // Zipf(1.2) over 205 symbols through a 65,536-entry inverse-CDF table, xorshift64* indices: the shape of the benchmark's data
    std::vector<uint8_t> inv(65536);
    double w[205], sum = 0;
    for (int r = 0; r < 205; r++)
      sum += (w[r] = 1.0 / pow((double)(r + 1), 1.2));
    double acc = 0;
    size_t at = 0;
    for (int r = 0; r < 205; r++)
    {
      acc += w[r] / sum;
      const size_t end = r == 204 ? 65536 : (size_t)(acc * 65536.0);
      for (; at < end && at < 65536; at++)
        inv[at] = (uint8_t)((r * 37 + 11) & 0xFF); // (any fixed symbol -> byte map)
    }
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i += 4)
    {
      x ^= x >> 12, x ^= x << 25, x ^= x >> 27;
      const uint64_t v = x * 0x2545F4914F6CDD1Dull;
      data[i] = inv[v & 0xFFFF], data[i + 1] = inv[(v >> 16) & 0xFFFF], data[i + 2] = inv[(v >> 32) & 0xFFFF], data[i + 3] = inv[v >> 48];
    }
  }
  const size_t stream_len = encode(HSRANS_RAW, 64, bits, data.data(), n, stream.data(), stream.size(), nullptr, nullptr);
  if (stream_len == 0)
    return HSRANS_E_FORMAT;
  // Scope guards first: whatever leaves this function — a return, or an exception on its way to the handler below (bad_alloc from
  // one of the vectors) — frees the device buffers, destroys the plan of the iteration in flight and puts the context's launch
  // geometry back (the iterations overwrite it with trial lengths).  The context's lock is held throughout: other entries read
  // ctx->geom (hsrans_index_boundaries, every launch_shape).
  std::lock_guard<std::mutex> calibration_guard(ctx->lock);
  struct DeviceBuffers
  {
    uint8_t *stream = nullptr, *out = nullptr;
    uint64_t *finish = nullptr;
    hsrans_dplan *dplan = nullptr;
    hsrans_batch *batch = nullptr;        // copies > 1: the iteration's batch of `copies` members and its other members' plans
    std::vector<hsrans_dplan *> more;
    void drop_iteration()
    {
      if (batch)
        hsrans_dplan_batch_destroy(batch);
      batch = nullptr;
      for (hsrans_dplan *d : more)
        hsrans_dplan_destroy(d);
      more.clear();
      if (dplan)
        hsrans_dplan_destroy(dplan);
      dplan = nullptr;
    }
    ~DeviceBuffers()
    {
      drop_iteration();
      if (stream)
        (void)hipFree(stream);
      if (out)
        (void)hipFree(out);
      if (finish)
        (void)hipFree(finish);
    }
  } dev;
  struct GeomRestore
  {
    hsrans_ctx *ctx;
    DeviceGeom saved;
    ~GeomRestore() { ctx->geom = saved; }
  } geom_restore{ctx, ctx->geom};
  uint8_t *&d_stream = dev.stream, *&d_out = dev.out;
  uint64_t *&d_finish = dev.finish;
  std::vector<uint64_t> groups(1 << 16), finish;
  std::vector<uint8_t> plan(plan_capacity_chains(HSRANS_RAW, 64, n, 1 << 14, 0));
  uint32_t best_w[8] = {}, cur_w[8];
  double best_last = 1e30, first_last = 0, first_spread = 0, best_spread = 0;
  int rc = HSRANS_E_HIP;
  do
  {
    // The launches that are measured look like the ones the fit is for: back to back, and every one on another (stream, output)
    // pair of a set larger than the 256 MB Infinity Cache — a lone launch on warm buffers shows the youngest wave class only
    // 0.3 us late, a launch of a sustained rotation 1.5 us (its prologue loads and its stores are served last), and chains
    // fitted to the former leave that class to finish the rotated launch alone.  (One pair if the device cannot spare 400 MB.)
    const size_t stream_stride = ((stream_len + 15) / 16 * 16 + 255) / 256 * 256 + 256;
    uint32_t pairs = copies > 1 ? copies + 2 : 5; // (a batch launch writes `copies` outputs: the next launch's are other buffers)
    if (hipMalloc((void **)&d_stream, pairs * stream_stride) != hipSuccess || hipMalloc((void **)&d_out, pairs * n) != hipSuccess)
    {
      (void)hipGetLastError();
      if (d_stream)
        (void)hipFree(d_stream);
      d_stream = nullptr;
      pairs = copies;
      if (hipMalloc((void **)&d_stream, pairs * stream_stride) != hipSuccess || hipMalloc((void **)&d_out, pairs * n) != hipSuccess)
        break;
    }
    bool uploaded = true;
    for (uint32_t k = 0; k < pairs && uploaded; k++)
      uploaded = hipMemcpy(d_stream + k * stream_stride, stream.data(), stream_len, hipMemcpyHostToDevice) == hipSuccess;
    if (!uploaded)
      break;
    // start from the lengths in use (the compiled-in fit, or an earlier calibration)
    {
      PlanHeader h{};
      h.states = 64, h.bits = bits, h.shared_hist = 1, h.n_chains = 1u << 30;
      const TableChoice tc = choose_table(bits, 64, true);
      const LaunchShape L = launch_shape(h, ctx->geom, true, tc.mode, 0, false, true, tc.dual);
      if (L.dual || L.waves != 16 || L.grid <= ctx->geom.num_cus) // not the launch shape the classes are defined for: nothing to fit
      {
        rc = HSRANS_E_ARG;
        break;
      }
      for (int k = 0; k < 8; k++)
        cur_w[k] = L.weights[k];
    }
    bool failed = false;
    for (uint32_t it = 0; it < iterations && !failed; it++)
    {
      ctx->geom.have_direct_weights = 1;
      ctx->geom.n_weight_sets = 0; // (the trial lengths, not an interpolation of earlier fits)
      for (int k = 0; k < 8; k++)
        ctx->geom.direct_weights[k] = cur_w[k];
      const uint64_t T = (n - 63) / 64; // whole groups (hsrans_index_boundaries)
      size_t chains;
      BatchShape bshape{};
      if (copies == 1)
        chains = direct_boundaries(ctx->geom, 64, bits, T, groups.data(), groups.size());
      else
      {
        // `copies` members of this one stream in one launch: every member indexed for its share of the wave slots (batch_boundaries)
        bshape = batch_direct_shape(ctx->geom, bits, 0);
        std::vector<uint64_t> totals(copies, T);
        chains = batch_boundaries(totals.data(), copies, 0, bshape.grid, bshape.waves, cur_w, groups.data(), groups.size());
      }
      if (chains < 2)
      {
        failed = true;
        break;
      }
      const size_t plan_len = cpu::index_build(cpu::best_level(), 1, HSRANS_RAW, 64, bits, stream.data(), stream_len, groups.data(), chains - 1, plan.data(), plan.size());
      hsrans_dplan *&dp = dev.dplan; // (owned by the guard until the iteration hands it back)
      if (plan_len == 0 || hsrans_dplan_create(ctx, plan.data(), plan_len, &dp) != HSRANS_OK)
      {
        failed = true;
        break;
      }
      for (uint32_t k = 1; k < copies && !failed; k++)
      {
        hsrans_dplan *extra = nullptr;
        if (hsrans_dplan_create(ctx, plan.data(), plan_len, &extra) != HSRANS_OK)
          failed = true;
        else
          dev.more.push_back(extra);
      }
      if (!failed && copies > 1)
      {
        std::vector<hsrans_dplan *> all{dp};
        all.insert(all.end(), dev.more.begin(), dev.more.end());
        // (the batch's own weights must be the trial lengths: HSRANS_BATCH_WEIGHTS aside, batch_direct_shape reads ctx->geom, set above)
        if (hsrans_dplan_batch_create(ctx, all.data(), copies, &dev.batch) != HSRANS_OK || dev.batch->direct.size() != 1 || !dev.batch->solo.empty())
          failed = true;
      }
      if (failed)
        break;
      const uint32_t W = copies == 1 ? (uint32_t)chains : bshape.grid * bshape.waves; // one chain per wave
      if (W > (1u << 14)) // (the finish-time buffer below is sized for 16,384 waves: twice an MI355X)
      {
        failed = true;
        break;
      }
      const size_t finish_stride = ((size_t)1 << 14) + 1; // words per launch: one finish time per wave + the first wave's entry
      const uint32_t batch = pairs == 1 ? 4 : 2 * pairs;  // launches per batch, back to back, launch l on pair l % pairs
      if (d_finish == nullptr && hipMalloc((void **)&d_finish, (size_t)batch * finish_stride * 8) != hipSuccess)
        failed = true;
      double cls_t[8] = {}, cls_n[8] = {}, last = 0;
      // Rounds of back-to-back launches; only the last one is measured.  The ones before it run until the device has been busy for
      // 20 ms (and at least twice): every iteration begins with host work (the index pass, the device plans) during which the GPU
      // idles, and a GPU that wakes from idle runs the first ~12 ms at other clocks than it then keeps (tools/settle_probe.py,
      // profiles/r04_settle.txt) — lengths fitted in that transient left the youngest class 4 % short on some boxes (round 5:
      // the classes of a 4 x 100 MB batch finished 5 us apart after a fit whose own last round had them within 0.5 us).
      const auto t_busy = std::chrono::steady_clock::now();
      bool measured = false;
      for (int round = 0; !measured && !failed; round++)
      {
        const bool settle = round < 2 || std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_busy).count() < 20.0;
        measured = !settle;
        failed = hipMemset(d_finish, 0, (size_t)batch * finish_stride * 8) != hipSuccess;
        for (uint32_t l = 0; l < batch && !failed; l++)
        {
          if (copies == 1)
          {
            dp->d_finish = d_finish + l * finish_stride;
            failed = dplan_launch(dp, d_stream + (l % pairs) * stream_stride, stream_len, d_out + (size_t)(l % pairs) * n, n, nullptr) != HSRANS_OK;
            continue;
          }
          const void *ins[16];
          void *outs[16];
          size_t in_len[16], out_cap[16];
          for (uint32_t k = 0; k < copies; k++)
          {
            const uint32_t buf = (l * copies + k) % pairs;
            ins[k] = d_stream + buf * stream_stride, in_len[k] = stream_len;
            outs[k] = d_out + (size_t)buf * n, out_cap[k] = n;
          }
          if (dev.batch->finish_owned) // (HSRANS_BATCH_STAMPS=1 gave the batch a buffer of its own: this fit uses its own)
          {
            (void)hipFree(dev.batch->d_finish);
            dev.batch->finish_owned = false;
          }
          dev.batch->d_finish = d_finish + l * finish_stride;
          failed = hsrans_decode_device_batch(ctx, dev.batch, ins, in_len, outs, out_cap, nullptr) != HSRANS_OK;
        }
        failed = hipDeviceSynchronize() != hipSuccess || failed; // (nothing may still be writing the buffers, whatever failed)
        if (failed || settle)
          continue;
        finish.resize((size_t)batch * finish_stride);
        if (hipMemcpy(finish.data(), d_finish, finish.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        {
          failed = true;
          break;
        }
        const uint32_t waves = copies == 1 ? dp->info.waves_per_block : bshape.waves, grid = copies == 1 ? dp->info.grid : bshape.grid, first_half = (grid + 1) / 2;
        if (waves != 16 || (uint64_t)grid * waves != W)
        {
          failed = true;
          break;
        }
        for (uint32_t l = 0; l < batch; l++)
        {
          const uint64_t *f = finish.data() + l * finish_stride;
          double launch_last = 0;
          for (uint32_t w = 0; w < W; w++)
          {
            const double t = (double)(f[w] - f[W]) / 100.0; // us
            const uint32_t cls = (w / waves >= first_half ? 4 : 0) + (w % waves) / 4;
            cls_t[cls] += t, cls_n[cls] += 1;
            launch_last = t > launch_last ? t : launch_last;
          }
          last += launch_last / batch; // (mean over the launches of each launch's last wave: one late wave in one launch does not decide)
        }
      }
      dp->d_finish = nullptr;
      uint32_t status_ok = hsrans_dplan_status(ctx, dp, nullptr) == HSRANS_OK;
      for (hsrans_dplan *extra : dev.more)
        status_ok = hsrans_dplan_status(ctx, extra, nullptr) == HSRANS_OK && status_ok;
      dev.drop_iteration();
      if (failed || !status_ok)
      {
        failed = true;
        break;
      }
      double mean = 0, lo = 1e30, hi = 0;
      for (int k = 0; k < 8; k++)
      {
        cls_t[k] /= cls_n[k] > 0 ? cls_n[k] : 1;
        mean += cls_t[k] / 8;
        lo = cls_t[k] < lo ? cls_t[k] : lo, hi = cls_t[k] > hi ? cls_t[k] : hi;
      }
      if (it == 0)
        first_last = last, first_spread = hi - lo;
      if (last < best_last)
      {
        best_last = last, best_spread = hi - lo;
        for (int k = 0; k < 8; k++)
          best_w[k] = cur_w[k];
      }
      if (report)
        for (int k = 0; k < 8; k++)
          report->class_finish_us_last_iteration[k] = cls_t[k];
      double nw[8], s = 0;
      for (int k = 0; k < 8; k++)
        s += (nw[k] = (double)cur_w[k] * pow(mean / cls_t[k], 0.8));
      for (int k = 0; k < 8; k++)
        cur_w[k] = (uint32_t)(nw[k] * 8000.0 / s + 0.5);
    }
    if (failed)
      break;
    rc = HSRANS_OK;
  } while (false);
  if (rc == HSRANS_OK)
  {
    // what the guard puts back: the geometry as it was, with the fitted lengths — as the default set (the one-stream fit) and as
    // the set of this run length (groups per wave)
    DeviceGeom &g = geom_restore.saved;
    if (copies == 1)
    {
      g.have_direct_weights = 1;
      for (int k = 0; k < 8; k++)
        g.direct_weights[k] = best_w[k];
    }
    const uint32_t run = (uint32_t)((uint64_t)copies * ((n - 63) / 64) / (2 * 16 * (uint64_t)g.num_cus));
    uint32_t at = 0;
    while (at < g.n_weight_sets && g.set_run[at] < run)
      at++;
    if (!(at < g.n_weight_sets && g.set_run[at] == run))
    {
      if (g.n_weight_sets == 4) // (full: the nearest one goes)
        at = at < 4 ? at : 3;
      else
      {
        for (uint32_t k = g.n_weight_sets; k > at; k--)
        {
          g.set_run[k] = g.set_run[k - 1];
          memcpy(g.set_weights[k], g.set_weights[k - 1], sizeof(g.set_weights[k]));
        }
        g.n_weight_sets++;
      }
    }
    g.set_run[at] = run;
    for (int k = 0; k < 8; k++)
      g.set_weights[at][k] = best_w[k];
    if (report)
    {
      for (int k = 0; k < 8; k++)
        report->class_weights[k] = best_w[k];
      report->last_wave_us_before = first_last, report->last_wave_us_after = best_last;
      report->class_spread_us_before = first_spread, report->class_spread_us_after = best_spread;
      report->iterations = iterations;
      report->bytes = (uint64_t)copies * n;
    }
  }
  return rc;
}

int hsrans_ctx_calibrate(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, hsrans_calibration *report)
try
{
  return calibrate_impl(ctx, bits, iterations, 1, report);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}

int hsrans_ctx_calibrate_runs(hsrans_ctx *ctx, uint32_t bits, uint32_t iterations, uint32_t copies, hsrans_calibration *report)
try
{
  return calibrate_impl(ctx, bits, iterations, copies, report);
}
catch (...) // (std::bad_alloc and friends: nothing is thrown across the C ABI)
{
  return HSRANS_E_HIP;
}


} // extern "C"
