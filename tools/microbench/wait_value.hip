// wait_value.hip — does hipStreamWaitValue32 work on this device, on which kinds of memory, and what does a completion word cost?
// (round 6: a rank's sub-runs are ONE launch; the exchange's stream waits for sub-run k's completion word instead of an event
// between launches — csrc/hsrans_comm.cpp.)  Stream A runs a kernel that spins ~delay_us and then publishes `value` to a word;
// stream B waits for the word (hipStreamWaitValue32 ... Gte) and then runs a kernel that records the time; B's stamp must come
// after A's store.  Build: hipcc --offload-arch=gfx950 wait_value.hip -o wait_value
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x)                                                                                                                                     \
  do                                                                                                                                                 \
  {                                                                                                                                                  \
    hipError_t e_ = (x);                                                                                                                             \
    if (e_ != hipSuccess)                                                                                                                            \
    {                                                                                                                                                \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                                                                                               \
      (void)hipGetLastError();                                                                                                                       \
      return 1;                                                                                                                                      \
    }                                                                                                                                                \
  } while (0)

__global__ void k_publish(uint32_t *word, uint32_t value, uint64_t spin_ticks, uint64_t *stamp, uint8_t *payload, size_t payload_bytes)
{
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  // something to flush: the payload written with plain stores, as the decoder's output would be
  for (size_t i = threadIdx.x; i < payload_bytes / 16; i += blockDim.x)
    ((uint4 *)payload)[i] = make_uint4(value, value, value, value);
  while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks)
    ;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    stamp[0] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void k_after(uint64_t *stamp, const uint8_t *payload, uint32_t *seen)
{
  stamp[1] = __builtin_amdgcn_s_memrealtime();
  seen[0] = ((const uint32_t *)payload)[0];
}

static int run(const char *what, uint32_t *word, int reps)
{
  hipStream_t a, b;
  CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  uint64_t *stamp;
  uint32_t *seen;
  uint8_t *payload;
  const size_t payload_bytes = 1 << 20;
  CHECK(hipMalloc((void **)&stamp, 16));
  CHECK(hipMalloc((void **)&seen, 4));
  CHECK(hipMalloc((void **)&payload, payload_bytes));
  double worst = -1e9, sum = 0;
  int bad = 0;
  for (int r = 1; r <= reps; r++)
  {
    // B first: it must block until A publishes
    hipError_t e = hipStreamWaitValue32(b, word, (uint32_t)r, hipStreamWaitValueGte, 0xFFFFFFFFu);
    if (e != hipSuccess)
    {
      printf("%-28s hipStreamWaitValue32 -> %s\n", what, hipGetErrorString(e));
      (void)hipGetLastError();
      return 1;
    }
    hipLaunchKernelGGL(k_after, dim3(1), dim3(64), 0, b, stamp, payload, seen);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, a, word, (uint32_t)r, (uint64_t)2000 /* 20 us at 100 MHz */, stamp, payload, payload_bytes);
    CHECK(hipStreamSynchronize(a));
    CHECK(hipStreamSynchronize(b));
    uint64_t st[2];
    uint32_t s = 0;
    CHECK(hipMemcpy(st, stamp, 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost));
    const double us = ((double)st[1] - (double)st[0]) / 100.0;
    if (us < 0 || s != (uint32_t)r)
      bad++;
    worst = us > worst ? us : worst;
    sum += us;
  }
  printf("%-28s ok: %d reps, consumer starts %.1f us (mean) / %.1f us (worst) after the word is published; %d out of order or stale\n", what, reps, sum / reps, worst, bad);
  (void)hipFree(stamp);
  (void)hipFree(seen);
  (void)hipFree(payload);
  (void)hipStreamDestroy(a);
  (void)hipStreamDestroy(b);
  return bad != 0;
}

int main()
{
  int can = -1;
  CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  uint32_t *w = nullptr;
  if (hipMalloc((void **)&w, 64) == hipSuccess && hipMemset(w, 0, 64) == hipSuccess)
    run("hipMalloc", w, 20);
  uint64_t *sig = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(8, hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
  if (e == hipSuccess && hipMemset(sig, 0, 8) == hipSuccess)
    run("signal memory", (uint32_t *)sig, 20);
  (void)hipGetLastError();
  uint32_t *fine = nullptr;
  e = hipExtMallocWithFlags((void **)&fine, 64, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(64, hipDeviceMallocFinegrained) -> %s\n", hipGetErrorString(e));
  if (e == hipSuccess && hipMemset(fine, 0, 64) == hipSuccess)
    run("fine-grained device memory", fine, 20);
  (void)hipGetLastError();
  uint32_t *host = nullptr;
  if (hipHostMalloc((void **)&host, 64, hipHostMallocDefault) == hipSuccess)
  {
    host[0] = 0;
    run("page-locked host memory", host, 20);
  }
  return 0;
}
