// Read-bandwidth calibration for the scatter-max roofline: how fast can ANY
// kernel stream a [E, C] fp32 matrix once (600 MB, larger than the 256 MB
// Infinity Cache)?  Variants: plain / non-temporal dwordx4 loads, 4 or 8 loads
// in flight per lane, contiguous-per-wave vs grid-stride.
//   hipcc -O3 --offload-arch=gfx950 -o tools/hbm_read_peak tools/hbm_read_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, bool CHUNKED>
__global__ __launch_bounds__(256) void read_kernel(const v4 *__restrict__ p,
                                                   size_t n, float *out) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * 256;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  if (CHUNKED) {
    // a wave owns a contiguous range (like scatter_max's rows_per_wave)
    const size_t wave = tid >> 6, nwaves = nthreads >> 6;
    const int lane = threadIdx.x & 63;
    const size_t per = (n / 64 + nwaves - 1) / nwaves;  // 64-lane rows
    size_t r0 = wave * per, r1 = r0 + per;
    if (r1 > n / 64) r1 = n / 64;
    for (size_t r = r0; r < r1; r += UNROLL) {
      v4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        size_t rr = r + u < r1 ? r + u : r1 - 1;
        const v4 *q = p + rr * 64 + lane;
        v[u] = NT ? __builtin_nontemporal_load(q) : *q;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  } else {
    for (size_t i = tid; i < n; i += nthreads * UNROLL) {
      v4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        size_t j = i + u * nthreads;
        if (j >= n) j = n - 1;
        v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

template <int UNROLL, bool NT, bool CHUNKED>
void run(const char *name, const v4 *bufs[3], size_t n, float *out, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((read_kernel<UNROLL, NT, CHUNKED>), dim3(blocks), dim3(256),
                       0, 0, bufs[i % 3], n, out);
  float best = 1e9f, sum = 0;
  const int reps = 12;
  for (int i = 0; i < reps; ++i) {
    hipEventRecord(a);
    hipLaunchKernelGGL((read_kernel<UNROLL, NT, CHUNKED>), dim3(blocks), dim3(256),
                       0, 0, bufs[i % 3], n, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
    sum += ms;
  }
  printf("%-34s blocks %5d  avg %7.1f us %6.2f TB/s   best %7.1f us %6.2f TB/s\n",
         name, blocks, sum / reps * 1e3, n * 16.0 / (sum / reps * 1e-3) / 1e12,
         best * 1e3, n * 16.0 / (best * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = 603ull << 20;  // [502k, 300] fp32
  const size_t n = bytes / 16;
  const v4 *bufs[3];
  for (int i = 0; i < 3; ++i) {
    void *p;
    hipMalloc(&p, bytes);
    hipMemset(p, 0, bytes);
    bufs[i] = (const v4 *)p;
  }
  float *out;
  hipMalloc(&out, 4);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  for (int per_cu : {4, 8, 16}) {
    const int blocks = cus * per_cu;
    run<4, false, false>("stride  plain  x4", bufs, n, out, blocks);
    run<8, false, false>("stride  plain  x8", bufs, n, out, blocks);
    run<4, true, false>("stride  nontemporal x4", bufs, n, out, blocks);
    run<8, true, false>("stride  nontemporal x8", bufs, n, out, blocks);
    run<8, true, true>("chunked nontemporal x8", bufs, n, out, blocks);
    run<16, true, true>("chunked nontemporal x16", bufs, n, out, blocks);
  }
  // copy for comparison
  {
    void *d;
    hipMalloc(&d, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipMemcpyAsync(d, bufs[0], bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i)
      hipMemcpyAsync(d, bufs[i % 3], bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemcpy D2D: %.1f us per copy, %.2f TB/s (read+write)\n",
           ms / 5 * 1e3, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
  }
  return 0;
}
