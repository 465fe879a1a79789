// Calibration: what does a loop of nothing but v_mfma_f32_16x16x4_f32 reach on
// this chip?  (157.3 TFLOP/s is the datasheet figure at 2.4 GHz; under a
// sustained matrix load the clock drops.)  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed) {
  v4f acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int blocks, int iters, const char *what) {
  float *out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 4 * NACC * 2048.0;
    printf("%s rep %d: %.3f ms  %.1f TFLOP/s\n", what, rep, ms, flops / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  // ~1 ms and ~10 ms bursts: short bursts see the boost clock, long ones the
  // sustained clock
  run<20>(256, 2000, "1 wave/SIMD, 20 acc, short");
  run<20>(512, 1000, "2 waves/SIMD, 20 acc, short");
  run<20>(512, 20000, "2 waves/SIMD, 20 acc, long");
  run<4>(512, 100000, "2 waves/SIMD, 4 acc (dependent chains), long");
  return 0;
}
