// Probe kernels for tools/corun_probe.py: what slows a small latency-bound
// kernel down when it shares the chip with the persistent MFMA kernels?
// Each kernel is one dependent chain of `iters` steps of ONE kind; launched
// with a few workgroups it takes iters x (step latency).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int PRIO>
__device__ __forceinline__ void prio() {
  if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
}

// kind 0: VALU chain (no memory, no barrier)
template <int PRIO>
__global__ void probe_alu(float *out, int iters) {
  prio<PRIO>();
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[0] = v;
}
// kind 1: barrier chain
template <int PRIO>
__global__ void probe_barrier(float *out, int iters) {
  prio<PRIO>();
  __shared__ int s[1024];
  int v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    s[threadIdx.x] = v;
    __syncthreads();
    v = s[(threadIdx.x + 65) % blockDim.x] + 1;
    __syncthreads();
  }
  if (v == -1) out[0] = v;
}
// kind 2: dependent global loads (pointer chase through L2)
template <int PRIO>
__global__ void probe_chase(const int *next, float *out, int iters) {
  prio<PRIO>();
  int p = threadIdx.x + blockIdx.x * blockDim.x;
  for (int i = 0; i < iters; ++i) p = next[p];
  if (p == -1) out[0] = p;
}
// kind 3: dependent LDS reads, one wave
template <int PRIO>
__global__ void probe_lds(float *out, int iters) {
  prio<PRIO>();
  __shared__ int s[256];
  s[threadIdx.x] = (threadIdx.x * 7 + 1) & 255;
  __syncthreads();
  int p = threadIdx.x;
  for (int i = 0; i < iters; ++i) p = s[p];
  if (p == -1) out[0] = p;
}

extern "C" int probe_launch(int kind, int prio_on, int blocks, int threads,
                            int iters, const int *next, float *out,
                            void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
#define L(K, ...)                                                           \
  if (prio_on) hipLaunchKernelGGL(K<3>, dim3(blocks), dim3(threads), 0, s,  \
                                  __VA_ARGS__);                             \
  else hipLaunchKernelGGL(K<0>, dim3(blocks), dim3(threads), 0, s, __VA_ARGS__)
  switch (kind) {
    case 0: L(probe_alu, out, iters); break;
    case 1: L(probe_barrier, out, iters); break;
    case 2: L(probe_chase, next, out, iters); break;
    case 3: L(probe_lds, out, iters); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
