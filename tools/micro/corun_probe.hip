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

// kind 4: a stand-in for a small-footprint K-row MLP kernel: 256 threads,
// 24 KB of LDS, per step one ds_read_b128 + five fp32 MFMAs on a dependent
// accumulator chain per wave (the shape of mlp_engine.h's layer pass with five
// column tiles per wave), weights from L2.
typedef float pv4 __attribute__((ext_vector_type(4)));
template <int PRIO>
__global__ __launch_bounds__(256) void probe_mlp(const float *w, float *out,
                                                 int iters) {
  prio<PRIO>();
  __shared__ float tile[16 * 308 + 64];
  for (int i = threadIdx.x; i < 16 * 308; i += 256) tile[i] = 1e-3f * (i & 63);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  pv4 acc[5];
  for (int t = 0; t < 5; ++t) acc[t] = (pv4){0.f, 0.f, 0.f, 0.f};
  const pv4 *wp = reinterpret_cast<const pv4 *>(w) + lane;
  for (int i = 0; i < iters; ++i) {
    const int q = i % 19;
    const pv4 a = *reinterpret_cast<const pv4 *>(&tile[(lane & 15) * 308 +
                                                        16 * q + 4 * (lane >> 4)]);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const pv4 b = wp[(q * 19 + t) * 64];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[t], 0, 0, 0);
    }
  }
  float v = 0.f;
  for (int t = 0; t < 5; ++t) v += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (v == 12345.f) out[0] = v;
}

extern "C" int probe_mlp_launch(int prio_on, int blocks, int iters,
                                const float *w, float *out, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (prio_on)
    hipLaunchKernelGGL(probe_mlp<3>, dim3(blocks), dim3(256), 0, s, w, out, iters);
  else
    hipLaunchKernelGGL(probe_mlp<0>, dim3(blocks), dim3(256), 0, s, w, out, iters);
  return (int)hipGetLastError();
}
