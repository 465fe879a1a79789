// Which physical CUs does a CU-masked stream run on?  Every workgroup records
// (XCC id, HW_ID) and spins a little so that the grid spreads over all CUs
// the stream may use.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cu_probe_kernel(uint32_t *out, int spin) {
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
  if (v == 1234.5f) out[0] = 0;
}

extern "C" int cu_probe_launch(uint32_t *out, int blocks, int threads, int spin,
                               void *stream) {
  hipLaunchKernelGGL(cu_probe_kernel, dim3(blocks), dim3(threads), 0,
                     (hipStream_t)stream, out, spin);
  return (int)hipGetLastError();
}

// A stand-in for a persistent kernel: `blocks` workgroups that hold `lds`
// bytes of LDS each and spin for `usec` microseconds.
__global__ void hog_kernel(uint32_t *out, long long ticks) {
  extern __shared__ char hog_lds[];
  const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0 && out) {
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

extern "C" int hog_launch(uint32_t *out, int blocks, int threads, int lds,
                          int usec, void *stream) {
  static int set = 0;
  if (set < lds) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(hog_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    set = lds;
  }
  hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(threads), lds,
                     (hipStream_t)stream, out, (long long)usec * 100);
  return (int)hipGetLastError();
}

__global__ void cu_probe_lds_kernel(uint32_t *out, int spin) {
  extern __shared__ char pad_lds[];
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
  if (v == 1234.5f) out[0] = 0;
}

extern "C" int cu_probe_lds_launch(uint32_t *out, int blocks, int threads,
                                   int spin, int lds, void *stream) {
  static int set = 0;
  if (set < lds) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(cu_probe_lds_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    set = lds;
  }
  hipLaunchKernelGGL(cu_probe_lds_kernel, dim3(blocks), dim3(threads), lds,
                     (hipStream_t)stream, out, spin);
  return (int)hipGetLastError();
}
