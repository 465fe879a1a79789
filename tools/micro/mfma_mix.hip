// What does ONE more instruction cost next to a stream of bf16 MFMAs on gfx950?
// Loop body: one v_mfma (5 independent accumulators in rotation) followed by K
// filler instructions of one kind on independent registers; 1 or 2 waves per
// SIMD; cycles per loop trip from s_memtime (shader clock).  K = 0 is the bare
// MFMA rate.  The edge kernel of csrc/edge_ws_bf16.h wants ~2.4 VALU passes,
// 0.5 ds_read_b128 and 0.13 global loads hidden behind every 16x16x32 MFMA;
// this table says which of them are.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum Op { SUB, MED3, PERM, AND, PKADD, CVTBF, CVTF16, FMAMIX, PKMUL, PKMAXU16,
          DSR128, GLD, DEPSUB, NOPS };
static const char *kOpName[] = {"v_sub_f32", "v_med3_f32", "v_perm_b32", "v_and_b32",
                                "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32",
                                "v_fma_mix_f32", "v_pk_mul_f32", "v_pk_max_u16",
                                "ds_read_b128", "global_load_dwordx4(L1)",
                                "v_sub_f32 (dependent chain)", "s_nop 0"};

template <int OP>
__device__ __forceinline__ void filler(float (&x)[8], v2f (&y)[4], v4u (&z)[2],
                                       const float c, int i, const v4u *lds,
                                       const v4u *glob) {
  const int j = i & 7;
  if constexpr (OP == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  if constexpr (OP == DEPSUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[0]) : "v"(c));
  if constexpr (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(x[j]) : "v"(c));
  if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(c));
  if constexpr (OP == AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[j]));
  if constexpr (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i & 3]) : "v"(y[(i + 1) & 3]));
  if constexpr (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i & 3]) : "v"(y[(i + 1) & 3]));
  if constexpr (OP == CVTBF) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[j]) : "v"(c), "v"(x[(j + 4) & 7]));
  if constexpr (OP == CVTF16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[j]) : "v"(c), "v"(x[(j + 4) & 7]));
  if constexpr (OP == FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[j]) : "v"(c), "v"(c));
  if constexpr (OP == PKMAXU16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x[j]) : "v"(c));
  if constexpr (OP == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(z[i & 1]) : "v"((u32)((threadIdx.x & 63) * 16 + (i & 7) * 1024)));
  if constexpr (OP == GLD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(z[i & 1]) : "v"(glob + (threadIdx.x & 63) + (i & 7) * 64));
  if constexpr (OP == NOPS) asm volatile("s_nop 0");
}

// MF: 0 = 16x16x32 bf16 (16 cycles), 1 = 32x32x16 bf16 (32 cycles),
// 2 = 16x16x16 bf16 (the CDNA3 shape: half the K of MF 0 -- half the time?)
template <int OP, int K, int MF>
__global__ __launch_bounds__(512) void mix(long long *cyc, float *out, int iters,
                                          float seed, const v4u *glob) {
  __shared__ v4u lds[1024];
  v4f acc[5];
  v16f big[2];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
  v4u a = {threadIdx.x, 1u, 2u, 3u}, b = {5u, threadIdx.x, 7u, 8u};
  typedef u32 v2u __attribute__((ext_vector_type(2)));
  v2u a2 = {threadIdx.x, 1u}, b2 = {5u, threadIdx.x};
  float x[8];
  v2f y[4];
  v4u z[2] = {a, b};
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = (v2f){seed + i, seed - i};
  const float c = seed * 0.25f;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = a;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 10; ++m) {
      if constexpr (MF == 0)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % 5]) : "v"(a), "v"(b));
      else if constexpr (MF == 2)
        asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % 5]) : "v"(a2), "v"(b2));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[m % 2]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) filler<OP>(x, y, z, c, m * K + k, lds, glob);
    }
    if constexpr (OP == DSR128 || OP == GLD) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 2; ++i) s += big[i][0] + big[i][15];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += y[i][0] + y[i][1];
  s += __uint_as_float(z[0][0] ^ z[1][3]);
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static long long *g_cyc;
static float *g_out;
static v4u *g_glob;

template <int OP, int K, int MF>
double run(int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
  double best = 1e30;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((mix<OP, K, MF>), dim3(blocks), dim3(threads), 0, 0,
                       g_cyc, g_out, iters, 1.0f, g_glob);
    hipDeviceSynchronize();
    static long long h[256 * 8];
    const int n = blocks * threads / 64;
    hipMemcpy(h, g_cyc, n * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += (double)h[i];
    const double per = sum / n / (iters * 10.0);  // cycles per (MFMA + K fillers) per wave
    if (per < best) best = per;
  }
  return best;
}

template <int OP, int MF>
void row() {
  double r1[5], r2[5];
  r1[0] = run<OP, 0, MF>(1); r2[0] = run<OP, 0, MF>(2);
  r1[1] = run<OP, 1, MF>(1); r2[1] = run<OP, 1, MF>(2);
  r1[2] = run<OP, 2, MF>(1); r2[2] = run<OP, 2, MF>(2);
  r1[3] = run<OP, 3, MF>(1); r2[3] = run<OP, 3, MF>(2);
  r1[4] = run<OP, 4, MF>(1); r2[4] = run<OP, 4, MF>(2);
  printf("%-30s |", kOpName[OP]);
  for (int k = 0; k < 5; ++k) printf(" %6.1f", r1[k]);
  printf(" |");
  // two waves share the SIMD: SIMD cycles per MFMA = per-wave cycles / 2
  for (int k = 0; k < 5; ++k) printf(" %6.1f", r2[k] / 2);
  printf("\n");
  fflush(stdout);
}

template <int MF>
void table() {
  printf("%s: cycles per (MFMA + K fillers); left: 1 wave/SIMD (per wave), right: 2 waves/SIMD (per SIMD = per wave / 2)\n",
         MF == 0 ? "v_mfma_f32_16x16x32_bf16" : MF == 1 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x16_bf16");
  printf("%-30s |    K=0      1      2      3      4 |    K=0      1      2      3      4\n", "filler");
  row<SUB, MF>(); row<DEPSUB, MF>(); row<MED3, MF>(); row<PERM, MF>(); row<AND, MF>();
  row<PKADD, MF>(); row<PKMUL, MF>(); row<CVTBF, MF>(); row<CVTF16, MF>(); row<FMAMIX, MF>();
  row<PKMAXU16, MF>(); row<DSR128, MF>(); row<GLD, MF>(); row<NOPS, MF>();
}

int main() {
  hipMalloc(&g_cyc, 256 * 8 * sizeof(long long));
  hipMalloc(&g_out, 4);
  hipMalloc(&g_glob, 64 * 8 * 16 + 4096);
  hipMemset(g_glob, 0, 64 * 8 * 16 + 4096);
  if (getenv("MFMA_MIX_ONLY16")) {
    printf("v_mfma_f32_16x16x16_bf16 (K = 0 fillers, K = 2 v_sub): 1 wave %.1f %.1f | 2 waves %.1f %.1f\n",
           run<SUB, 0, 2>(1), run<SUB, 2, 2>(1), run<SUB, 0, 2>(2) / 2, run<SUB, 2, 2>(2) / 2);
    printf("v_mfma_f32_16x16x32_bf16 (same):                            1 wave %.1f %.1f | 2 waves %.1f %.1f\n",
           run<SUB, 0, 0>(1), run<SUB, 2, 0>(1), run<SUB, 0, 0>(2) / 2, run<SUB, 2, 0>(2) / 2);
    return 0;
  }
  table<0>();
  table<1>();
  return 0;
}
