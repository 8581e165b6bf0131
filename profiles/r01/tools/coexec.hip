// Micro-benchmark: do VALU instructions of a wave overlap its own / its SIMD partner's MFMAs?  (kernel development aid)
// build: hipcc --offload-arch=gfx950 -O3 tools/coexec.hip -o tools/coexec ; run on the GPU box: ./tools/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int NM, int NF, int NE, int DEP>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
  s16x8 a = (s16x8)(0x3f80), b = (s16x8)(0x3f80);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[DEP ? 0 : (m & 3)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[DEP ? 0 : (m & 3)], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) v[f & 7] = __builtin_fmaf(v[f & 7], 1.0001f, 0.5f);
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e & 7] = __builtin_amdgcn_exp2f(v[e & 7] * 0.001f);
    asm volatile("" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// same work, but the VALU instructions are placed BETWEEN the MFMAs (PER valu + PERE transcendental per MFMA gap)
template <int PER, int PERE>
__global__ __launch_bounds__(512, 1) void ki(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
  s16x8 a = (s16x8)(0x3f80), b = (s16x8)(0x3f80);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < 4 * PER; ++f) v[f & 7] = __builtin_fmaf(v[f & 7], 1.0001f, 0.5f);
#pragma unroll
    for (int e = 0; e < 4 * PERE; ++e) v[e & 7] = __builtin_amdgcn_exp2f(v[e & 7]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, PER + PERE, 0);   // then PER + PERE VALU
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PER, int PERE>
void runi(float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((ki<PER, PERE>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("interleaved: 4 x (mfma + %d fma + %d exp)          : %7.1f ticks / iteration\n", PER, PERE, (double)c / iters);
}

template <int NM, int NF, int NE, int DEP>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NM, NF, NE, DEP>), dim3(1), dim3(512), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s mfma %2d fma %3d exp %2d dep %d : %7.1f cycles / iteration (2 waves per SIMD)\n", name, NM, NF, NE, DEP, (double)c / iters);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 4 * 4); hipMalloc(&cyc, 64);
  run<4, 0, 0, 0>("mfma only (4 accs)", out, cyc);
  run<4, 0, 0, 1>("mfma only (1 acc chain)", out, cyc);
  run<0, 32, 0, 0>("fma only", out, cyc);
  run<0, 0, 8, 0>("exp only", out, cyc);
  run<4, 16, 0, 0>("mfma + 16 fma", out, cyc);
  run<4, 32, 0, 0>("mfma + 32 fma", out, cyc);
  run<4, 48, 0, 0>("mfma + 48 fma", out, cyc);
  run<4, 0, 8, 0>("mfma + 8 exp", out, cyc);
  run<4, 24, 8, 0>("mfma + 24 fma + 8 exp", out, cyc);
  run<4, 24, 8, 1>("mfma(chain) + 24 fma + 8 exp", out, cyc);
  runi<2, 0>(out, cyc); runi<4, 0>(out, cyc); runi<6, 0>(out, cyc); runi<8, 0>(out, cyc); runi<12, 0>(out, cyc);
  runi<4, 1>(out, cyc); runi<4, 2>(out, cyc); runi<6, 2>(out, cyc); runi<0, 2>(out, cyc);
  return 0;
}
