// Cycles per MFMA on gfx950 for the shapes the attention kernels choose between: 16x16x32 bf16 (the K = 32 form), the legacy
// 16x16x16 bf16_1k (K = 16: what a ONE-visitor-row accumulation step would use) and 32x32x16 bf16.  One wave per SIMD, four
// independent accumulators, operands in registers; cycles from s_memtime around the loop (median over workgroups).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int ITERS = 2048;

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(const short* in, float* out, long long* ticks) {
  const int lane = threadIdx.x;
  s16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = in[(lane * 8 + i) & 1023]; b[i] = in[(lane * 8 + i + 512) & 1023]; }
  s16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f32x4 c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
  f32x16 d0 = 0.f, d1 = 0.f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    } else if constexpr (KIND == 1) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c3, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
  out[blockIdx.x * 256 + lane] = s;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, const short* in, float* out, long long* ticks) {
  const int nb = 256;
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(nb), dim3(256), 0, 0, in, out, ticks);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(nb), dim3(256), 0, 0, in, out, ticks);
  hipDeviceSynchronize();
  std::vector<long long> h(nb);
  hipMemcpy(h.data(), ticks, nb * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-28s %7.2f cycles per MFMA (median of %d workgroups, one wave per SIMD, %d MFMAs)\n", name,
         (double)h[nb / 2] / (4.0 * ITERS), nb, 4 * ITERS);
}

int main() {
  short* in; float* out; long long* ticks;
  hipMalloc(&in, 1024 * sizeof(short));
  hipMalloc(&out, 256 * 256 * sizeof(float));
  hipMalloc(&ticks, 256 * sizeof(long long));
  std::vector<short> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (short)(0x3f80 + (rand() & 0x7f));     // bf16 values in [1, 2)
  hipMemcpy(in, h.data(), 1024 * sizeof(short), hipMemcpyHostToDevice);
  run<0>("v_mfma_f32_16x16x32_bf16", in, out, ticks);
  run<1>("v_mfma_f32_16x16x16_bf16", in, out, ticks);
  run<2>("v_mfma_f32_32x32x16_bf16", in, out, ticks);
  return 0;
}
