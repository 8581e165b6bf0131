// The two steps either side of the denoiser in the training loop:
//   wmz_corrupt_tokens  main.py:246-259  mask + uniform token corruption of the last latent frame, without the [B,HW,C]
//                                        one-hot / lerp / multinomial temporaries (closed form, counter-based RNG in-kernel)
//   wmz_ce_fwd / _bwd   main.py:266-274  CrossEntropyLoss(reduction='none') over the last-frame logits and its gradient,
//                                        written directly in the GEMM operand dtype
#include "wmz_common.h"

namespace {

// Philox4x32-10 (Salmon et al. 2011): counter = (index, stream), key = seed
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(unsigned long long idx, unsigned long long seed, unsigned long long stream, float (&u)[4]) {
  unsigned c[4] = {(unsigned)idx, (unsigned)(idx >> 32), (unsigned)stream, (unsigned)(stream >> 32)};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
}

// one thread per last-frame position.  a = 0.1 r: with probability a redraw uniformly over the C codes, else keep the
// token (== multinomial(lerp(one_hot, 1/C, a))); then positions with u < r become the mask token C.
__global__ __launch_bounds__(256) void corrupt_kernel(const int64_t* __restrict__ z_last, long clip_stride,
                                                      const float* __restrict__ r, int64_t* __restrict__ out,
                                                      long out_stride, int64_t* __restrict__ target, int B, int HW, int C,
                                                      unsigned long long seed, unsigned long long stream,
                                                      const unsigned long long* __restrict__ counter) {
  if (counter != nullptr) stream |= *counter & ((1ull << 40) - 1);      // per-call stream id kept in device memory (hipGraph replay)
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (long)b * HW);
    const int64_t tok = z_last[b * clip_stride + p];
    float u[4];
    philox4((unsigned long long)i, seed, stream, u);
    const float rb = r[b];
    int64_t d = tok;
    if (u[0] < rb * 0.1f) { int k = (int)(u[1] * (float)C); d = k < C ? k : C - 1; }
    if (u[2] < rb) d = C;
    if (target) target[i] = tok;
    out[b * out_stride + p] = d;
  }
}

// one wave per row; C <= 64 * 4 * KV handled by a strided loop
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ target,
                                                     float* __restrict__ loss, float* __restrict__ lse, long R, int C) {
  const int lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += (long)gridDim.x * 4) {
    const float* x = logits + row * ld;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      const float l = m + logf(s);
      long t = target[row];
      t = t < 0 ? 0 : (t >= C ? C - 1 : t);
      lse[row] = l;
      loss[row] = l - x[t];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ grow,
                                                     T* __restrict__ dlogits, long R, int C) {
  const long total = R * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float p = __expf(logits[row * ld + c] - lse[row]);
    const float v = (p - (target[row] == c ? 1.f : 0.f)) * grow[row];
    dlogits[i] = Elem<T>::from_f32(v);
  }
}

}  // namespace

extern "C" int wmz_corrupt_tokens(const int64_t* z_last, long clip_stride, const float* r, int64_t* out, long out_stride,
                                  int64_t* target, int B, int HW, int C, unsigned long long seed, unsigned long long stream_id,
                                  void* stream) {
  WMZ_REQUIRE(z_last && r && out && B > 0 && HW > 0 && C > 0, "wmz_corrupt_tokens: bad arguments");
  const long total = (long)B * HW;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(corrupt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z_last, clip_stride, r, out, out_stride,
                     target, B, HW, C, seed, stream_id, (const unsigned long long*)nullptr);
  WMZ_LAUNCH_CHECK("wmz_corrupt_tokens");
  return WMZ_OK;
}

// The same with the low 40 bits of the Philox stream id read from device memory at run time (`counter`, which the caller
// advances between launches): the launch can sit in a hipGraph and still draw a fresh mask on every replay.
extern "C" int wmz_corrupt_tokens_dev(const int64_t* z_last, long clip_stride, const float* r, int64_t* out, long out_stride,
                                      int64_t* target, int B, int HW, int C, unsigned long long seed,
                                      unsigned long long stream_hi, const unsigned long long* counter, void* stream) {
  WMZ_REQUIRE(z_last && r && out && counter && B > 0 && HW > 0 && C > 0, "wmz_corrupt_tokens_dev: bad arguments");
  const long total = (long)B * HW;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(corrupt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z_last, clip_stride, r, out, out_stride,
                     target, B, HW, C, seed, stream_hi & ~((1ull << 40) - 1), counter);
  WMZ_LAUNCH_CHECK("wmz_corrupt_tokens_dev");
  return WMZ_OK;
}

extern "C" int wmz_ce_fwd(const float* logits, long ld, const int64_t* target, float* loss, float* lse, long R, int C,
                          void* stream) {
  WMZ_REQUIRE(logits && target && loss && lse && R > 0 && C > 0, "wmz_ce_fwd: bad arguments");
  const int grid = (int)((R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ld, target, loss, lse, R, C);
  WMZ_LAUNCH_CHECK("wmz_ce_fwd");
  return WMZ_OK;
}

extern "C" int wmz_ce_bwd(const float* logits, long ld, const int64_t* target, const float* lse, const float* grad_rows,
                          void* dlogits, long R, int C, int dtype, void* stream) {
  WMZ_REQUIRE(logits && target && lse && grad_rows && dlogits && R > 0 && C > 0, "wmz_ce_bwd: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_ce_bwd: bad dtype %d", dtype);
  const long total = R * C;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16) hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, logits, ld, target, lse, grad_rows, (bf16_t*)dlogits, R, C);
  else hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, logits, ld, target, lse, grad_rows, (float*)dlogits, R, C);
  WMZ_LAUNCH_CHECK("wmz_ce_bwd");
  return WMZ_OK;
}
