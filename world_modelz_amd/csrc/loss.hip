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

// One step of the sampler loop between two forward passes (main.py:76-104: top-k filter -> softmax -> multinomial -> re-mask):
// one wave per row of logits, everything in registers, both uniforms from one Philox call.  Lane l holds the NV consecutive
// classes l * NV .. (classes past C are -inf).
//   * top-k: the k-th largest logit by a 32-step bitwise search on order-preserving integer keys (count of keys >= candidate
//     by wave ballots); logits below it are dropped, ties with it kept (the reference's `logits < v[:, -1]`);
//   * softmax weights w = exp(l - max), total by wave reduction;
//   * the draw: the number of classes whose cumulative weight is <= u * total (inverse CDF, the definition of
//     sample.categorical_from_uniform), clamped to C - 1;
//   * re-mask: with a second uniform u2 > alpha (and, with `last_mask`, only where the previous iteration masked:
//     consistent masking) the position gets the mask token instead of the draw.
// alpha = alphas[*counter % n_alpha] and the Philox stream id = *counter live in device memory: the launch sits in a hipGraph.
template <int NV>
__global__ __launch_bounds__(256) void sample_tokens_kernel(const float* __restrict__ logits, long ld, int R, int C, int top_k,
                                                            const float* __restrict__ alphas, int n_alpha, int64_t mask_token,
                                                            int64_t* __restrict__ out_tokens, long rows_per_block,
                                                            long block_stride, int64_t* __restrict__ denoised,
                                                            unsigned char* __restrict__ last_mask, unsigned long long seed,
                                                            const long long* __restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const long ctr = *counter;
  const float alpha = alphas[(int)(ctr % n_alpha)];
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += (long)gridDim.x * 4) {
    const float* x = logits + row * ld + lane * NV;
    float v[NV];
#pragma unroll
    for (int e = 0; e < NV; e += 4) {
      if (lane * NV + e + 3 < C) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(x + e);
        v[e] = q[0]; v[e + 1] = q[1]; v[e + 2] = q[2]; v[e + 3] = q[3];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e + i] = lane * NV + e + i < C ? x[e + i] : -INFINITY;
      }
    }
    if (top_k > 0 && top_k < C) {
      // order-preserving keys: flip all bits of negatives, the sign bit of non-negatives
      unsigned key[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const unsigned b = __float_as_uint(v[e]);
        key[e] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      }
      unsigned T = 0;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < NV; ++e) cnt += __popcll(__ballot(key[e] >= cand));
        if (cnt >= top_k) T = cand;
      }
#pragma unroll
      for (int e = 0; e < NV; ++e) if (key[e] < T) v[e] = -INFINITY;
    }
    float m = v[0];
#pragma unroll
    for (int e = 1; e < NV; ++e) m = fmaxf(m, v[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float w[NV], part = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) { w[e] = __expf(v[e] - m); part += w[e]; w[e] = part; }    // lane-local running sums
    float inc = part;                                          // inclusive scan of the lane totals, in lane order
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    const float total = __shfl(inc, 63);
    const float base = inc - part;
    float u[4];
    philox4((unsigned long long)row, seed, (unsigned long long)ctr, u);
    const float xq = u[0] * total;
    int below = 0;
#pragma unroll
    for (int e = 0; e < NV; ++e) below += (lane * NV + e < C && base + w[e] <= xq) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    const int draw = below < C ? below : C - 1;
    if (lane == 0) {
      bool mask = u[1] > alpha;
      if (last_mask != nullptr) { mask = mask && last_mask[row] != 0; last_mask[row] = mask ? 1 : 0; }
      denoised[row] = draw;
      const long blk = row / rows_per_block;
      out_tokens[blk * block_stride + (row - blk * rows_per_block)] = mask ? mask_token : (int64_t)draw;
    }
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

template <typename T> __device__ __forceinline__ void ce_store4(T* p, const float (&f)[4]);
template <> __device__ __forceinline__ void ce_store4<float>(float* p, const float (&f)[4]) {
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = f[e];
  *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void ce_store4<bf16_t>(bf16_t* p, const float (&f)[4]) {
  s16x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (short)f32_to_bf16_bits(f[e]);
  *reinterpret_cast<s16x4*>(p) = v;
}

// Both of the above in ONE pass: a wave holds its row in registers (NCH float4 per lane, C <= 256 NCH), so the [R, C] logits
// are read once instead of three times (max / sum-exp / gradient) and every access is 16 bytes (8 for the 16-bit gradient).
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ce_fused_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ target,
                                                       float* __restrict__ loss, float* __restrict__ lse,
                                                       const float* __restrict__ grow, T* __restrict__ dlogits, long R, int C) {
  const int lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += (long)gridDim.x * 4) {
    const float* x = logits + row * ld;
    f32x4 v[NCH];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      if (c + 4 <= C) {
        v[q] = *reinterpret_cast<const f32x4*>(x + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][e] = c + e < C ? x[c + e] : -INFINITY;
      }
      m = fmaxf(fmaxf(m, fmaxf(v[q][0], v[q][1])), fmaxf(v[q][2], v[q][3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __expf(v[q][e] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float l = m + logf(s);
    long t = target[row];
    const int tc = (int)(t < 0 ? 0 : (t >= C ? C - 1 : t));
    if (lane == 0) {
      lse[row] = l;
      loss[row] = l - x[tc];
    }
    const float g = grow[row];
    const int tt = (int)t;                      // (an out-of-range target matches no column, as in ce_bwd_kernel)
    T* d = dlogits + row * (long)C;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = q * 256 + lane * 4;
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = (__expf(v[q][e] - l) - (c + e == tt ? 1.f : 0.f)) * g;
      if (c + 4 <= C && (C & 3) == 0) {
        ce_store4<T>(d + c, o4);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < C) d[c + e] = Elem<T>::from_f32(o4[e]);
      }
    }
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

extern "C" int wmz_sample_tokens_dev(const float* logits, long ld, int R, int C, int top_k, const float* alphas, int n_alpha,
                                     int64_t mask_token, int64_t* out_tokens, long rows_per_block, long block_stride,
                                     int64_t* denoised, unsigned char* last_mask, unsigned long long seed,
                                     const long long* counter, void* stream) {
  WMZ_REQUIRE(logits && alphas && out_tokens && denoised && counter && R > 0 && C > 0 && n_alpha > 0 && rows_per_block > 0,
              "wmz_sample_tokens_dev: bad arguments");
  WMZ_REQUIRE(ld >= C && ld % 4 == 0 && (((uintptr_t)logits) & 15) == 0, "wmz_sample_tokens_dev: logits rows must be 16-byte aligned");
  if (C > 2048) {
    wmz_set_error("wmz_sample_tokens_dev: built for <= 2048 classes (got %d)", C);
    return WMZ_ERR_UNSUPPORTED;
  }
  const int grid = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
  hipStream_t st = (hipStream_t)stream;
#define WMZ_SMP(NV) hipLaunchKernelGGL(sample_tokens_kernel<NV>, dim3(grid), dim3(256), 0, st, logits, ld, R, C, top_k, alphas, n_alpha, \
                                       mask_token, out_tokens, rows_per_block, block_stride, denoised, last_mask, seed, counter)
  if (C <= 256) WMZ_SMP(4); else if (C <= 512) WMZ_SMP(8); else if (C <= 1024) WMZ_SMP(16); else WMZ_SMP(32);
#undef WMZ_SMP
  WMZ_LAUNCH_CHECK("wmz_sample_tokens_dev");
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

extern "C" int wmz_ce_fwd_bwd(const float* logits, long ld, const int64_t* target, float* loss, float* lse, const float* grad_rows,
                              void* dlogits, long R, int C, int dtype, void* stream) {
  WMZ_REQUIRE(logits && target && loss && lse && grad_rows && dlogits && R > 0 && C > 0, "wmz_ce_fwd_bwd: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_ce_fwd_bwd: bad dtype %d", dtype);
  if (C > 8192 || (ld & 3) != 0) {               // rows that do not fit a wave's registers / unaligned rows: the two passes
    const int rc = wmz_ce_fwd(logits, ld, target, loss, lse, R, C, stream);
    return rc != WMZ_OK ? rc : wmz_ce_bwd(logits, ld, target, lse, grad_rows, dlogits, R, C, dtype, stream);
  }
  const int grid = (int)((R + 3) / 4 < 4096 ? (R + 3) / 4 : 4096);
  hipStream_t st = (hipStream_t)stream;
#define WMZ_CEF(T, NCH) hipLaunchKernelGGL((ce_fused_kernel<T, NCH>), dim3(grid), dim3(256), 0, st, logits, ld, target, loss, lse, \
                                           grad_rows, (T*)dlogits, R, C)
#define WMZ_CEF_T(T) do { if (C <= 1024) WMZ_CEF(T, 4); else if (C <= 2048) WMZ_CEF(T, 8); else if (C <= 4096) WMZ_CEF(T, 16); \
                          else WMZ_CEF(T, 32); } while (0)
  if (dtype == WMZ_BF16) WMZ_CEF_T(bf16_t); else WMZ_CEF_T(float);
#undef WMZ_CEF_T
#undef WMZ_CEF
  WMZ_LAUNCH_CHECK("wmz_ce_fwd_bwd");
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
