// The scalar side of the VQ auto-encoder's training step (train_vqae.py:139-150, vq.py:66-73) as five launches instead of ~40
// device ops on scalars and on the step's smallest tensors (each a graph node of the captured step: ~5 us + ~3.5 us to the next):
//
//   wmz_vq_tail_fwd    vq.py:67   commitment_loss = mse_loss(quantized.detach(), input)
//                      vq.py:70   quantized = input + (quantized - input).detach()          (straight-through estimator)
//                      vq.py:72-73 perplexity = exp(-sum(avg_probs * log(avg_probs + 1e-10)))
//                      -> the straight-through tensor in the decoder's operand dtype (channel-padded for its first conv), the loss
//                      and the perplexity: one element-wise launch with per-workgroup partial sums + one single-workgroup launch
//                      that adds the partials in a fixed order (deterministic) and walks the code counts;
//   wmz_vq_tail_bwd    d input = d quantized (straight-through: identity) + g_loss * 2 / (N E) * (input - quantized), written in
//                      the encoder's activation dtype;
//   wmz_recon_loss_fwd train_vqae.py:264-271 SmoothL1 / MSE / L1 of the reconstruction against the frames, reduction 'mean',
//                      read from the decoder's NHWC (channel-padded) output in place -- no NCHW fp32 copy of the reconstruction;
//   wmz_recon_loss_bwd its gradient, written NHWC in the decoder's activation dtype (zero in the padding channels): the operand of
//                      the last conv's data / weight gradients.
// Sums: fp32, per-workgroup partials in a fixed grid, added by one workgroup in index order -- the same bits on every run.
#include "wmz_common.h"
#include "wmz_internal.h"

namespace {

constexpr int VL_THREADS = 256;
constexpr int VL_MAX_BLOCKS = 1024;

template <typename T> __device__ __forceinline__ float vl_load(const T* p);
template <> __device__ __forceinline__ float vl_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float vl_load<bf16_t>(const bf16_t* p) {
  return bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(p));
}
template <typename T> __device__ __forceinline__ void vl_store(T* p, float v);
template <> __device__ __forceinline__ void vl_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void vl_store<bf16_t>(bf16_t* p, float v) {
  *reinterpret_cast<unsigned short*>(p) = f32_to_bf16_bits(v);
}

// sum over the workgroup (VL_THREADS = 4 waves), result in thread 0
__device__ __forceinline__ float vl_block_sum(float v, float* scratch) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < VL_THREADS / 64; ++w) s += scratch[w];
  }
  return s;
}

// st[n, e] = x + (q - x) (fp32, then the output dtype); st[n, E..Ep) = 0; partial[block] = sum (q - x)^2
template <typename TO>
__global__ __launch_bounds__(VL_THREADS) void vq_tail_fwd_kernel(const float* __restrict__ x, const float* __restrict__ q,
                                                                 TO* __restrict__ st, long N, int E, int Ep,
                                                                 float* __restrict__ partial) {
  __shared__ float scratch[VL_THREADS / 64];
  const long total = N * Ep;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * VL_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * VL_THREADS) {
    const long n = i / Ep;
    const int e = (int)(i - n * Ep);
    float v = 0.f;
    if (e < E) {
      const float xv = x[n * E + e], qv = q[n * E + e];
      const float d = qv - xv;
      acc = fmaf(d, d, acc);
      v = xv + d;                                            // the reference's own two roundings (vq.py:70)
    }
    vl_store<TO>(st + i, v);
  }
  const float s = vl_block_sum(acc, scratch);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one workgroup: out[0] = scale * sum(partial[0..nparts)) in index order; with counts: perplexity[0] = exp(-sum p log(p + 1e-10)),
// p = counts / N
__global__ __launch_bounds__(VL_THREADS) void loss_finalize_kernel(const float* __restrict__ partial, int nparts, float scale,
                                                                   float* __restrict__ out, const float* __restrict__ counts, int C,
                                                                   float inv_n, float* __restrict__ perplexity) {
  __shared__ float scratch[VL_THREADS / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += VL_THREADS) acc += partial[i];
  const float s = vl_block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = s * scale;
  if (counts != nullptr) {
    __syncthreads();
    float h = 0.f;
    for (int c = threadIdx.x; c < C; c += VL_THREADS) {
      const float p = counts[c] * inv_n;
      h += p * logf(p + 1e-10f);
    }
    const float hs = vl_block_sum(h, scratch);
    if (threadIdx.x == 0) perplexity[0] = expf(-hs);
  }
}

// dx[n, e] = dst[n, e] (or 0) + gl * coef * (x - q)
template <typename TI, typename TO>
__global__ __launch_bounds__(VL_THREADS) void vq_tail_bwd_kernel(const TO* __restrict__ dst, const float* __restrict__ x,
                                                                 const float* __restrict__ q, const float* __restrict__ gloss,
                                                                 float coef, TI* __restrict__ dx, long N, int E, int Ep) {
  const float gl = gloss != nullptr ? gloss[0] * coef : 0.f;
  const long total = N * E;
  for (long i = (long)blockIdx.x * VL_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * VL_THREADS) {
    const long n = i / E;
    const int e = (int)(i - n * E);
    float g = dst != nullptr ? vl_load<TO>(dst + n * Ep + e) : 0.f;
    g = fmaf(gl, x[i] - q[i], g);
    vl_store<TI>(dx + i, g);
  }
}

// kind: 0 SmoothL1 (beta 1), 1 MSE, 2 L1
__device__ __forceinline__ float recon_loss_of(float d, int kind) {
  const float a = fabsf(d);
  if (kind == 1) return d * d;
  if (kind == 2) return a;
  return a < 1.f ? 0.5f * d * d : a - 0.5f;
}
__device__ __forceinline__ float recon_dloss_of(float d, int kind) {
  if (kind == 1) return 2.f * d;
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  if (kind == 2) return sgn;
  return fabsf(d) < 1.f ? d : sgn;
}

// y [B, HW, Cp] (NHWC, channel-padded), t [B, C, HW] (NCHW fp32).  BWD: dy = g * scale * loss'(y - t), 0 in the padding channels
template <typename T, bool BWD>
__global__ __launch_bounds__(VL_THREADS) void recon_loss_kernel(const T* __restrict__ y, const float* __restrict__ t, long B, long HW,
                                                                int C, int Cp, int kind, float* __restrict__ partial,
                                                                const float* __restrict__ g, float scale, T* __restrict__ dy) {
  __shared__ float scratch[VL_THREADS / 64];
  const long total = B * HW * Cp;
  float acc = 0.f;
  float gs = 0.f;
  if constexpr (BWD) gs = (g != nullptr ? g[0] : 1.f) * scale;
  for (long i = (long)blockIdx.x * VL_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * VL_THREADS) {
    const long px = i / Cp;                                  // b * HW + p
    const int c = (int)(i - px * Cp);
    float out = 0.f;
    if (c < C) {
      const long b = px / HW, p = px - b * HW;
      const float d = vl_load<T>(y + i) - t[(b * C + c) * HW + p];
      if constexpr (BWD) out = gs * recon_dloss_of(d, kind);
      else acc += recon_loss_of(d, kind);
    }
    if constexpr (BWD) vl_store<T>(dy + i, out);
  }
  if constexpr (!BWD) {
    const float s = vl_block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  }
}

inline int vl_blocks(long total) {
  const long b = (total + VL_THREADS * 4 - 1) / (VL_THREADS * 4);
  return (int)(b < 1 ? 1 : (b > VL_MAX_BLOCKS ? VL_MAX_BLOCKS : b));
}

}  // namespace

// floats of caller workspace the two forward entry points need (the per-workgroup partial sums)
extern "C" long wmz_loss_partials_workspace_floats(void) { return VL_MAX_BLOCKS; }

extern "C" int wmz_vq_tail_fwd(const float* x, const float* q, const float* counts, void* st, float* partial, float* loss,
                               float* perplexity, long N, int E, int Ep, int C, int out_dtype, void* stream) {
  WMZ_REQUIRE(x && q && counts && st && partial && loss && perplexity, "wmz_vq_tail_fwd: null tensor");
  WMZ_REQUIRE(N > 0 && E > 0 && Ep >= E && C > 0, "wmz_vq_tail_fwd: bad shape");
  WMZ_REQUIRE(out_dtype == WMZ_F32 || out_dtype == WMZ_BF16, "wmz_vq_tail_fwd: bad dtype %d", out_dtype);
  hipStream_t s = (hipStream_t)stream;
  const int nb = vl_blocks(N * Ep);
  if (out_dtype == WMZ_F32) hipLaunchKernelGGL(vq_tail_fwd_kernel<float>, dim3(nb), dim3(VL_THREADS), 0, s, x, q, (float*)st, N, E, Ep, partial);
  else hipLaunchKernelGGL(vq_tail_fwd_kernel<bf16_t>, dim3(nb), dim3(VL_THREADS), 0, s, x, q, (bf16_t*)st, N, E, Ep, partial);
  WMZ_LAUNCH_CHECK("wmz_vq_tail_fwd");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(VL_THREADS), 0, s, (const float*)partial, nb, 1.f / ((float)N * (float)E), loss,
                     counts, C, 1.f / (float)N, perplexity);
  WMZ_LAUNCH_CHECK("wmz_vq_tail_fwd");
  return WMZ_OK;
}

extern "C" int wmz_vq_tail_bwd(const void* d_st, const float* x, const float* q, const float* g_loss, void* d_x, long N, int E, int Ep,
                               int in_dtype, int st_dtype, void* stream) {
  WMZ_REQUIRE(x && q && d_x && (d_st || g_loss), "wmz_vq_tail_bwd: null tensor");
  WMZ_REQUIRE(N > 0 && E > 0 && Ep >= E, "wmz_vq_tail_bwd: bad shape");
  WMZ_REQUIRE((in_dtype == WMZ_F32 || in_dtype == WMZ_BF16) && (st_dtype == WMZ_F32 || st_dtype == WMZ_BF16), "wmz_vq_tail_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  const int nb = vl_blocks(N * E);
  const float coef = 2.f / ((float)N * (float)E);
#define VL_BWD(TI, TO) hipLaunchKernelGGL((vq_tail_bwd_kernel<TI, TO>), dim3(nb), dim3(VL_THREADS), 0, s, (const TO*)d_st, x, q, g_loss, coef, (TI*)d_x, N, E, Ep)
  if (in_dtype == WMZ_F32 && st_dtype == WMZ_F32) VL_BWD(float, float);
  else if (in_dtype == WMZ_F32) VL_BWD(float, bf16_t);
  else if (st_dtype == WMZ_F32) VL_BWD(bf16_t, float);
  else VL_BWD(bf16_t, bf16_t);
#undef VL_BWD
  WMZ_LAUNCH_CHECK("wmz_vq_tail_bwd");
  return WMZ_OK;
}

extern "C" int wmz_recon_loss_fwd(const void* y, const float* target, float* partial, float* loss, long B, long HW, int C, int Cp,
                                  int kind, int dtype, void* stream) {
  WMZ_REQUIRE(y && target && partial && loss, "wmz_recon_loss_fwd: null tensor");
  WMZ_REQUIRE(B > 0 && HW > 0 && C > 0 && Cp >= C && kind >= 0 && kind <= 2, "wmz_recon_loss_fwd: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_recon_loss_fwd: bad dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const int nb = vl_blocks(B * HW * Cp);
  if (dtype == WMZ_F32)
    hipLaunchKernelGGL((recon_loss_kernel<float, false>), dim3(nb), dim3(VL_THREADS), 0, s, (const float*)y, target, B, HW, C, Cp, kind, partial,
                       (const float*)nullptr, 0.f, (float*)nullptr);
  else
    hipLaunchKernelGGL((recon_loss_kernel<bf16_t, false>), dim3(nb), dim3(VL_THREADS), 0, s, (const bf16_t*)y, target, B, HW, C, Cp, kind, partial,
                       (const float*)nullptr, 0.f, (bf16_t*)nullptr);
  WMZ_LAUNCH_CHECK("wmz_recon_loss_fwd");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(VL_THREADS), 0, s, (const float*)partial, nb,
                     1.f / ((float)B * (float)HW * (float)C), loss, (const float*)nullptr, 0, 0.f, (float*)nullptr);
  WMZ_LAUNCH_CHECK("wmz_recon_loss_fwd");
  return WMZ_OK;
}

extern "C" int wmz_recon_loss_bwd(const void* y, const float* target, const float* g_loss, void* d_y, long B, long HW, int C, int Cp,
                                  int kind, int dtype, void* stream) {
  WMZ_REQUIRE(y && target && d_y, "wmz_recon_loss_bwd: null tensor");
  WMZ_REQUIRE(B > 0 && HW > 0 && C > 0 && Cp >= C && kind >= 0 && kind <= 2, "wmz_recon_loss_bwd: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_recon_loss_bwd: bad dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const int nb = vl_blocks(B * HW * Cp);
  const float scale = 1.f / ((float)B * (float)HW * (float)C);
  if (dtype == WMZ_F32)
    hipLaunchKernelGGL((recon_loss_kernel<float, true>), dim3(nb), dim3(VL_THREADS), 0, s, (const float*)y, target, B, HW, C, Cp, kind,
                       (float*)nullptr, g_loss, scale, (float*)d_y);
  else
    hipLaunchKernelGGL((recon_loss_kernel<bf16_t, true>), dim3(nb), dim3(VL_THREADS), 0, s, (const bf16_t*)y, target, B, HW, C, Cp, kind,
                       (float*)nullptr, g_loss, scale, (bf16_t*)d_y);
  WMZ_LAUNCH_CHECK("wmz_recon_loss_bwd");
  return WMZ_OK;
}
