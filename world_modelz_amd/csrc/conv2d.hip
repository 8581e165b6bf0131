// Conv encoder / decoder tiles of the VQ auto-encoder (vq-video-diffusion/autoencoder.py): NHWC implicit-GEMM
// convolution on MFMA plus the BatchNorm / LeakyReLU / bilinear pieces around it.
//
//   conv:  out[b,ho,wo,co] = sum_{kh,kw,ci} x[b, ho*s+kh-p, wo*s+kw-p, ci] * w[co,kh,kw,ci]   (+bias)
//          as C[M = B*Ho*Wo, N = Cout] = A_im2col[M, K = kh*kw*Cin] . Wt[N, K]^T; the im2col row is never built:
//          the A-slab fetch computes the source address per 16-byte chunk (Cin % 8 == 0 keeps a chunk in one tap).
//          Epilogue: bias, per-channel affine (folded eval-mode BatchNorm), residual add, LeakyReLU, and optionally
//          per-channel sum / sum-of-squares of the stored output (training-mode BatchNorm statistics).
//   channel_stats / bn_finalize / affine_act / bilinear2x: the memory-bound companions.
#include "wmz_common.h"
#include "bn_lazy.h"
#ifndef WMZ_ABL_NOSTAT
#define WMZ_ABL_NOSTAT 0      // timing ablation (tools/build_variant.py): drop the global statistics atomics
#endif

namespace {

constexpr int ROWB = 128, CPR = 8, NT = 256;      // (tile sizes: template parameters of conv2d_kernel)

struct ConvParams {
  const void* x; const void* w; void* out;
  const float* bias; const float* scale; const float* shift;
  const void* res;
  float* stat_sum; float* stat_sq;
  int B, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int M, K, nbn;
  float slope; int leaky;
  // input prologue (1x1 convs): a' = LeakyReLU(a * in_scale[c] + in_shift[c]) applied while the A slab is staged -- the
  // training-mode BatchNorm + activation in FRONT of the conv, without a pass over the tensor of its own
  const float* in_scale; const float* in_shift; float in_slope;
};

__device__ __forceinline__ int swz128(int r) { return ((r >> 1) << 4) & 112; }

template <typename T> __device__ __forceinline__ void chunk_to_f32(const i32x4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const i32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __int_as_float(c[i]);
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const i32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(((unsigned)c[i]) << 16);
    f[2 * i + 1] = __uint_as_float(((unsigned)c[i]) & 0xFFFF0000u);
  }
}
template <typename T> __device__ __forceinline__ i32x4 f32_to_chunk(const float* f);
template <> __device__ __forceinline__ i32x4 f32_to_chunk<float>(const float* f) {
  i32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = __float_as_int(f[i]);
  return c;
}
template <> __device__ __forceinline__ i32x4 f32_to_chunk<bf16_t>(const float* f) {
  i32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = (int)((unsigned)f32_to_bf16_bits(f[2 * i]) | ((unsigned)f32_to_bf16_bits(f[2 * i + 1]) << 16));
  return c;
}

// TALL = false: output tile 128 x 128, the four waves as 2 x 2.  TALL = true (Cout <= 64: conv_1, the 1x1 and the 2x2
// down-sampling convolutions): output tile 256 x 64, the waves stacked -- no MFMA or weight-slab traffic spent on columns
// that do not exist, and twice the rows per workgroup behind each (short: K = 72 .. 256) reduction and epilogue.
template <typename T, bool TALL>
__global__ __launch_bounds__(NT, 2) void conv2d_kernel(ConvParams P) {
  constexpr int BM = TALL ? 256 : 128, BN = TALL ? 64 : 128;
  constexpr int AI = BM / 32, BI = BN / 32;             // A / B rows staged per thread and slab
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = CPR * EPC;
  constexpr int KSTEPS = BK / 16;
  __shared__ __attribute__((aligned(16))) char tiles[(BM + BN) * ROWB];       // A slab | B slab; reused by the epilogue
  __shared__ float stat_l[2][BN];
  char* As = tiles;
  char* Bs = tiles + BM * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lid / P.nbn, bn = lid - bm * P.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const T* X = reinterpret_cast<const T*>(P.x);
  const T* Wt = reinterpret_cast<const T*>(P.w);
  const int K = P.K;

  // the AI output pixels this thread stages (rows rr + 32 i): decompose once
  const int cc = tid & 7, rr = tid >> 3;
  int pb[AI], ph[AI], pw[AI];
  bool pok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + rr + 32 * i;
    pok[i] = m < P.M;
    const int mm = pok[i] ? m : 0;
    const int wo = mm % P.Wo, t = mm / P.Wo;
    pw[i] = wo * P.stride - P.pad;
    ph[i] = (t % P.Ho) * P.stride - P.pad;
    pb[i] = t / P.Ho;
  }

  i32x4 ra[AI], rb[BI];
  auto pre = [&](i32x4 v, int k) {          // k = first input channel of the chunk (1x1 conv: k IS the channel)
    float f[EPC];
    chunk_to_f32<T>(v, f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float y = fmaf(f[e], P.in_scale[k + e], P.in_shift[k + e]);
      f[e] = y > 0.f ? y : y * P.in_slope;
    }
    return f32_to_chunk<T>(f);
  };
  auto fetch = [&](int k0) {
    const int k = k0 + cc * EPC;
    const int tap = k / P.Cin, ci = k - tap * P.Cin;
    const int kh = tap / P.KW, kw = tap - kh * P.KW;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      ra[i] = (i32x4)(0);
      if (k < K) {
        const int hi = ph[i] + kh, wi = pw[i] + kw;
        if (pok[i] && hi >= 0 && hi < P.Hi && wi >= 0 && wi < P.Wi)
          ra[i] = *reinterpret_cast<const i32x4*>(X + (((long)pb[i] * P.Hi + hi) * P.Wi + wi) * P.Cin + ci);
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      rb[i] = (i32x4)(0);
      const int r = rr + 32 * i;
      if (k < K && n0 + r < P.Cout) rb[i] = *reinterpret_cast<const i32x4*>(Wt + (long)(n0 + r) * K + k);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
  const int wr = TALL ? wave * 64 : (wave >> 1) * 64, wc = TALL ? 0 : (wave & 1) * 64;
  const int l31 = lane & 31, hh = lane >> 5;

  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int r = rr + 32 * i;
      const int off = r * ROWB + ((cc << 4) ^ swz128(r));
      i32x4 av = ra[i];
      if (P.in_scale != nullptr && k0 + cc * EPC < K && pok[i]) av = pre(av, k0 + cc * EPC);   // (rows past M stay zero)
      *reinterpret_cast<i32x4*>(As + off) = av;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int r = rr + 32 * i;
      *reinterpret_cast<i32x4*>(Bs + r * ROWB + ((cc << 4) ^ swz128(r))) = rb[i];
    }
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      Frag8<T> af[2], bf[2];
      const int b0 = (kk * 16 + hh * 8) * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wr + 32 * i + l31, rn = wc + 32 * i + l31;
        const char* rowa = As + r * ROWB;
        const char* rowb = Bs + rn * ROWB;
        const int sa = swz128(r), sb = swz128(rn);
        if constexpr (sizeof(T) == 2) {
          af[i].v = *reinterpret_cast<const s16x8*>(rowa + (b0 ^ sa));
          bf[i].v = *reinterpret_cast<const s16x8*>(rowb + (b0 ^ sb));
        } else {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(rowa + (b0 ^ sa));
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(rowa + ((b0 + 16) ^ sa));
          const f32x4 y0 = *reinterpret_cast<const f32x4*>(rowb + (b0 ^ sb));
          const f32x4 y1 = *reinterpret_cast<const f32x4*>(rowb + ((b0 + 16) ^ sb));
#pragma unroll
          for (int e = 0; e < 4; ++e) { af[i].v[e] = x0[e]; af[i].v[4 + e] = x1[e]; bf[i].v[e] = y0[e]; bf[i].v[4 + e] = y1[e]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], af[i], bf[j]);
    }
  }

  const T* R = reinterpret_cast<const T*>(P.res);
  T* O = reinterpret_cast<T*>(P.out);
  if constexpr (sizeof(T) == 2) {
    if ((P.Cout & 7) == 0) {
      // 16-bit outputs: a lane owns ONE column of the accumulator tile, so direct stores (and residual loads) would be 2
      // bytes each -- 64 of them per lane.  Stage the affine result in fp32 through LDS (64 rows per round, in the space the
      // slabs occupied) and finish on whole 16-byte row chunks: residual add, LeakyReLU, the single rounding, the store,
      // and the per-channel statistics of what was stored (LDS partial sums, one global atomic per channel and workgroup).
      float* stage = reinterpret_cast<float*>(tiles);                 // [64][BN] fp32
      if (tid < BN) { stat_l[0][tid] = 0.f; stat_l[1][tid] = 0.f; }
#pragma unroll 1
      for (int round = 0; round < BM / 64; ++round) {
        __syncthreads();
        if ((TALL ? wave : (wave >> 1)) == round) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int cl = wc + 32 * j + l31;
              const int col = n0 + cl;
              const bool cok = col < P.Cout;
              const float bv = (cok && P.bias) ? P.bias[col] : 0.f;
              const float sc = (cok && P.scale) ? P.scale[col] : 1.f;
              const float sh = (cok && P.shift) ? P.shift[col] : 0.f;
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                const int rl = 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                stage[rl * BN + cl] = (acc[i][j][reg] + bv) * sc + sh;
              }
            }
        }
        __syncthreads();
        // 64 rows x BN / 8 chunks of 8 columns, 4 (2 when TALL) per thread; a thread keeps its chunk column
        constexpr int CH = BN / 8, RPI = NT / CH;            // chunk columns; rows covered per iteration
        const int ch = tid & (CH - 1), col = n0 + ch * 8;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        if (col < P.Cout) {
#pragma unroll
          for (int it = 0; it < 64 / RPI; ++it) {
            const int rl = tid / CH + RPI * it;
            const int row = m0 + round * 64 + rl;
            if (row >= P.M) continue;
            const f32x4 a = *reinterpret_cast<const f32x4*>(stage + rl * BN + ch * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(stage + rl * BN + ch * 8 + 4);
            float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            if (R) {
              float r8[8];
              chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(R + (long)row * P.Cout + col), r8);
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] += r8[e];
            }
            if (P.leaky) {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * P.slope;
            }
            const i32x4 pk = f32_to_chunk<T>(f);
            *reinterpret_cast<i32x4*>(O + (long)row * P.Cout + col) = pk;
            if (P.stat_sum != nullptr) {
              float q[8];
              chunk_to_f32<T>(pk, q);                                 // statistics of what the next stage will read
#pragma unroll
              for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
            }
          }
          if (P.stat_sum != nullptr) {
            // the threads of a chunk column sit CH lanes apart: fold them inside each wave first
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
              for (int o = CH; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o); s2[e] += __shfl_xor(s2[e], o); }
            }
            if (lane < CH) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { atomicAdd(&stat_l[0][ch * 8 + e], s1[e]); atomicAdd(&stat_l[1][ch * 8 + e], s2[e]); }
            }
          }
        }
      }
      if (P.stat_sum != nullptr) {
        __syncthreads();
        if (tid < BN && n0 + tid < P.Cout && !WMZ_ABL_NOSTAT) {
          const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * P.Cout;
          atomicAdd(P.stat_sum + rep + n0 + tid, stat_l[0][tid]);
          atomicAdd(P.stat_sq + rep + n0 + tid, stat_l[1][tid]);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc + 32 * j + l31;
    const bool cok = col < P.Cout;
    const float bv = (cok && P.bias) ? P.bias[col] : 0.f;
    const float sc = (cok && P.scale) ? P.scale[col] : 1.f;
    const float sh = (cok && P.shift) ? P.shift[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wr + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (!cok || row >= P.M) continue;
        float v = (acc[i][j][reg] + bv) * sc + sh;
        if (R) v += Elem<T>::to_f32(R[(long)row * P.Cout + col]);
        if (P.leaky) v = v > 0.f ? v : v * P.slope;
        const T tv = Elem<T>::from_f32(v);
        O[(long)row * P.Cout + col] = tv;
        const float q = Elem<T>::to_f32(tv);       // statistics of what the next stage will read
        s1 += q;
        s2 += q * q;
      }
    if (P.stat_sum != nullptr) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (hh == 0 && cok) {
        const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * P.Cout;
        atomicAdd(P.stat_sum + rep + col, s1);
        atomicAdd(P.stat_sq + rep + col, s2);
      }
    }
  }
}

// 3 x 3, stride 1, pad 1, Cin = 64 -> Cout = 128 in bf16 on planes whose width is a multiple of 32 and height of 8 (the encoder's
// big convolutions: autoencoder.py:29 conv3x3 inside Residual): a DIRECT form of the same GEMM.  The implicit-GEMM kernel
// re-reads every input pixel nine times (once per tap) and the weights once per 128-pixel tile through L2 -- 2.4 GB for
// the 64 x 64 layer.  Here a workgroup owns an 8 x 32 block of output pixels of one image: its 10 x 34 input patch goes
// to LDS ONCE (43.5 KB) and the nine taps read it at shifted pixel addresses; the weights stream tap by tap through a
// double-buffered 16 KB slab (register prefetch, one barrier per tap).  A wave owns two output rows (64 pixels) x all 128
// channels.  Same K order (tap-major, channels inside) and the same epilogue arithmetic as conv2d_kernel: bit-identical.
constexpr int D3_TH = 8, D3_TW = 32, D3_PH = D3_TH + 2, D3_PW = D3_TW + 2, D3_CIN = 64, D3_COUT = 128;
__global__ __launch_bounds__(NT, 2) void conv3x3s1_kernel(ConvParams P) {
  using T = bf16_t;
  constexpr int PATCH = D3_PH * D3_PW * 128;                   // bytes: a pixel = 64 channels = 128 B = 8 chunks of 16 B
  __shared__ __attribute__((aligned(16))) char patch[PATCH];     // chunk c of pixel p at p*128 + ((c ^ (p & 7)) << 4)
  __shared__ __attribute__((aligned(16))) char Bs2[2][D3_COUT * ROWB];
  __shared__ float stat_l[2][D3_COUT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int tw = P.Wo / D3_TW, th = P.Ho / D3_TH;
  int bid = blockIdx.x;
  const int bx = bid % tw; bid /= tw;
  const int by = bid % th;
  const int b = bid / th;
  const int oy0 = by * D3_TH, ox0 = bx * D3_TW;
  const T* X = reinterpret_cast<const T*>(P.x);
  const T* Wt = reinterpret_cast<const T*>(P.w);
  constexpr int K = 9 * D3_CIN;

  // ---- input patch -> LDS (zeros outside the image)
  for (int idx = tid; idx < D3_PH * D3_PW * 8; idx += NT) {
    const int p = idx >> 3, c = idx & 7;
    const int py = p / D3_PW, px = p - py * D3_PW;
    const int hi = oy0 + py - 1, wi = ox0 + px - 1;
    i32x4 v = (i32x4)(0);
    if (hi >= 0 && hi < P.Hi && wi >= 0 && wi < P.Wi)
      v = *reinterpret_cast<const i32x4*>(X + (((long)b * P.Hi + hi) * P.Wi + wi) * D3_CIN + c * 8);
    *reinterpret_cast<i32x4*>(patch + p * 128 + ((c ^ (p & 7)) << 4)) = v;
  }
  // ---- weights: tap slab [128 n][64 k], 4 chunks per thread
  const int cc = tid & 7, rr = tid >> 3;
  i32x4 rb[4];
  auto fetchB = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const i32x4*>(Wt + (long)(rr + 32 * i) * K + tap * D3_CIN + cc * 8);
  };
  auto stashB = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 32 * i;
      *reinterpret_cast<i32x4*>(Bs2[buf] + r * ROWB + ((cc << 4) ^ swz128(r))) = rb[i];
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x16)(0.f);
  fetchB(0);
  stashB(0);
  fetchB(1);
  __syncthreads();
  for (int tap = 0; tap < 9; ++tap) {
    const int cur = tap & 1;
    if (tap + 1 < 9) stashB(cur ^ 1);                            // (its last readers passed the barrier that ended tap - 1)
    if (tap + 2 < 9) fetchB(tap + 2);
    const int kh = tap / 3, kw = tap - kh * 3;
    const char* Bs = Bs2[cur];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      Frag8<T> af[2], bf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = (2 * wave + i + kh) * D3_PW + l31 + kw;    // the pixel this lane's output pixel reads at this tap
        af[i].v = *reinterpret_cast<const s16x8*>(patch + p * 128 + (((kk * 2 + hh) ^ (p & 7)) << 4));
      }
      const int b0 = (kk * 16 + hh * 8) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rn = 32 * j + l31;
        bf[j].v = *reinterpret_cast<const s16x8*>(Bs + rn * ROWB + (b0 ^ swz128(rn)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma32(acc[i][j], af[i], bf[j]);
    }
    __syncthreads();
  }

  // ---- epilogue (as conv2d_kernel's staged form): a wave's 64 pixels x 128 channels per round through LDS in fp32, then
  // 16-byte row chunks: residual, LeakyReLU, rounding, store, statistics
  const T* R = reinterpret_cast<const T*>(P.res);
  T* O = reinterpret_cast<T*>(P.out);
  float* stage = reinterpret_cast<float*>(patch);               // [64][128] fp32 = 32 KB (the patch is dead)
  static_assert(PATCH >= 64 * 128 * 4, "stage fits the patch");
  if (tid < D3_COUT) { stat_l[0][tid] = 0.f; stat_l[1][tid] = 0.f; }
#pragma unroll 1
  for (int round = 0; round < 4; ++round) {
    __syncthreads();
    if (wave == round) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = 32 * j + l31;
          const float bv = P.bias ? P.bias[col] : 0.f;
          const float sc = P.scale ? P.scale[col] : 1.f;
          const float sh = P.shift ? P.shift[col] : 0.f;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int rl = 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            stage[rl * D3_COUT + col] = (acc[i][j][reg] + bv) * sc + sh;
          }
        }
    }
    __syncthreads();
    const int ch = tid & 15, col = ch * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = (tid >> 4) + 16 * it;                       // local pixel: row 2 round + rl / 32, column rl % 32
      const long row = ((long)b * P.Ho + oy0 + 2 * round + (rl >> 5)) * P.Wo + ox0 + (rl & 31);
      const f32x4 a = *reinterpret_cast<const f32x4*>(stage + rl * D3_COUT + col);
      const f32x4 bq = *reinterpret_cast<const f32x4*>(stage + rl * D3_COUT + col + 4);
      float f[8] = {a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]};
      if (R) {
        float r8[8];
        chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(R + row * D3_COUT + col), r8);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += r8[e];
      }
      if (P.leaky) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * P.slope;
      }
      const i32x4 pk = f32_to_chunk<T>(f);
      *reinterpret_cast<i32x4*>(O + row * D3_COUT + col) = pk;
      if (P.stat_sum != nullptr) {
        float q[8];
        chunk_to_f32<T>(pk, q);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += q[e]; s2[e] += q[e] * q[e]; }
      }
    }
    if (P.stat_sum != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], 16); s1[e] += __shfl_xor(s1[e], 32);
        s2[e] += __shfl_xor(s2[e], 16); s2[e] += __shfl_xor(s2[e], 32);
      }
      if ((lane >> 4) == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { atomicAdd(&stat_l[0][col + e], s1[e]); atomicAdd(&stat_l[1][col + e], s2[e]); }
      }
    }
  }
  if (P.stat_sum != nullptr) {
    __syncthreads();
    if (tid < D3_COUT && !WMZ_ABL_NOSTAT) {
      const int rep = (int)(blockIdx.x % WMZ_STAT_REPLICAS) * D3_COUT;
      atomicAdd(P.stat_sum + rep + tid, stat_l[0][tid]);
      atomicAdd(P.stat_sq + rep + tid, stat_l[1][tid]);
    }
  }
}

// per-channel sum / sumsq of an NHWC tensor [M, C]: a wave sweeps 64 channels x a slice of rows
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, long M, int C, float* __restrict__ sum,
                                                            float* __restrict__ sq) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (long m = (long)blockIdx.x * 4 + sub; m < M; m += (long)gridDim.x * 4) {
      const float v = Elem<T>::to_f32(x[m * C + c]);
      s1 += v;
      s2 += v * v;
    }
  __shared__ float r1[4][64], r2[4][64];
  r1[sub][threadIdx.x & 63] = s1;
  r2[sub][threadIdx.x & 63] = s2;
  __syncthreads();
  if (sub == 0 && c < C) {
    const int l = threadIdx.x;
    const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * C;
    atomicAdd(sum + rep + c, r1[0][l] + r1[1][l] + r1[2][l] + r1[3][l]);
    atomicAdd(sq + rep + c, r2[0][l] + r2[1][l] + r2[2][l] + r2[3][l]);
  }
}

// nn.BatchNorm2d bookkeeping for one layer: batch mean / biased var -> (scale, shift); running stats with momentum and
// the unbiased variance; training==0 folds the running statistics instead.
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sq, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                   float eps, int training, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out, int C,
                                   int64_t* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && num_batches_tracked != nullptr) *num_batches_tracked += 1;     // nn.BatchNorm2d's step counter
  if (c >= C) return;
  if (training) {                                             // (bn_lazy.h: the arithmetic shared with the consumers that finalise themselves)
    BnStats b{};
    b.sum = sum; b.sq = sq; b.gamma = gamma; b.beta = beta; b.running_mean = running_mean; b.running_var = running_var;
    b.nbt = nullptr;                                          // (counted above)
    b.scale = scale; b.shift = shift; b.mean = mean_out; b.rstd = rstd_out;
    b.count = count; b.momentum = momentum; b.eps = eps;
    float sc, sh;
    bn_channel(b, C, c, true, sc, sh);
    return;
  }
  const float mean = running_mean[c], var = running_var[c];
  const float rs = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * rs;
  shift[c] = b - mean * g * rs;
  if (mean_out) { mean_out[c] = mean; rstd_out[c] = rs; }
}

// y = act( a*sa[c] + ta[c]  (+ b*sb[c] + tb[c]) ),  act = LeakyReLU(slope) or identity; NHWC [M, C], C % 4 == 0
template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ a, const float* __restrict__ sa,
                                                         const float* __restrict__ ta, const T* __restrict__ b,
                                                         const float* __restrict__ sb, const float* __restrict__ tb,
                                                         T* __restrict__ y, long total, int C, int leaky, float slope) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (long)gridDim.x * blockDim.x * 4) {
    const int c = (int)(i % C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = Elem<T>::to_f32(a[i + e]);
      if (sa) v = v * sa[c + e] + ta[c + e];
      if (b) {
        float u = Elem<T>::to_f32(b[i + e]);
        if (sb) u = u * sb[c + e] + tb[c + e];
        v += u;
      }
      if (leaky) v = v > 0.f ? v : v * slope;
      y[i + e] = Elem<T>::from_f32(v);
    }
  }
}

// The same on 16-byte vectors (8 bf16 / 4 fp32 per lane and access; C a multiple of the vector width, 16-byte aligned
// tensors): a thread keeps its channel group for the whole launch when the grid stride is a multiple of C / VW vectors --
// the host picks the grid that way -- so its scale / shift values are loaded once; 32-bit index arithmetic.
// (The scalar form above moved 2 bytes per access and paid a 64-bit modulo per 4 elements: 6-10x off the HBM rate.)
// Shape of the streaming kernels below (affine_act, BatchNorm backward, channel statistics), measured on the frame encoder's and the
// VQ-AE step's tensors (tools/time_affine.py): (i) a grid that is resident at once (<= EW_GRID workgroups of 256 threads, 8 per CU)
// instead of 8192 short-lived ones; (ii) the per-channel constants come from an LDS table the workgroup fills once (one channel per
// thread, coalesced) -- the first version had every thread fetch its 8 channels of 4-7 arrays by 32-56 global_load_dword, most of
// a 100-MB launch; (iii) U vectors per operand in flight per thread, pinned by issued(): hipcc otherwise sinks the second
// operand's load behind the first one's use (load, wait, convert, load, wait).  affine_act on 100 MB: 40 -> 20 us.
constexpr int EW_GRID = 2048;

__device__ __forceinline__ void issued(i32x4& a, i32x4& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void issued(i32x4& a, i32x4& b, i32x4& c, i32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void issued(i32x4& a, i32x4& b, i32x4& c, i32x4& d, i32x4& e, i32x4& f) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void issued(i32x4& a, i32x4& b, i32x4& c, i32x4& d, i32x4& e, i32x4& f, i32x4& g, i32x4& h) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

// row k of the workgroup's LDS table [rows][C]: this thread's VW channels
template <int VW>
__device__ __forceinline__ void tab_row(const float* tab, int C, int k, int c0, float* out) {
#pragma unroll
  for (int e = 0; e < VW; e += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(tab + k * C + c0 + e);
    out[e] = v[0]; out[e + 1] = v[1]; out[e + 2] = v[2]; out[e + 3] = v[3];
  }
}

template <typename T>
__device__ __forceinline__ i32x4 ld16(const T* p, long i) { return *reinterpret_cast<const i32x4*>(p + i * (16 / (int)sizeof(T))); }

template <typename T>
__global__ __launch_bounds__(256) void affine_act_vec_kernel(const T* __restrict__ a, const float* __restrict__ sa,
                                                             const float* __restrict__ ta, const T* __restrict__ b,
                                                             const float* __restrict__ sb, const float* __restrict__ tb,
                                                             T* __restrict__ y, long nvec, int C, int leaky, float slope,
                                                             BnStats bna, BnStats bnb) {
  // bna / bnb (sum != nullptr): the affine of that operand is a training-mode BatchNorm whose statistics arrive raw -- finalised here
  // while the table is filled, published by workgroup 0 (bn_lazy.h)
  constexpr int VW = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float ew_tab[];       // [4][C]: sa, ta, sb, tb
  for (int c = threadIdx.x; c < C; c += 256) {
    float s1v = sa ? sa[c] : 1.f, t1v = sa ? ta[c] : 0.f, s2v = sb ? sb[c] : 1.f, t2v = sb ? tb[c] : 0.f;
    if (bna.sum != nullptr) bn_channel(bna, C, c, blockIdx.x == 0, s1v, t1v);
    if (bnb.sum != nullptr) bn_channel(bnb, C, c, blockIdx.x == 0, s2v, t2v);
    ew_tab[c] = s1v; ew_tab[C + c] = t1v; ew_tab[2 * C + c] = s2v; ew_tab[3 * C + c] = t2v;
  }
  __syncthreads();
  const int cv = C / VW;                                        // vectors per pixel
  const long stride = (long)gridDim.x * blockDim.x;             // a multiple of cv (host)
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(i % cv) * VW;
  float s1[VW], t1[VW], s2[VW], t2[VW];
  tab_row<VW>(ew_tab, C, 0, c0, s1); tab_row<VW>(ew_tab, C, 1, c0, t1);
  tab_row<VW>(ew_tab, C, 2, c0, s2); tab_row<VW>(ew_tab, C, 3, c0, t2);
  auto finish = [&](const i32x4& va, const i32x4& vb, long at) {
    float f[VW];
    chunk_to_f32<T>(va, f);
#pragma unroll
    for (int e = 0; e < VW; ++e) f[e] = fmaf(f[e], s1[e], t1[e]);
    if (b) {
      float u[VW];
      chunk_to_f32<T>(vb, u);
#pragma unroll
      for (int e = 0; e < VW; ++e) f[e] += fmaf(u[e], s2[e], t2[e]);
    }
    if (leaky) {
#pragma unroll
      for (int e = 0; e < VW; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * slope;
    }
    *reinterpret_cast<i32x4*>(y + at * VW) = f32_to_chunk<T>(f);
  };
  if (b) {
    for (; i + stride < nvec; i += 2 * stride) {
      i32x4 a0 = ld16(a, i), a1 = ld16(a, i + stride), b0 = ld16(b, i), b1 = ld16(b, i + stride);
      issued(a0, a1, b0, b1);
      finish(a0, b0, i); finish(a1, b1, i + stride);
    }
  } else {
    for (; i + 3 * stride < nvec; i += 4 * stride) {
      i32x4 a0 = ld16(a, i), a1 = ld16(a, i + stride), a2 = ld16(a, i + 2 * stride), a3 = ld16(a, i + 3 * stride);
      issued(a0, a1, a2, a3);
      finish(a0, a0, i); finish(a1, a1, i + stride); finish(a2, a2, i + 2 * stride); finish(a3, a3, i + 3 * stride);
    }
  }
  for (; i < nvec; i += stride) {
    i32x4 a0 = ld16(a, i), b0 = b ? ld16(b, i) : a0;
    issued(a0, b0);
    finish(a0, b0, i);
  }
}

// dz[b, s y, s x, :] = dy[b, y, x, :], zero elsewhere: the zero-inserted plane on which a strided convolution's data gradient runs as a
// stride-1 convolution (autoencoder._Conv2dFn.backward).  One pass of 16-byte stores (the torch form: a fill + a strided copy that
// ran at a few hundred GB/s).  One thread per output vector, rows of the output plane on blockIdx.y.
template <typename T>
__global__ __launch_bounds__(256) void dilate_vec_kernel(const T* __restrict__ dy, T* __restrict__ dz, int Ho, int Wo, int cv, int Hz,
                                                         int Wz, int stride) {
  constexpr int VW = 16 / (int)sizeof(T);
  const int rowv = Wz * cv;                                       // vectors per output row
  const int b = blockIdx.z, y = blockIdx.y;
  const bool yin = (y % stride) == 0 && y / stride < Ho;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rowv; i += gridDim.x * 256) {
    const int x = i / cv, c = i - x * cv;
    i32x4 v = (i32x4)(0);
    if (yin && (x % stride) == 0 && x / stride < Wo)
      v = *reinterpret_cast<const i32x4*>(dy + (((long)b * Ho + y / stride) * Wo + x / stride) * cv * VW + c * VW);
    *reinterpret_cast<i32x4*>(dz + (((long)b * Hz + y) * Wz) * cv * VW + (long)i * VW) = v;
  }
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) on NHWC
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float sh = fmaxf((ho + 0.5f) * 0.5f - 0.5f, 0.f), sw = fmaxf((wo + 0.5f) * 0.5f - 0.5f, 0.f);
    const int h0 = (int)sh, w0 = (int)sw;
    const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
    const float lh = sh - h0, lw = sw - w0;
    const T* p = x + (long)b * H * W * C + c;
    const float v00 = Elem<T>::to_f32(p[((long)h0 * W + w0) * C]), v01 = Elem<T>::to_f32(p[((long)h0 * W + w1) * C]);
    const float v10 = Elem<T>::to_f32(p[((long)h1 * W + w0) * C]), v11 = Elem<T>::to_f32(p[((long)h1 * W + w1) * C]);
    // ATen's upsample_bilinear2d order: interpolate along w inside each row, then along h
    const float top = (1.f - lw) * v00 + lw * v01, bot = (1.f - lw) * v10 + lw * v11;
    y[i] = Elem<T>::from_f32((1.f - lh) * top + lh * bot);
  }
}

// ---- backward companions (training-mode BatchNorm + LeakyReLU, bilinear x2)
// g = dy * act'(y)  (act' from the sign of the stored post-activation y); per channel: sum_g += g, sum_gx += g * xhat
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                const T* __restrict__ dy, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, T* __restrict__ g_out,
                                                                float* __restrict__ sum_g, float* __restrict__ sum_gx,
                                                                long M, int C, int leaky, float slope) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean ? mean[c] : 0.f, rs = rstd ? rstd[c] : 1.f;
    for (long m = (long)blockIdx.x * 4 + sub; m < M; m += (long)gridDim.x * 4) {
      float g = Elem<T>::to_f32(dy[m * C + c]);
      if (leaky && Elem<T>::to_f32(y[m * C + c]) <= 0.f) g *= slope;
      if (g_out) g_out[m * C + c] = Elem<T>::from_f32(g);
      s1 += g;
      if (x) s2 += g * (Elem<T>::to_f32(x[m * C + c]) - mu) * rs;
    }
  }
  __shared__ float r1[4][64], r2[4][64];
  r1[sub][threadIdx.x & 63] = s1;
  r2[sub][threadIdx.x & 63] = s2;
  __syncthreads();
  if (sub == 0 && c < C && sum_g) {
    const int l = threadIdx.x;
    atomicAdd(sum_g + c, r1[0][l] + r1[1][l] + r1[2][l] + r1[3][l]);
    atomicAdd(sum_gx + c, r2[0][l] + r2[1][l] + r2[2][l] + r2[3][l]);
  }
}

// dx = gamma * rstd * (g - sum_g/M - xhat * sum_gx/M)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                           const float* __restrict__ sum_gx, T* __restrict__ dx, long M,
                                                           int C, const T* __restrict__ add) {
  const long total = M * C;
  const float invM = 1.f / (float)M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float rs = rstd[c];
    const float xh = (Elem<T>::to_f32(x[i]) - mean[c]) * rs;
    float v = gamma[c] * rs * (Elem<T>::to_f32(g[i]) - sum_g[c] * invM - xh * sum_gx[c] * invM);
    if (add) v += Elem<T>::to_f32(add[i]);
    dx[i] = Elem<T>::from_f32(v);
  }
}

// adjoint of bilinear2x_kernel, gather form (deterministic): each source pixel collects from the <= 6x6 destination
// pixels that could reference it, recomputing the forward's indices and weights
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W,
                                                             int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float acc = 0.f;
    for (int ho = max(2 * h - 2, 0); ho <= min(2 * h + 3, Ho - 1); ++ho) {
      const float sh = fmaxf((ho + 0.5f) * 0.5f - 0.5f, 0.f);
      const int h0 = (int)sh, h1 = min(h0 + 1, H - 1);
      const float lh = sh - h0;
      const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (wh == 0.f) continue;
      for (int wo = max(2 * w - 2, 0); wo <= min(2 * w + 3, Wo - 1); ++wo) {
        const float sw = fmaxf((wo + 0.5f) * 0.5f - 0.5f, 0.f);
        const int w0 = (int)sw, w1 = min(w0 + 1, W - 1);
        const float lw = sw - w0;
        const float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
        if (ww != 0.f) acc += wh * ww * Elem<T>::to_f32(dy[(((long)b * Ho + ho) * Wo + wo) * C + c]);
      }
    }
    dx[i] = Elem<T>::from_f32(acc);
  }
}

// ---- 16-byte versions of the training-path elementwise kernels (the scalar forms above move 2 bytes per access: 3-4x off the
// HBM rate on the 8-270 MB tensors of a VQ-AE training step).  A thread keeps its channel group (VW channels) for the whole
// launch -- the grid stride is a multiple of C / VW vectors, the host picks the grid that way (256 % (C / VW) == 0) -- so per-channel
// constants are loaded once and per-channel sums stay in registers; the threads of a workgroup that share a channel group meet in
// LDS, one atomic per channel and workgroup (<= 256 workgroups: chains of same-address atomics cost ~15 ns a link).
constexpr int RED_NT = 512;      // threads per workgroup of the two reducing kernels: <= 256 workgroups (the atomic chain), 8 waves per CU
template <typename T>
__global__ __launch_bounds__(RED_NT) void bn_act_bwd_reduce_vec_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                    const T* __restrict__ dy, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, T* __restrict__ g_out,
                                                                    float* __restrict__ sum_g, float* __restrict__ sum_gx,
                                                                    long nvec, int C, int leaky, float slope,
                                                                    const float* __restrict__ msc, const float* __restrict__ msh) {
  // msc / msh (optional): the forward's scale / shift -- the LeakyReLU mask is recomputed from x (sign of fmaf(x, scale, shift), the
  // expression affine_act_vec_kernel evaluated) instead of read from the stored output: one tensor less to read, none to write
  constexpr int VW = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float ew_tab[];       // [4][C]: mean, rstd, msc, msh; behind the loop: red[2][RED_NT][VW + 1]
  for (int c = threadIdx.x; c < C; c += RED_NT) {
    ew_tab[c] = mean ? mean[c] : 0.f; ew_tab[C + c] = rstd ? rstd[c] : 1.f;
    ew_tab[2 * C + c] = msc ? msc[c] : 0.f; ew_tab[3 * C + c] = msc ? msh[c] : 0.f;
  }
  __syncthreads();
  const int cv = C / VW;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(i % cv) * VW;
  float mu[VW], rs[VW], s1[VW], s2[VW];
  tab_row<VW>(ew_tab, C, 0, c0, mu); tab_row<VW>(ew_tab, C, 1, c0, rs);
#pragma unroll
  for (int e = 0; e < VW; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (msc != nullptr) {
    float ms[VW], mt[VW];
    tab_row<VW>(ew_tab, C, 2, c0, ms); tab_row<VW>(ew_tab, C, 3, c0, mt);
    auto take = [&](const i32x4& vg, const i32x4& vx) {
      float g[VW], xv[VW];
      chunk_to_f32<T>(vg, g);
      chunk_to_f32<T>(vx, xv);
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        if (fmaf(xv[e], ms[e], mt[e]) <= 0.f) g[e] *= slope;
        s1[e] += g[e];
        s2[e] += g[e] * (xv[e] - mu[e]) * rs[e];
      }
    };
    for (; i + 3 * stride < nvec; i += 4 * stride) {
      i32x4 g0 = ld16(dy, i), g1 = ld16(dy, i + stride), g2 = ld16(dy, i + 2 * stride), g3 = ld16(dy, i + 3 * stride);
      i32x4 x0 = ld16(x, i), x1 = ld16(x, i + stride), x2 = ld16(x, i + 2 * stride), x3 = ld16(x, i + 3 * stride);
      issued(g0, g1, g2, g3, x0, x1, x2, x3);
      take(g0, x0); take(g1, x1); take(g2, x2); take(g3, x3);
    }
    for (; i < nvec; i += stride) {
      i32x4 g0 = ld16(dy, i), x0 = ld16(x, i);
      issued(g0, x0);
      take(g0, x0);
    }
  } else {
    auto take = [&](const i32x4& vg, const i32x4& vy, const i32x4& vx, long at) {
      float g[VW];
      chunk_to_f32<T>(vg, g);
      if (leaky) {
        float yv[VW];
        chunk_to_f32<T>(vy, yv);
#pragma unroll
        for (int e = 0; e < VW; ++e) g[e] = yv[e] <= 0.f ? g[e] * slope : g[e];
      }
      if (g_out) {
        const i32x4 gv = f32_to_chunk<T>(g);
        *reinterpret_cast<i32x4*>(g_out + at * VW) = gv;
        chunk_to_f32<T>(gv, g);                           // the sums see what was stored (as the scalar kernel's do)
      }
#pragma unroll
      for (int e = 0; e < VW; ++e) s1[e] += g[e];
      if (x) {
        float xv[VW];
        chunk_to_f32<T>(vx, xv);
#pragma unroll
        for (int e = 0; e < VW; ++e) s2[e] += g[e] * (xv[e] - mu[e]) * rs[e];
      }
    };
    for (; i + stride < nvec; i += 2 * stride) {
      i32x4 g0 = ld16(dy, i), g1 = ld16(dy, i + stride);
      i32x4 y0 = leaky ? ld16(y, i) : g0, y1 = leaky ? ld16(y, i + stride) : g1;
      i32x4 x0 = x ? ld16(x, i) : g0, x1 = x ? ld16(x, i + stride) : g1;
      issued(g0, g1, y0, y1, x0, x1);
      take(g0, y0, x0, i); take(g1, y1, x1, i + stride);
    }
    for (; i < nvec; i += stride) {
      i32x4 g0 = ld16(dy, i);
      i32x4 y0 = leaky ? ld16(y, i) : g0, x0 = x ? ld16(x, i) : g0;
      issued(g0, y0); issued(g0, x0);
      take(g0, y0, x0, i);
    }
  }
  if (sum_g == nullptr) return;
  __syncthreads();                                              // every thread is through with the table: the space is red[][][] now
  float (*red)[RED_NT][VW + 1] = reinterpret_cast<float (*)[RED_NT][VW + 1]>(ew_tab);
#pragma unroll
  for (int e = 0; e < VW; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  for (int t = threadIdx.x; t < C; t += RED_NT) {       // channel t: its partners are the threads t / VW, t / VW + cv, ..
    const int grp = t / VW, e = t - grp * VW;
    float a = 0.f, b = 0.f;
    for (int p = grp; p < RED_NT; p += cv) { a += red[0][p][e]; b += red[1][p][e]; }
    atomicAdd(sum_g + t, a);
    atomicAdd(sum_gx + t, b);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                               const float* __restrict__ sum_gx, T* __restrict__ dx, long nvec,
                                                               int C, float invM, const float* __restrict__ msc,
                                                               const float* __restrict__ msh, float slope,
                                                               const T* __restrict__ add) {
  // dx = gamma rstd (g - mean(g) - xhat mean(g xhat)) [+ add];  msc / msh (optional): g holds the gradient BEHIND the LeakyReLU and
  // the mask is recomputed from x (see the reducing kernel);  add (optional): the gradient x receives from its OTHER consumer (the
  // skip path of a residual block) -- summed here instead of by a pass of its own behind this kernel
  constexpr int VW = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float ew_tab[];       // [6][C]: gamma rstd, mean(g), mean, rstd mean(g xhat), msc, msh
  for (int c = threadIdx.x; c < C; c += 256) {
    const float rs = rstd[c];
    ew_tab[c] = gamma[c] * rs; ew_tab[C + c] = sum_g[c] * invM; ew_tab[2 * C + c] = mean[c]; ew_tab[3 * C + c] = rs * (sum_gx[c] * invM);
    ew_tab[4 * C + c] = msc ? msc[c] : 0.f; ew_tab[5 * C + c] = msc ? msh[c] : 1.f;          // (no mask: 0 x + 1 > 0)
  }
  __syncthreads();
  const int cv = C / VW;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(i % cv) * VW;
  float gr[VW], sg[VW], mu[VW], q[VW], ms[VW], mt[VW];
  tab_row<VW>(ew_tab, C, 0, c0, gr); tab_row<VW>(ew_tab, C, 1, c0, sg); tab_row<VW>(ew_tab, C, 2, c0, mu);
  tab_row<VW>(ew_tab, C, 3, c0, q); tab_row<VW>(ew_tab, C, 4, c0, ms); tab_row<VW>(ew_tab, C, 5, c0, mt);
  auto finish = [&](const i32x4& vx, const i32x4& vg, const i32x4& va, long at) {
    float xv[VW], gv[VW];
    chunk_to_f32<T>(vx, xv);
    chunk_to_f32<T>(vg, gv);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      if (fmaf(xv[e], ms[e], mt[e]) <= 0.f) gv[e] *= slope;
      gv[e] = gr[e] * (gv[e] - sg[e] - (xv[e] - mu[e]) * q[e]);
    }
    if (add) {
      float av[VW];
      chunk_to_f32<T>(va, av);
#pragma unroll
      for (int e = 0; e < VW; ++e) gv[e] += av[e];
    }
    *reinterpret_cast<i32x4*>(dx + at * VW) = f32_to_chunk<T>(gv);
  };
  for (; i + stride < nvec; i += 2 * stride) {
    i32x4 x0 = ld16(x, i), x1 = ld16(x, i + stride), g0 = ld16(g, i), g1 = ld16(g, i + stride);
    i32x4 a0 = add ? ld16(add, i) : g0, a1 = add ? ld16(add, i + stride) : g1;
    issued(x0, x1, g0, g1, a0, a1);
    finish(x0, g0, a0, i); finish(x1, g1, a1, i + stride);
  }
  for (; i < nvec; i += stride) {
    i32x4 x0 = ld16(x, i), g0 = ld16(g, i);
    i32x4 a0 = add ? ld16(add, i) : g0;
    issued(x0, g0); issued(x0, a0);
    finish(x0, g0, a0, i);
  }
}

template <typename T>
__global__ __launch_bounds__(RED_NT) void channel_stats_vec_kernel(const T* __restrict__ x, long nvec, int C, float* __restrict__ sum,
                                                                float* __restrict__ sq) {
  constexpr int VW = 16 / (int)sizeof(T);
  const int cv = C / VW;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float s1[VW], s2[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  auto take = [&](const i32x4& vx) {
    float v[VW];
    chunk_to_f32<T>(vx, v);
#pragma unroll
    for (int e = 0; e < VW; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
  };
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    i32x4 x0 = ld16(x, i), x1 = ld16(x, i + stride), x2 = ld16(x, i + 2 * stride), x3 = ld16(x, i + 3 * stride);
    issued(x0, x1, x2, x3);
    take(x0); take(x1); take(x2); take(x3);
  }
  for (; i < nvec; i += stride) take(ld16(x, i));
  __shared__ float red[2][RED_NT][VW + 1];
#pragma unroll
  for (int e = 0; e < VW; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  for (int t = threadIdx.x; t < C; t += RED_NT) {
    const int grp = t / VW, e = t - grp * VW;
    float a = 0.f, b = 0.f;
    for (int p = grp; p < RED_NT; p += cv) { a += red[0][p][e]; b += red[1][p][e]; }
    const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * C;
    atomicAdd(sum + rep + t, a);
    atomicAdd(sq + rep + t, b);
  }
}

// bilinear x2 forward / adjoint, one 16-byte channel vector of one output pixel per thread (same arithmetic and order as the
// scalar kernels)
template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int VW = 16 / (int)sizeof(T);
  const int cv = C / VW, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * Ho * Wo * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VW;
    long t = i / cv;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float sh = fmaxf((ho + 0.5f) * 0.5f - 0.5f, 0.f), sw = fmaxf((wo + 0.5f) * 0.5f - 0.5f, 0.f);
    const int h0 = (int)sh, w0 = (int)sw;
    const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
    const float lh = sh - h0, lw = sw - w0;
    const T* p = x + (long)b * H * W * C + c;
    float v00[VW], v01[VW], v10[VW], v11[VW], o[VW];
    chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(p + ((long)h0 * W + w0) * C), v00);
    chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(p + ((long)h0 * W + w1) * C), v01);
    chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(p + ((long)h1 * W + w0) * C), v10);
    chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(p + ((long)h1 * W + w1) * C), v11);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const float top = (1.f - lw) * v00[e] + lw * v01[e], bot = (1.f - lw) * v10[e] + lw * v11[e];
      o[e] = (1.f - lh) * top + lh * bot;
    }
    *reinterpret_cast<i32x4*>(y + i * VW) = f32_to_chunk<T>(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear2x_bwd_vec_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W,
                                                                 int C) {
  constexpr int VW = 16 / (int)sizeof(T);
  const int cv = C / VW, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * H * W * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VW;
    long t = i / cv;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float acc[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) acc[e] = 0.f;
    for (int ho = max(2 * h - 2, 0); ho <= min(2 * h + 3, Ho - 1); ++ho) {
      const float sh = fmaxf((ho + 0.5f) * 0.5f - 0.5f, 0.f);
      const int h0 = (int)sh, h1 = min(h0 + 1, H - 1);
      const float lh = sh - h0;
      const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (wh == 0.f) continue;
      for (int wo = max(2 * w - 2, 0); wo <= min(2 * w + 3, Wo - 1); ++wo) {
        const float sw = fmaxf((wo + 0.5f) * 0.5f - 0.5f, 0.f);
        const int w0 = (int)sw, w1 = min(w0 + 1, W - 1);
        const float lw = sw - w0;
        const float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
        if (ww != 0.f) {
          float v[VW];
          chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(dy + (((long)b * Ho + ho) * Wo + wo) * C + c), v);
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[e] += wh * ww * v[e];
        }
      }
    }
    *reinterpret_cast<i32x4*>(dx + i * VW) = f32_to_chunk<T>(acc);
  }
}

// vector forms apply: C a multiple of the vector width, channel groups dividing the workgroup, 16-byte aligned tensors
static bool vec_ok(int C, int dtype, std::initializer_list<const void*> ptrs) {
  const int VW = dtype == WMZ_BF16 ? 8 : 4;
  if (C % VW != 0 || 256 % (C / VW) != 0) return false;
  for (const void* q : ptrs) if (q != nullptr && (((uintptr_t)q) & 15) != 0) return false;
  return true;
}

// dynamic LDS of the reducing BatchNorm-backward kernel: its constant table [4][C], re-used as red[2][RED_NT][VW + 1]
size_t red_lds(int C, int VW) {
  const size_t tab = (size_t)4 * C * sizeof(float), red = (size_t)2 * RED_NT * (VW + 1) * sizeof(float);
  return tab > red ? tab : red;
}

int grid_for(long total, int per_block, int cap) {
  long b = (total + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int wmz_conv2d_nhwc_fwd(const void* x, const void* w, void* out, const float* bias, const float* scale,
                                   const float* shift, const void* residual, float* stat_sum, float* stat_sq, int B,
                                   int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int leaky,
                                   float slope, int dtype, void* stream) {
  return wmz_conv2d_nhwc_fwd_pre(x, w, out, bias, scale, shift, residual, stat_sum, stat_sq, nullptr, nullptr, 0.f, B, Hi, Wi,
                                 Cin, Cout, KH, KW, stride, pad, leaky, slope, dtype, stream);
}

extern "C" int wmz_conv2d_nhwc_fwd_pre(const void* x, const void* w, void* out, const float* bias, const float* scale,
                                       const float* shift, const void* residual, float* stat_sum, float* stat_sq,
                                       const float* in_scale, const float* in_shift, float in_slope, int B, int Hi, int Wi,
                                       int Cin, int Cout, int KH, int KW, int stride, int pad, int leaky, float slope,
                                       int dtype, void* stream) {
  WMZ_REQUIRE(x && w && out, "wmz_conv2d_nhwc_fwd: null tensor");
  WMZ_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wmz_conv2d_nhwc_fwd: in_scale and in_shift go together");
  WMZ_REQUIRE(in_scale == nullptr || (KH == 1 && KW == 1 && pad == 0), "wmz_conv2d_nhwc_fwd: the input prologue is built for 1x1 convolutions without padding");
  WMZ_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
              "wmz_conv2d_nhwc_fwd: bad shape");
  WMZ_REQUIRE(Cin % 8 == 0, "wmz_conv2d_nhwc_fwd: Cin must be a multiple of 8 (zero-pad the input channels), got %d", Cin);
  WMZ_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "wmz_conv2d_nhwc_fwd: stat_sum and stat_sq go together");
  WMZ_REQUIRE((scale == nullptr) == (shift == nullptr), "wmz_conv2d_nhwc_fwd: scale and shift go together");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_conv2d_nhwc_fwd: bad dtype %d", dtype);
  ConvParams P;
  P.x = x; P.w = w; P.out = out; P.bias = bias; P.scale = scale; P.shift = shift; P.res = residual;
  P.stat_sum = stat_sum; P.stat_sq = stat_sq;
  P.B = B; P.Hi = Hi; P.Wi = Wi; P.Cin = Cin; P.Cout = Cout; P.KH = KH; P.KW = KW; P.stride = stride; P.pad = pad;
  P.Ho = (Hi + 2 * pad - KH) / stride + 1;
  P.Wo = (Wi + 2 * pad - KW) / stride + 1;
  WMZ_REQUIRE(P.Ho > 0 && P.Wo > 0, "wmz_conv2d_nhwc_fwd: empty output");
  const bool tall = Cout <= 64;
  const int BMl = tall ? 256 : 128, BNl = tall ? 64 : 128;
  P.M = B * P.Ho * P.Wo; P.K = KH * KW * Cin; P.nbn = wmz_cdiv(Cout, BNl);
  P.leaky = leaky; P.slope = slope;
  P.in_scale = in_scale; P.in_shift = in_shift; P.in_slope = in_slope;
  dim3 grid((unsigned)(wmz_cdiv(P.M, BMl) * P.nbn)), block(NT);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin == D3_CIN && Cout == D3_COUT &&
      in_scale == nullptr && P.Wo % D3_TW == 0 && P.Ho % D3_TH == 0) {
    hipLaunchKernelGGL(conv3x3s1_kernel, dim3((unsigned)(B * (P.Ho / D3_TH) * (P.Wo / D3_TW))), block, 0, st, P);
    WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_fwd");
    return WMZ_OK;
  }
  if (dtype == WMZ_BF16) {
    if (tall) hipLaunchKernelGGL((conv2d_kernel<bf16_t, true>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((conv2d_kernel<bf16_t, false>), grid, block, 0, st, P);
  } else {
    if (tall) hipLaunchKernelGGL((conv2d_kernel<float, true>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((conv2d_kernel<float, false>), grid, block, 0, st, P);
  }
  WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_fwd");
  return WMZ_OK;
}

extern "C" int wmz_channel_stats_nhwc(const void* x, long M, int C, float* sum, float* sq, int dtype, void* stream) {
  WMZ_REQUIRE(x && sum && sq && M > 0 && C > 0, "wmz_channel_stats_nhwc: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_channel_stats_nhwc: bad dtype %d", dtype);
  if (vec_ok(C, dtype, {x})) {
    const int VW = dtype == WMZ_BF16 ? 8 : 4;
    const long nvec = M * C / VW;
    const int gridv = grid_for(nvec, RED_NT * 4, 256);
    if (dtype == WMZ_BF16) hipLaunchKernelGGL(channel_stats_vec_kernel<bf16_t>, dim3(gridv), dim3(RED_NT), 0, (hipStream_t)stream, (const bf16_t*)x, nvec, C, sum, sq);
    else hipLaunchKernelGGL(channel_stats_vec_kernel<float>, dim3(gridv), dim3(RED_NT), 0, (hipStream_t)stream, (const float*)x, nvec, C, sum, sq);
    WMZ_LAUNCH_CHECK("wmz_channel_stats_nhwc");
    return WMZ_OK;
  }
  dim3 grid((unsigned)grid_for(M, 64, 512), (unsigned)wmz_cdiv(C, 64));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16) hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, M, C, sum, sq);
  else hipLaunchKernelGGL(channel_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, M, C, sum, sq);
  WMZ_LAUNCH_CHECK("wmz_channel_stats_nhwc");
  return WMZ_OK;
}

extern "C" int wmz_bn_finalize(const float* sum, const float* sq, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, double momentum, double eps, int training,
                               float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                               int64_t* num_batches_tracked, void* stream) {
  WMZ_REQUIRE(running_mean && running_var && scale && shift && C > 0, "wmz_bn_finalize: bad arguments");
  WMZ_REQUIRE(!training || (sum && sq && count > 0), "wmz_bn_finalize: training mode needs the batch statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(wmz_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sum, sq, (float)count,
                     gamma, beta, running_mean, running_var, (float)momentum, (float)eps, training, scale, shift, mean_out, rstd_out, C,
                     num_batches_tracked);
  WMZ_LAUNCH_CHECK("wmz_bn_finalize");
  return WMZ_OK;
}

extern "C" int wmz_affine_act_bn_supported(int C, int dtype) {
  const int VW = dtype == WMZ_BF16 ? 8 : 4;
  return (dtype == WMZ_F32 || dtype == WMZ_BF16) && C > 0 && C % VW == 0 && 256 % (C / VW) == 0 ? 1 : 0;
}

extern "C" int wmz_affine_act_nhwc_bn(const void* a, const float* sa, const float* ta, const wmz_bn_stats* bna, const void* b,
                                      const float* sb, const float* tb, const wmz_bn_stats* bnb, void* y, long M, int C, int leaky,
                                      float slope, int dtype, void* stream) {
  WMZ_REQUIRE(a && y && M > 0 && C > 0 && C % 4 == 0, "wmz_affine_act_nhwc: bad arguments (C %% 4 == 0 required)");
  WMZ_REQUIRE((sa == nullptr) == (ta == nullptr) && (sb == nullptr) == (tb == nullptr), "wmz_affine_act_nhwc: scale/shift pairs");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_affine_act_nhwc: bad dtype %d", dtype);
  WMZ_REQUIRE(bn_stats_ok(bna) && bn_stats_ok(bnb) && (bnb == nullptr || b != nullptr), "wmz_affine_act_nhwc_bn: incomplete wmz_bn_stats");
  const long total = M * C;
  hipStream_t st = (hipStream_t)stream;
  const int VW = dtype == WMZ_BF16 ? 8 : 4;
  const bool aligned = (((uintptr_t)a | (uintptr_t)y | (uintptr_t)(b ? b : a)) & 15) == 0;
  if (C % VW == 0 && aligned && 256 % (C / VW) == 0) {
    // every thread keeps its channel group: the grid stride (grid * 256 vectors) is a multiple of C / VW since 256 is
    const long nvec = total / VW;
    const int gridv = grid_for(nvec, 256, EW_GRID);
    if (dtype == WMZ_BF16)
      hipLaunchKernelGGL(affine_act_vec_kernel<bf16_t>, dim3(gridv), dim3(256), (size_t)4 * C * sizeof(float), st, (const bf16_t*)a, sa, ta, (const bf16_t*)b, sb, tb, (bf16_t*)y, nvec, C, leaky, slope, bn_stats_from(bna), bn_stats_from(bnb));
    else
      hipLaunchKernelGGL(affine_act_vec_kernel<float>, dim3(gridv), dim3(256), (size_t)4 * C * sizeof(float), st, (const float*)a, sa, ta, (const float*)b, sb, tb, (float*)y, nvec, C, leaky, slope, bn_stats_from(bna), bn_stats_from(bnb));
    WMZ_LAUNCH_CHECK("wmz_affine_act_nhwc");
    return WMZ_OK;
  }
  WMZ_REQUIRE(bna == nullptr && bnb == nullptr, "wmz_affine_act_nhwc_bn: raw BatchNorm statistics need the 16-byte kernel (wmz_affine_act_bn_supported, 16-byte aligned tensors)");
  const int grid = grid_for(total, 1024, 4096);
  if (dtype == WMZ_BF16)
    hipLaunchKernelGGL(affine_act_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)a, sa, ta, (const bf16_t*)b, sb, tb, (bf16_t*)y, total, C, leaky, slope);
  else
    hipLaunchKernelGGL(affine_act_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)a, sa, ta, (const float*)b, sb, tb, (float*)y, total, C, leaky, slope);
  WMZ_LAUNCH_CHECK("wmz_affine_act_nhwc");
  return WMZ_OK;
}

extern "C" int wmz_affine_act_nhwc(const void* a, const float* sa, const float* ta, const void* b, const float* sb,
                                   const float* tb, void* y, long M, int C, int leaky, float slope, int dtype,
                                   void* stream) {
  return wmz_affine_act_nhwc_bn(a, sa, ta, nullptr, b, sb, tb, nullptr, y, M, C, leaky, slope, dtype, stream);
}

extern "C" int wmz_dilate_nhwc(const void* dy, void* dz, int B, int Ho, int Wo, int C, int Hz, int Wz, int stride, int dtype,
                               void* stream) {
  WMZ_REQUIRE(dy && dz && B > 0 && Ho > 0 && Wo > 0 && C > 0 && stride >= 1, "wmz_dilate_nhwc: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_dilate_nhwc: bad dtype %d", dtype);
  WMZ_REQUIRE(Hz >= (Ho - 1) * stride + 1 && Wz >= (Wo - 1) * stride + 1, "wmz_dilate_nhwc: the output plane %d x %d does not hold %d x %d at stride %d", Hz, Wz, Ho, Wo, stride);
  const int VW = dtype == WMZ_BF16 ? 8 : 4;
  WMZ_REQUIRE(C % VW == 0 && (((uintptr_t)dy | (uintptr_t)dz) & 15) == 0, "wmz_dilate_nhwc: C must be a multiple of %d and the tensors 16-byte aligned", VW);
  WMZ_REQUIRE(Hz <= 65535 && B <= 65535, "wmz_dilate_nhwc: plane height / batch beyond the launch grid");
  const int cv = C / VW;
  dim3 grid((unsigned)wmz_cdiv(Wz * cv, 256), (unsigned)Hz, (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16) hipLaunchKernelGGL(dilate_vec_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dz, Ho, Wo, cv, Hz, Wz, stride);
  else hipLaunchKernelGGL(dilate_vec_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (float*)dz, Ho, Wo, cv, Hz, Wz, stride);
  WMZ_LAUNCH_CHECK("wmz_dilate_nhwc");
  return WMZ_OK;
}

extern "C" int wmz_bilinear2x_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  WMZ_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "wmz_bilinear2x_nhwc: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_bilinear2x_nhwc: bad dtype %d", dtype);
  const long total = (long)B * 4 * H * W * C;
  if (vec_ok(C, dtype, {x, y})) {
    const int VW = dtype == WMZ_BF16 ? 8 : 4;
    const int gridv = grid_for(total / VW, 256, 8192);
    if (dtype == WMZ_BF16) hipLaunchKernelGGL(bilinear2x_vec_kernel<bf16_t>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
    else hipLaunchKernelGGL(bilinear2x_vec_kernel<float>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, H, W, C);
    WMZ_LAUNCH_CHECK("wmz_bilinear2x_nhwc");
    return WMZ_OK;
  }
  const int grid = grid_for(total, 256, 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16) hipLaunchKernelGGL(bilinear2x_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
  else hipLaunchKernelGGL(bilinear2x_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C);
  WMZ_LAUNCH_CHECK("wmz_bilinear2x_nhwc");
  return WMZ_OK;
}

extern "C" int wmz_bn_act_bwd_reduce(const void* x, const void* y, const void* dy, const float* mean, const float* rstd,
                                     void* g_out, float* sum_g, float* sum_gx, long M, int C, int leaky, float slope,
                                     int dtype, void* stream) {
  WMZ_REQUIRE(dy && M > 0 && C > 0, "wmz_bn_act_bwd_reduce: bad arguments");
  WMZ_REQUIRE(!leaky || y, "wmz_bn_act_bwd_reduce: LeakyReLU backward needs the stored output y");
  WMZ_REQUIRE((sum_g == nullptr) == (sum_gx == nullptr), "wmz_bn_act_bwd_reduce: sum_g and sum_gx go together");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_bn_act_bwd_reduce: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(C, dtype, {x, y, dy, g_out})) {
    const int VW = dtype == WMZ_BF16 ? 8 : 4;
    const long nvec = M * C / VW;
    const int gridv = grid_for(nvec, RED_NT * 4, 256);                 // (a multiple of C / VW vectors per sweep: 256 is)
    if (dtype == WMZ_BF16)
      hipLaunchKernelGGL(bn_act_bwd_reduce_vec_kernel<bf16_t>, dim3(gridv), dim3(RED_NT), red_lds(C, 8), st, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, mean, rstd, (bf16_t*)g_out, sum_g, sum_gx, nvec, C, leaky, slope, nullptr, nullptr);
    else
      hipLaunchKernelGGL(bn_act_bwd_reduce_vec_kernel<float>, dim3(gridv), dim3(RED_NT), red_lds(C, 4), st, (const float*)x, (const float*)y, (const float*)dy, mean, rstd, (float*)g_out, sum_g, sum_gx, nvec, C, leaky, slope, nullptr, nullptr);
    WMZ_LAUNCH_CHECK("wmz_bn_act_bwd_reduce");
    return WMZ_OK;
  }
  dim3 grid((unsigned)grid_for(M, 64, 512), (unsigned)wmz_cdiv(C, 64));
  if (dtype == WMZ_BF16)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, mean, rstd, (bf16_t*)g_out, sum_g, sum_gx, M, C, leaky, slope);
  else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)y, (const float*)dy, mean, rstd, (float*)g_out, sum_g, sum_gx, M, C, leaky, slope);
  WMZ_LAUNCH_CHECK("wmz_bn_act_bwd_reduce");
  return WMZ_OK;
}

extern "C" int wmz_bn_bwd_apply_add(const void* x, const void* g, const float* mean, const float* rstd, const float* gamma,
                                    const float* sum_g, const float* sum_gx, const void* add, void* dx, long M, int C, int dtype,
                                    void* stream) {
  WMZ_REQUIRE(x && g && mean && rstd && gamma && sum_g && sum_gx && dx && M > 0 && C > 0, "wmz_bn_bwd_apply: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_bn_bwd_apply: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(C, dtype, {x, g, dx, add})) {
    const int VW = dtype == WMZ_BF16 ? 8 : 4;
    const long nvec = M * C / VW;
    const int gridv = grid_for(nvec, 256, EW_GRID);
    if (dtype == WMZ_BF16)
      hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<bf16_t>, dim3(gridv), dim3(256), (size_t)6 * C * sizeof(float), st, (const bf16_t*)x, (const bf16_t*)g, mean, rstd, gamma, sum_g, sum_gx, (bf16_t*)dx, nvec, C, 1.f / (float)M, nullptr, nullptr, 1.f, (const bf16_t*)add);
    else
      hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<float>, dim3(gridv), dim3(256), (size_t)6 * C * sizeof(float), st, (const float*)x, (const float*)g, mean, rstd, gamma, sum_g, sum_gx, (float*)dx, nvec, C, 1.f / (float)M, nullptr, nullptr, 1.f, (const float*)add);
    WMZ_LAUNCH_CHECK("wmz_bn_bwd_apply");
    return WMZ_OK;
  }
  const int grid = grid_for(M * C, 256, 4096);
  if (dtype == WMZ_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)g, mean, rstd, gamma, sum_g, sum_gx, (bf16_t*)dx, M, C, (const bf16_t*)add);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)g, mean, rstd, gamma, sum_g, sum_gx, (float*)dx, M, C, (const float*)add);
  WMZ_LAUNCH_CHECK("wmz_bn_bwd_apply");
  return WMZ_OK;
}

extern "C" int wmz_bn_bwd_apply(const void* x, const void* g, const float* mean, const float* rstd, const float* gamma,
                                const float* sum_g, const float* sum_gx, void* dx, long M, int C, int dtype, void* stream) {
  return wmz_bn_bwd_apply_add(x, g, mean, rstd, gamma, sum_g, sum_gx, nullptr, dx, M, C, dtype, stream);
}

extern "C" int wmz_bn_leaky_bwd_supported(int C, int dtype) {
  const int VW = dtype == WMZ_BF16 ? 8 : 4;
  return (dtype == WMZ_F32 || dtype == WMZ_BF16) && C > 0 && C % VW == 0 && 256 % (C / VW) == 0 ? 1 : 0;
}

extern "C" int wmz_bn_leaky_bwd(const void* x, const void* dy, const float* scale, const float* shift, const float* mean,
                                const float* rstd, const float* gamma, float* sum_g, float* sum_gx, const void* add, void* dx,
                                long M, int C, float slope, int dtype, void* stream) {
  WMZ_REQUIRE(x && dy && scale && shift && mean && rstd && gamma && sum_g && sum_gx && dx && M > 0, "wmz_bn_leaky_bwd: bad arguments");
  WMZ_REQUIRE(wmz_bn_leaky_bwd_supported(C, dtype) && vec_ok(C, dtype, {x, dy, dx, add}),
              "wmz_bn_leaky_bwd: C = %d / dtype %d / alignment not built (wmz_bn_leaky_bwd_supported; 16-byte aligned tensors)", C, dtype);
  WMZ_REQUIRE(slope >= 0.f && slope <= 1.f, "wmz_bn_leaky_bwd: LeakyReLU slope in [0, 1] expected");
  hipStream_t st = (hipStream_t)stream;
  const int VW = dtype == WMZ_BF16 ? 8 : 4;
  const long nvec = M * C / VW;
  const int gridr = grid_for(nvec, RED_NT * 4, 256), grida = grid_for(nvec, 256, EW_GRID);
  if (dtype == WMZ_BF16) {
    hipLaunchKernelGGL(bn_act_bwd_reduce_vec_kernel<bf16_t>, dim3(gridr), dim3(RED_NT), red_lds(C, 8), st, (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)dy, mean, rstd, (bf16_t*)nullptr, sum_g, sum_gx, nvec, C, 1, slope, scale, shift);
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<bf16_t>, dim3(grida), dim3(256), (size_t)6 * C * sizeof(float), st, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, gamma, sum_g, sum_gx, (bf16_t*)dx, nvec, C, 1.f / (float)M, scale, shift, slope, (const bf16_t*)add);
  } else {
    hipLaunchKernelGGL(bn_act_bwd_reduce_vec_kernel<float>, dim3(gridr), dim3(RED_NT), red_lds(C, 4), st, (const float*)x, (const float*)nullptr, (const float*)dy, mean, rstd, (float*)nullptr, sum_g, sum_gx, nvec, C, 1, slope, scale, shift);
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<float>, dim3(grida), dim3(256), (size_t)6 * C * sizeof(float), st, (const float*)x, (const float*)dy, mean, rstd, gamma, sum_g, sum_gx, (float*)dx, nvec, C, 1.f / (float)M, scale, shift, slope, (const float*)add);
  }
  WMZ_LAUNCH_CHECK("wmz_bn_leaky_bwd");
  return WMZ_OK;
}

extern "C" int wmz_bilinear2x_nhwc_bwd(const void* dy, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  WMZ_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "wmz_bilinear2x_nhwc_bwd: bad arguments");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_bilinear2x_nhwc_bwd: bad dtype %d", dtype);
  const long total = (long)B * H * W * C;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(C, dtype, {dy, dx})) {
    const int VW = dtype == WMZ_BF16 ? 8 : 4;
    const int gridv = grid_for(total / VW, 256, 8192);
    if (dtype == WMZ_BF16) hipLaunchKernelGGL(bilinear2x_bwd_vec_kernel<bf16_t>, dim3(gridv), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C);
    else hipLaunchKernelGGL(bilinear2x_bwd_vec_kernel<float>, dim3(gridv), dim3(256), 0, st, (const float*)dy, (float*)dx, B, H, W, C);
    WMZ_LAUNCH_CHECK("wmz_bilinear2x_nhwc_bwd");
    return WMZ_OK;
  }
  const int grid = grid_for(total, 256, 8192);
  if (dtype == WMZ_BF16) hipLaunchKernelGGL(bilinear2x_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C);
  else hipLaunchKernelGGL(bilinear2x_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (float*)dx, B, H, W, C);
  WMZ_LAUNCH_CHECK("wmz_bilinear2x_nhwc_bwd");
  return WMZ_OK;
}
