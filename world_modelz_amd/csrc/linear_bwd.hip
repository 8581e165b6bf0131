// Backward pieces of the nn.Linear / PreNorm-LayerNorm family (the reference gets these from autograd):
//   wmz_linear_wgrad    dW[N,K] += dC[M,N]^T . A'[M,K],  dbias[N] += colsum(dC)      A' = A | LN(A) | GELU(A)
//   wmz_layernorm_stats mean / rstd per row
//   wmz_layernorm_bwd   dx = LN'(x)^T dyhat (+ skip gradient),  dgamma, dbeta
// (the data gradient dA' = dC . W is wmz_linear_fwd on the transposed weight, optionally x gelu'(z).)
#include "wmz_common.h"
#include "wmz_internal.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float gelu_erf(float v) { return wmz_gelu(v); }

template <typename T> __device__ __forceinline__ void unpack_chunk(const i32x4& c, float* f);
template <> __device__ __forceinline__ void unpack_chunk<float>(const i32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __int_as_float(c[i]);
}
template <> __device__ __forceinline__ void unpack_chunk<bf16_t>(const i32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(((unsigned)c[i]) << 16);
    f[2 * i + 1] = __uint_as_float(((unsigned)c[i]) & 0xFFFF0000u);
  }
}
template <typename T> __device__ __forceinline__ i32x4 pack_chunk(const float* f);
template <> __device__ __forceinline__ i32x4 pack_chunk<float>(const float* f) {
  i32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = __float_as_int(f[i]);
  return c;
}
template <> __device__ __forceinline__ i32x4 pack_chunk<bf16_t>(const float* f) {
  i32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    c[i] = (int)((unsigned)f32_to_bf16_bits(f[2 * i]) | ((unsigned)f32_to_bf16_bits(f[2 * i + 1]) << 16));
  return c;
}

// ------------------------------------------------------------------------------------------------ wgrad
// Output tile 128 (n) x 128 (k') per workgroup, reduction over a slice of M in steps of 32 rows.  Both operands are
// "m-major" in memory, so both tiles are staged row = m and read as TRANSPOSED fragments (8 consecutive m of one column):
// bf16 through ds_read_b64_tr_b16 from a 64-byte-granule swizzled image, fp32 as scalar reads.
struct WgParams {
  const void* dC; long ldc;
  const void* A; long lda;
  float* dW; float* dbias;
  int M, N, K;
  const float* gamma; const float* beta; const float* mean; const float* rstd;
  int gelu_in;
  int a_tiled;      // A is the fused path's TILED stream (bf16, K = 256): per 32-row tile [16 chunks][2 halves][32 rows][8]
  int nbn, nbk, rows_per_wg, nsplit;
  int direct;       // nsplit == 1 and nobody else writes the tile: dW (+)= acc straight from the registers (1: +=, 2: =), no workspace
  // PRO == 3: A is the implicit im2col of an NHWC tensor x[B,Hi,Wi,Cin] (row m = output pixel, k = (kh, kw, ci))
  int Hi, Wi, Cin, Ho, Wo, KW, cstride, cpad;
};
// several independent weight gradients in ONE launch pair (wgrad2_kernel + wgrad_reduce_kernel): the workgroups of problem
// i are first[i] .. first[i+1]-1, its partial tiles live at workspace + wsoff[i]
constexpr int WG_MAXB = 6;
struct WgBatch { WgParams p[WG_MAXB]; int first[WG_MAXB + 1]; long wsoff[WG_MAXB]; int n; };
struct RedProb {
  float* dW; float* dbias; long wsoff, NK; int nsplit, N, nblk_w, overwrite, first;
  // conv weight gradients stored in nn.Conv2d's layout: element (n, k = tap * cin_p + c) of the [N, K] result goes to
  // dW[(n * ci + c) * taps + tap] when n < co and c < ci (channel padding cropped); taps == 0: plain [N, K]
  int taps, cin_p, co, ci;
};
struct RedBatch { RedProb p[WG_MAXB]; int n; };

constexpr int WG_BN = 128, WG_BK = 128, WG_MS = 32;

template <typename T> __device__ __forceinline__ int wswz(int r) { return sizeof(T) == 2 ? ((r & 3) << 6) : 0; }

// transposed 32x32x16 operand: element j = img[m0 + 8*(lane>>5) + j][c0 + (lane&31)]
template <typename T>
__device__ __forceinline__ void wg_col_frag(Frag8<T>& f, const char* img, int m0, int c0, int lane) {
  constexpr int ROWB = WG_BN * (int)sizeof(T);
  if constexpr (sizeof(T) == 2) {
    const int gi = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int ra = m0 + 8 * (gi >> 1) + q, rb = ra + 4;
    const int cb = (c0 + 16 * (gi & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + ra * ROWB + (cb ^ wswz<T>(ra))));
    const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + rb * ROWB + (cb ^ wswz<T>(rb))));
    f.v = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    const int h = lane >> 5, c = c0 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = *reinterpret_cast<const float*>(img + (m0 + 8 * h + j) * ROWB + c * 4);
  }
}

template <typename T, int PRO>   // PRO: 0 raw A, 1 LN(A) from mean/rstd/gamma/beta, 2 GELU(A), 3 implicit im2col of NHWC x
__global__ __launch_bounds__(NT, 2) void wgrad_kernel(WgParams P) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int ROWB = WG_BN * (int)sizeof(T);
  constexpr int CPR = ROWB / 16;                       // chunks per tile row: 16 (bf16) / 32 (f32)
  constexpr int PER_T = WG_MS * CPR / NT;              // chunks per thread per tile: 2 / 4
  __shared__ __attribute__((aligned(16))) char Cs[WG_MS * ROWB];
  __shared__ __attribute__((aligned(16))) char As[WG_MS * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = xcd_remap(blockIdx.x, gridDim.x);          // the tiles of one slice of M share an XCD: its L2 serves the operands' re-reads
  const int bk = bid % P.nbk; bid /= P.nbk;
  const int bn = bid % P.nbn; bid /= P.nbn;
  const int split = bid;
  const int n0 = bn * WG_BN, k0 = bk * WG_BK;
  const int m_begin = split * P.rows_per_wg, m_end = min(P.M, m_begin + P.rows_per_wg);
  const T* dC = reinterpret_cast<const T*>(P.dC);
  const T* A = reinterpret_cast<const T*>(P.A);

  i32x4 rc[PER_T], ra[PER_T];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int it = 0; it < PER_T; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CPR, c = idx - r * CPR;
      const int m = m0 + r;
      rc[it] = (i32x4)(0);
      ra[it] = (i32x4)(0);
      if (m < m_end) {
        if (n0 + c * EPC < P.N) rc[it] = *reinterpret_cast<const i32x4*>(dC + (long)m * P.ldc + n0 + c * EPC);
        if constexpr (PRO == 3) {
          const int k = k0 + c * EPC;
          if (k < P.K) {
            const int tap = k / P.Cin, ci = k - tap * P.Cin;
            const int kh = tap / P.KW, kw = tap - kh * P.KW;
            const int wo = m % P.Wo, t = m / P.Wo;
            const int ho = t % P.Ho, b = t / P.Ho;
            const int hi = ho * P.cstride - P.cpad + kh, wi = wo * P.cstride - P.cpad + kw;
            if (hi >= 0 && hi < P.Hi && wi >= 0 && wi < P.Wi)
              ra[it] = *reinterpret_cast<const i32x4*>(A + (((long)b * P.Hi + hi) * P.Wi + wi) * P.Cin + ci);
          }
        } else if (k0 + c * EPC < P.K) {
          i32x4 v = *reinterpret_cast<const i32x4*>(A + (long)m * P.lda + k0 + c * EPC);
          if constexpr (PRO == 1 || PRO == 2) {
            float f[EPC];
            unpack_chunk<T>(v, f);
            if constexpr (PRO == 1) {
              const float mu = P.mean[m], rs = P.rstd[m];
#pragma unroll
              for (int e = 0; e < EPC; ++e) f[e] = (f[e] - mu) * rs * P.gamma[k0 + c * EPC + e] + P.beta[k0 + c * EPC + e];
            } else {
#pragma unroll
              for (int e = 0; e < EPC; ++e) f[e] = gelu_erf(f[e]);
            }
            v = pack_chunk<T>(f);
          }
          ra[it] = v;
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
  float bsum = 0.f;                                    // threads 0..127 of the bk == 0 workgroups: column sums of dC
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;

  if (m_begin < m_end) fetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += WG_MS) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PER_T; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CPR, c = idx - r * CPR;
      const int off = r * ROWB + ((c << 4) ^ wswz<T>(r));
      *reinterpret_cast<i32x4*>(Cs + off) = rc[it];
      *reinterpret_cast<i32x4*>(As + off) = ra[it];
    }
    __syncthreads();
    if (m0 + WG_MS < m_end) fetch(m0 + WG_MS);
    if (P.dbias != nullptr && bk == 0 && tid < WG_BN) {
#pragma unroll 8
      for (int r = 0; r < WG_MS; ++r) {
        const char* p = Cs + r * ROWB + ((tid * (int)sizeof(T)) ^ wswz<T>(r));
        bsum += Elem<T>::to_f32(*reinterpret_cast<const T*>(p));
      }
    }
#pragma unroll
    for (int ms = 0; ms < WG_MS; ms += 16) {
      Frag8<T> cf[2], af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wg_col_frag<T>(cf[i], Cs, ms, wn + 32 * i, lane);
        wg_col_frag<T>(af[i], As, ms, wk + 32 * i, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], cf[i], af[j]);
    }
  }

  // D: row (n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (k') = lane&31
  const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + wk + 32 * j + l31;
      if (kc >= P.K) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wn + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (n < P.N) atomicAdd(P.dW + (long)n * P.K + kc, acc[i][j][reg]);
      }
    }
  if (P.dbias != nullptr && bk == 0 && tid < WG_BN && n0 + tid < P.N) atomicAdd(P.dbias + n0 + tid, bsum);
}


// ---- wgrad, version 2: the same 128 x 128 output tile and operand handling, but
//   * slabs of 64 rows (bf16; 32 for fp32) in a DOUBLE-buffered LDS image: one barrier per slab and 16 MFMAs per wave
//     between two barriers (version 1: two barriers around 8 MFMAs);
//   * the slices of M do not meet in float atomics (version 1: 64 KB of same-address atomics per workgroup, all at the end,
//     serialised memory-side) but in a workspace [split][N][K] (+ [split][N] for the bias) that wgrad_reduce_kernel sums
//     into dW / dbias: two-stage reduction, deterministic summation order.
template <typename T> struct W2 { static constexpr int MS = sizeof(T) == 2 ? 64 : 32; };

template <typename T, int PRO>
__global__ __launch_bounds__(NT, 2) void wgrad2_kernel(WgBatch B, float* __restrict__ ws_base) {
  int pi = 0;                                            // this workgroup's problem (uniform)
#pragma unroll
  for (int i = 1; i < WG_MAXB; ++i) if (i < B.n && (int)blockIdx.x >= B.first[i]) pi = i;
  const WgParams P = B.p[pi];
  float* __restrict__ ws = ws_base + B.wsoff[pi];
  const int wg0 = B.first[pi], nwg = B.first[pi + 1] - B.first[pi];
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int ROWB = WG_BN * (int)sizeof(T);
  constexpr int CPR = ROWB / 16;
  constexpr int MS = W2<T>::MS;
  constexpr int PER_T = MS * CPR / NT;                 // 4
  constexpr int IMG = MS * ROWB;                       // 16 KB
  __shared__ __attribute__((aligned(16))) char smem[4 * IMG];   // [buffer][dC image | A image]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = xcd_remap((int)blockIdx.x - wg0, nwg);         // the tiles of one slice of M share an XCD: its L2 serves the operands' re-reads
  const int bk = bid % P.nbk; bid /= P.nbk;
  const int bn = bid % P.nbn; bid /= P.nbn;
  const int split = bid;
  const int n0 = bn * WG_BN, k0 = bk * WG_BK;
  const int m_begin = split * P.rows_per_wg, m_end = min(P.M, m_begin + P.rows_per_wg);
  const T* dC = reinterpret_cast<const T*>(P.dC);
  const T* A = reinterpret_cast<const T*>(P.A);

  // two register sets: the loads of slabs s + 1 and s + 2 are in flight while slab s is multiplied (one slab ahead the
  // loop runs at one memory round trip per slab: ~3k cycles for 16 MFMAs)
  i32x4 rcs[2][PER_T], ras[2][PER_T];
  float lmu[2][PER_T], lrs[2][PER_T];                  // LayerNorm statistics of the rows in flight (PRO 1)
  auto fetch = [&](auto setc, int m0) {
    constexpr int S = decltype(setc)::value;
    i32x4 (&rc)[PER_T] = rcs[S];
    i32x4 (&ra)[PER_T] = ras[S];
#pragma unroll
    for (int it = 0; it < PER_T; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CPR, c = idx - r * CPR;
      const int m = m0 + r;
      rc[it] = (i32x4)(0);
      ra[it] = (i32x4)(0);
      if constexpr (PRO == 1) { lmu[S][it] = 0.f; lrs[S][it] = 0.f; }
      if (m < m_end) {
        if (n0 + c * EPC < P.N) rc[it] = *reinterpret_cast<const i32x4*>(dC + (long)m * P.ldc + n0 + c * EPC);
        if constexpr (PRO == 3) {
          // implicit im2col of an NHWC tensor: row m = output pixel, k = (kh, kw, ci); taps outside the image are zero
          const int k = k0 + c * EPC;
          if (k < P.K) {
            const int tap = k / P.Cin, ci = k - tap * P.Cin;
            const int kh = tap / P.KW, kw = tap - kh * P.KW;
            const int wo = m % P.Wo, t = m / P.Wo;
            const int ho = t % P.Ho, b = t / P.Ho;
            const int hi = ho * P.cstride - P.cpad + kh, wi = wo * P.cstride - P.cpad + kw;
            if (hi >= 0 && hi < P.Hi && wi >= 0 && wi < P.Wi)
              ra[it] = *reinterpret_cast<const i32x4*>(A + (((long)b * P.Hi + hi) * P.Wi + wi) * P.Cin + ci);
          }
        } else if (k0 + c * EPC < P.K) {
          // (tiled: 16-byte chunk cg = 16 h + s of row m sits at ((2 s + h) * 32 + m % 32) * 16 bytes of its 32-row tile:
          //  a slab's rows of one chunk column are one contiguous 512-byte run)
          const int cg = (k0 >> 3) + c;
          const long aoff = P.a_tiled ? (long)(m >> 5) * (32 * 256) + ((((2 * (cg & 15) + (cg >> 4)) << 5) + (m & 31)) << 3)
                                      : (long)m * P.lda + k0 + c * EPC;
          ra[it] = *reinterpret_cast<const i32x4*>(A + aoff);
          if constexpr (PRO == 1) { if (P.mean != nullptr) { lmu[S][it] = P.mean[m]; lrs[S][it] = P.rstd[m]; } }
        }
      }
    }
  };
  // a thread always lands on the same chunk column (NT is a multiple of CPR): its LayerNorm affine lives in registers
  float gam[EPC], bet[EPC];
  // (a batch may mix problems with and without the LayerNorm prologue: the PRO == 1 instantiation then runs them all, the
  //  plain ones -- P.mean == NULL, uniform per workgroup -- skipping the arithmetic)
  const bool has_ln = PRO == 1 && P.mean != nullptr;
  if constexpr (PRO == 1) {
    const int kc0 = k0 + (tid % CPR) * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      gam[e] = (has_ln && kc0 + e < P.K) ? P.gamma[kc0 + e] : 0.f;
      bet[e] = (has_ln && kc0 + e < P.K) ? P.beta[kc0 + e] : 0.f;
    }
  }
  // the prologue arithmetic runs when the slab goes to LDS, not when it is requested: the loads stay in flight
  auto stash = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
    char* Cs = smem + buf * 2 * IMG;
    char* As = Cs + IMG;
#pragma unroll
    for (int it = 0; it < PER_T; ++it) {
      const int idx = tid + it * NT;
      const int r = idx / CPR, c = idx - r * CPR;
      const int off = r * ROWB + ((c << 4) ^ wswz<T>(r));
      i32x4 v = ras[S][it];
      if constexpr (PRO == 1 || PRO == 2) {
        if (k0 + c * EPC < P.K && (PRO != 1 || has_ln)) {
          float f[EPC];
          unpack_chunk<T>(v, f);
          if constexpr (PRO == 1) {
            const float rs = lrs[S][it], mr = -lmu[S][it] * rs;
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = fmaf(fmaf(f[e], rs, mr), gam[e], bet[e]);
          } else {
            // bf16: the fitted GELU of the fused forward (2.6e-5 abs from the erf form, two orders below bf16 resolution)
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = sizeof(T) == 2 ? wmz_gelu_fast(f[e]) : gelu_erf(f[e]);
          }
          v = pack_chunk<T>(f);
        }
      }
      *reinterpret_cast<i32x4*>(Cs + off) = rcs[S][it];
      *reinterpret_cast<i32x4*>(As + off) = v;
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
  float bsum = 0.f;                                    // threads 0..127 of the bk == 0 workgroups: column sums of dC
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;

  auto compute = [&](int cur) {
    const char* Cs = smem + cur * 2 * IMG;
    const char* As = Cs + IMG;
    if (P.dbias != nullptr && bk == 0 && tid < WG_BN) {
#pragma unroll 8
      for (int r = 0; r < MS; ++r) {
        const char* p = Cs + r * ROWB + ((tid * (int)sizeof(T)) ^ wswz<T>(r));
        bsum += Elem<T>::to_f32(*reinterpret_cast<const T*>(p));
      }
    }
#pragma unroll
    for (int ms = 0; ms < MS; ms += 16) {
      Frag8<T> cf[2], af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wg_col_frag<T>(cf[i], Cs, ms, wn + 32 * i, lane);
        wg_col_frag<T>(af[i], As, ms, wk + 32 * i, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], cf[i], af[j]);
    }
  };
  // slab s lives in LDS buffer s & 1; register set 0 carries the odd slabs on their way in, set 1 the even ones
  if (m_begin < m_end) {
    fetch(S1{}, m_begin);
    fetch(S0{}, m_begin + MS);                         // (rows past m_end come back as zeros)
    stash(S1{}, 0);
    fetch(S1{}, m_begin + 2 * MS);
  }
  __syncthreads();
  for (int m0 = m_begin; m0 < m_end; m0 += 2 * MS) {
    compute(0);                                        // slab s (even)
    stash(S0{}, 1);                                    // slab s + 1 -> buffer 1 (last read before the previous barrier)
    fetch(S0{}, m0 + 3 * MS);
    __syncthreads();
    if (m0 + MS < m_end) compute(1);                   // slab s + 1
    stash(S1{}, 0);                                    // slab s + 2
    fetch(S1{}, m0 + 4 * MS);
    __syncthreads();
  }

  // D: row (n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (k') = lane&31
  const int l31 = lane & 31, hh = lane >> 5;
  // workspace layout [256-float block of dW][slice][256]: the reduction then streams 64 KB runs instead of gathering one KB
  // from each of the slices 256 KB apart (same HBM channel, a new DRAM row every time)
  const long nblk256 = ((long)P.N * P.K + 255) >> 8;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + wk + 32 * j + l31;
      if (kc >= P.K) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wn + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (n < P.N) {
          const long q = (long)n * P.K + kc;
          if (P.direct == 0) ws[(((q >> 8) * P.nsplit + split) << 8) + (q & 255)] = acc[i][j][reg];
          else if (P.direct == 1) P.dW[q] += acc[i][j][reg];          // (the tile's only writer)
          else P.dW[q] = acc[i][j][reg];
        }
      }
    }
  if (P.dbias != nullptr && bk == 0 && tid < WG_BN && n0 + tid < P.N) {
    if (P.direct == 0) ws[nblk256 * P.nsplit * 256 + (long)split * P.N + n0 + tid] = bsum;
    else if (P.direct == 1) P.dbias[n0 + tid] += bsum;
    else P.dbias[n0 + tid] = bsum;
  }
}

// ---- wgrad, version 3 (bf16, plain operands): a 256 x 256 output tile per 8-wave workgroup, each wave 64 (n) x 128 (k').
// Version 2 runs at 13 % of the MFMA rate at config 4 for two reasons this tile removes: (i) a 64 x 64 wave tile reads one
// 1 KB LDS fragment per MFMA and waits for it -- the chain read -> wait -> MFMA bounds a wave, and at two waves per SIMD there
// is nobody to fill the gaps; with 64 x 128 wave tiles six fragments feed eight MFMAs; (ii) every operand row is re-read from
// L2 once per 128-wide output tile in the other dimension (536 MB per layer's launch against 300 MB of operands); at the
// default widths (N, K <= 256) a problem is ONE tile and each operand element enters the CU exactly once.
// Slabs of 32 rows travel global -> LDS by DMA (global_load_lds) into a ring of W3_NBUF buffers of 32 KB (dC image | A image),
// NBUF - 1 of them in flight or in use ahead of the multiply: no staging registers, no ds_write, and the operand stream is
// two to three slab-times ahead of its use (round 4: with ONE register set in flight a slab had half a multiply phase, ~0.4 us,
// to cross a ~1.5 us round trip -- the ablations added up, 88 us of streaming + 92 us of multiply = 181 us at dim 384).  LDS reads
// are inline asm with hand-counted waits (hipcc drains the DMA ring in front of every LDS read it can see); fragments of the
// next 16-row step are read while the MFMAs of this one run, across slab boundaries too.  The partial tiles go through the same
// workspace layout and reduction kernel as version 2's.
constexpr int W3_NT = 512, W3_T = 256, W3_MS = 64, W3_ROWB = W3_T * 2;
constexpr int W3_SL = 32, W3_SIMG = W3_SL * W3_ROWB, W3_BUF = 2 * W3_SIMG;     // 16 KB per operand image, 32 KB per ring buffer
#ifndef WMZ_W3_NBUF
#define WMZ_W3_NBUF 4
#endif
constexpr int W3_NBUF = WMZ_W3_NBUF;
#ifndef WMZ_W3_ABL            // timing ablations (tools/build_variant.py): 1 = no operand loads, 2 = no MFMA loops, 4 = no tile stores, 8 = no slab barriers
#define WMZ_W3_ABL 0
#endif
typedef const __attribute__((address_space(1))) void* w3_gptr_t;
typedef __attribute__((address_space(3))) void* w3_lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned w3_u32x2;

struct W3Set { s16x4 h[12]; };      // one 16-row step's operands: halves 2i, 2i+1 of dC fragment i (2), then of A fragment j (4)

// the step's twelve transposed reads (MS: first row of the step within the slab); nothing may touch the set before w3_wait
template <int MS>
__device__ __forceinline__ void w3_read_set(W3Set& S, unsigned base, const unsigned (&ca)[2], const unsigned (&aa)[4]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    S.h[2 * i] = ds_read_tr16_asm<MS * W3_ROWB>(base + ca[i]);
    S.h[2 * i + 1] = ds_read_tr16_asm<(MS + 4) * W3_ROWB>(base + ca[i]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    S.h[4 + 2 * j] = ds_read_tr16_asm<MS * W3_ROWB>(base + aa[j]);
    S.h[5 + 2 * j] = ds_read_tr16_asm<(MS + 4) * W3_ROWB>(base + aa[j]);
  }
}
__device__ __forceinline__ void w3_wait(W3Set& S) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(S.h[0]), "+v"(S.h[1]), "+v"(S.h[2]), "+v"(S.h[3]), "+v"(S.h[4]), "+v"(S.h[5]), "+v"(S.h[6]), "+v"(S.h[7]),
                 "+v"(S.h[8]), "+v"(S.h[9]), "+v"(S.h[10]), "+v"(S.h[11]) :: "memory");
}
__device__ __forceinline__ void w3_mma(f32x16 (&acc)[2][4], const W3Set& S) {
  if (WMZ_W3_ABL & 2) return;
  Frag8<bf16_t> cf[2], af[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) cf[i].v = __builtin_shufflevector(S.h[2 * i], S.h[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int j = 0; j < 4; ++j) af[j].v = __builtin_shufflevector(S.h[4 + 2 * j], S.h[5 + 2 * j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) mma32(acc[i][j], cf[i], af[j]);
}
template <int OFF>
__device__ __forceinline__ w3_u32x2 w3_read_b64(unsigned addr) {
  w3_u32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
// One 16-row step: its eight MFMAs on `cur`, with the twelve reads of the step after it slotted into the first six gaps (fragment
// g's two halves behind MFMA g; MS: that step's first row within the buffer at nbase) and four of the slab's bias-sum reads
// into the last two; dma(0), dma(1): the caller's two DMA instructions of this step, behind MFMAs 3 and 7 (a global_load_lds
// holds its wave's issue for 60-180 cycles: four in a burst behind the barrier, in both waves of a SIMD at once, idle the matrix
// pipe ~360 cycles per slab).  A wave so keeps the matrix pipe fed by itself: read bursts of the two waves of a SIMD fall in phase behind
// every barrier and leave the pipe idle for their length (bare loop, bursts: 67 % MFMA-busy).
#ifndef WMZ_W3_RGAPS
#define WMZ_W3_RGAPS 6
#endif
constexpr int W3_RGAPS = WMZ_W3_RGAPS;       // the twelve reads go into the first RGAPS gaps (12 / RGAPS each)
template <int MS, int KB, typename DMA>
__device__ __forceinline__ void w3_mma_read(f32x16 (&acc)[2][4], const W3Set& cur, W3Set& nxt, unsigned nbase,
                                            const unsigned (&ca)[2], const unsigned (&aa)[4], bool do_bias, w3_u32x2 (&bv)[4],
                                            unsigned baddr, DMA&& dma) {
  Frag8<bf16_t> cf[2], af[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) cf[i].v = __builtin_shufflevector(cur.h[2 * i], cur.h[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int j = 0; j < 4; ++j) af[j].v = __builtin_shufflevector(cur.h[4 + 2 * j], cur.h[5 + 2 * j], 0, 1, 2, 3, 4, 5, 6, 7);
  static_for<8>([&](auto G) {
    constexpr int g = G, i = g >> 2, j = g & 3;
    if (!(WMZ_W3_ABL & 2)) mma32(acc[i][j], cf[i], af[j]);
    if constexpr (g < W3_RGAPS) {
      static_for<12 / W3_RGAPS>([&](auto R) {
        constexpr int h = g * (12 / W3_RGAPS) + R, f = h >> 1;
        unsigned a = nbase;
        if constexpr (f < 2) a += ca[f]; else a += aa[f - 2];
        nxt.h[h] = ds_read_tr16_asm<(MS + 4 * (h & 1)) * W3_ROWB>(a);
      });
    }
    if constexpr (g >= 6) {
      if (do_bias) {
        constexpr int k = 2 * (g - 6);
        bv[k] = w3_read_b64<(KB + k) * 4 * W3_ROWB>(baddr);
        bv[k + 1] = w3_read_b64<(KB + k + 1) * 4 * W3_ROWB>(baddr);
      }
    }
    if constexpr ((g & 3) == 3) dma(std::integral_constant<int, (g >> 2)>{});      // behind MFMAs 3 and 7: one DMA instruction each
    __builtin_amdgcn_sched_barrier(0);
  });
}
// this wave's own DMA instructions of every slab but the `younger` most recent ones have landed (four instructions per slab)
__device__ __forceinline__ void w3_wait_dma(int younger) {
  if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
#ifdef WMZ_W3_STAMPS          // diagnostic build (tools/build_variant.py -DWMZ_W3_STAMPS): s_memtime stamps of wave 0 of every workgroup
__device__ unsigned long long w3_stamps[1024 * 8];
#define W3_STAMP(i) do { if (tid == 0) w3_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W3_PARK_BEGIN() unsigned long long park_t0 = __builtin_amdgcn_s_memtime()
#define W3_PARK_END() do { park_sum += __builtin_amdgcn_s_memtime() - park_t0; } while (0)
#else
#define W3_STAMP(i) do {} while (0)
#define W3_PARK_BEGIN() do {} while (0)
#define W3_PARK_END() do {} while (0)
#endif
static_assert(W3_NBUF >= 3 && W3_NBUF <= 6, "the DMA ring: 3 .. 6 buffers (w3_wait_dma counts up to three younger slabs)");

__global__ __launch_bounds__(W3_NT, 1) void wgrad3_kernel(WgBatch B, float* __restrict__ ws_base) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXB; ++i) if (i < B.n && (int)blockIdx.x >= B.first[i]) pi = i;
  const WgParams P = B.p[pi];
  float* __restrict__ ws = ws_base + B.wsoff[pi];
  const int wg0 = B.first[pi], nwg = B.first[pi + 1] - B.first[pi];
  __shared__ __attribute__((aligned(1024))) char w3_smem[W3_NBUF * W3_BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = xcd_remap((int)blockIdx.x - wg0, nwg);
  const int bk = bid % P.nbk; bid /= P.nbk;
  const int bn = bid % P.nbn; bid /= P.nbn;
  const int split = bid;
  const int n0 = bn * W3_T, k0 = bk * W3_T;
  const int m_begin = split * P.rows_per_wg, m_end = min(P.M, m_begin + P.rows_per_wg);
  const int ns = m_begin < m_end ? (m_end - m_begin + W3_SL - 1) / W3_SL : 0;
  const int rem = m_end - m_begin - (ns - 1) * W3_SL;                    // rows of the last slab
  const bf16_t* dC = reinterpret_cast<const bf16_t*>(P.dC);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);

  // ---- DMA: wave w brings rows 8 (w & 3) .. + 7 of operand w >> 2 (0: dC, 1: A) -- four instructions of two 512-byte image
  // rows each; lane (half, pc) lands on physical chunk pc of row r and therefore FETCHES logical chunk pc ^ ((r & 3) << 2) (the
  // image's 64-byte XOR swizzle).  Columns past N / K are fetched from the tile's first chunk instead (finite values in output
  // columns nobody stores); rows past m_end from row m_end - 1, and the dC rows are zeroed in LDS afterwards.
  const int op = wave >> 2, r0 = 8 * (wave & 3) + (lane >> 5), pc = lane & 31;
  auto src_of = [&](int m, int r) -> const char* {
    int lc = pc ^ ((r & 3) << 2);
    if (op == 0) {
      if (n0 + lc * 8 >= P.N) lc = 0;
      return reinterpret_cast<const char*>(dC + (long)m * P.ldc + n0 + lc * 8);
    }
    if (k0 + lc * 8 >= P.K) lc = 0;
    if (P.a_tiled) {
      const int cg = (k0 >> 3) + lc;                                      // 16-byte chunk index of the 256-wide row
      return reinterpret_cast<const char*>(A + (long)(m >> 5) * (32 * 256) + ((((2 * (cg & 15) + (cg >> 4)) << 5) + (m & 31)) << 3));
    }
    return reinterpret_cast<const char*>(A + (long)m * P.lda + k0 + lc * 8);
  };
  // rows r0, r0 + 2 (their swizzles differ) carry a pointer each; rows r0 + 4, r0 + 6 sit a wave-uniform step behind them
  const int mlast = max(m_end - 1, 0);
  const char* sp[2] = {src_of(min(m_begin + r0, mlast), r0), src_of(min(m_begin + r0 + 2, mlast), r0 + 2)};
  const long slab_stride = op == 0 ? (long)W3_SL * P.ldc * 2 : (P.a_tiled ? (long)32 * 256 * 2 : (long)W3_SL * P.lda * 2);
  const long step4 = op == 0 ? (long)4 * P.ldc * 2 : (P.a_tiled ? (long)4 * 8 * 2 : (long)4 * P.lda * 2);
  char* const dst0 = w3_smem + op * W3_SIMG + (wave & 3) * 4096;
  int ibuf = 0;                                                           // ring buffer of the slab being issued
  // instruction i (0..3) of slab s; the slab's four go out in order, not necessarily together (see the slab loop)
  auto issue_part = [&](int s, auto I) {
    constexpr int i = I;
    char* dst = dst0 + ibuf * W3_BUF + i * 1024;
    if (!(WMZ_W3_ABL & 1)) {
      const int m0 = m_begin + s * W3_SL;
      if (m0 + W3_SL <= m_end) {
        __builtin_amdgcn_global_load_lds((w3_gptr_t)(sp[i & 1] + (i >> 1) * step4), (w3_lptr_t)dst, 16, 0, 0);
      } else {
        const int r = r0 + 2 * i;
        __builtin_amdgcn_global_load_lds((w3_gptr_t)src_of(min(m0 + r, m_end - 1), r), (w3_lptr_t)dst, 16, 0, 0);
      }
    }
    if constexpr (i == 3) {
      ibuf = ibuf + 1 == W3_NBUF ? 0 : ibuf + 1;
      sp[0] += slab_stride;
      sp[1] += slab_stride;
    }
  };
  auto issue = [&](int s) { static_for<4>([&](auto I) { issue_part(s, I); }); };
  // rows rem .. 31 of the dC image of the last slab (in buffer `buf`) := 0, between two barriers
  auto zero_tail = [&](int buf) {
    char* Cs = w3_smem + buf * W3_BUF;
    for (int idx = tid; idx < (W3_SL - rem) * 32; idx += W3_NT)
      *reinterpret_cast<i32x4*>(Cs + (rem + (idx >> 5)) * W3_ROWB + (idx & 31) * 16) = (i32x4)(0);
    __syncthreads();
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x16)(0.f);
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 128;
  // fragment addresses within a ring buffer (w3_col_frag's addressing: row 8 (gi >> 1) + q (+ 4), the swizzle is q << 6)
  unsigned ca[2], aa[4];
  {
    const int gi = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const unsigned rowp = (8 * (gi >> 1) + q) * W3_ROWB, x = (16 * (gi & 1) + 4 * p) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) ca[i] = rowp + ((unsigned)((wn + 32 * i) * 2) ^ (unsigned)(q << 6)) + x;
#pragma unroll
    for (int j = 0; j < 4; ++j) aa[j] = W3_SIMG + rowp + ((unsigned)((wk + 32 * j) * 2) ^ (unsigned)(q << 6)) + x;
  }
  const unsigned smem0 = lds_addr(w3_smem);
  // column sums of dC (threads 0..255 of the bk == 0 workgroups): thread (g = tid >> 6, c4 = tid & 63) sums columns 4 c4 .. + 3
  // over the rows r = g (mod 4) -- one 8-byte read per row; the four row classes meet in LDS at the end
  const bool do_bias = P.dbias != nullptr && bk == 0 && tid < W3_T;
  float bsum4[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned bias_a = (unsigned)((tid >> 6) & 3) * W3_ROWB + ((unsigned)((tid & 63) * 8) ^ (unsigned)(((tid >> 6) & 3) << 6));

  unsigned long long park_sum = 0;
  W3_STAMP(0);
  const int npro = min(ns, W3_NBUF - 1);
  for (int s = 0; s < npro; ++s) issue(s);
  W3Set S0, S1;
  W3_STAMP(1);
  if (ns > 0) {
    w3_wait_dma(npro - 1);
    __builtin_amdgcn_s_barrier();
    if (ns == 1 && rem < W3_SL) zero_tail(0);
    w3_read_set<0>(S0, smem0, ca, aa);
    w3_wait(S0);
  }
  W3_STAMP(2);
  int cb = 0;                                                             // ring buffer of the slab being multiplied
  for (int s = 0; s < ns; ++s) {
    const unsigned base = smem0 + cb * W3_BUF;
    const int nb = cb + 1 == W3_NBUF ? 0 : cb + 1;
    w3_u32x2 bv[4];                                                       // bias-sum reads: rows g + 4 k, k < 4 with the first step, the rest with the second
    auto bias_add = [&]() {
      asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bsum4[0] += __uint_as_float(bv[k][0] << 16); bsum4[1] += __uint_as_float(bv[k][0] & 0xffff0000u);
        bsum4[2] += __uint_as_float(bv[k][1] << 16); bsum4[3] += __uint_as_float(bv[k][1] & 0xffff0000u);
      }
    };
    // rows 0..15 | reads rows 16..31 | the second half of slab s + NBUF - 2's DMA (its first half went out behind the last barrier)
    const bool dma_a = s >= 1 && s + W3_NBUF - 2 < ns;
    w3_mma_read<16, 0>(acc, S0, S1, base, ca, aa, do_bias, bv, base + bias_a, [&](auto K) {
      if (dma_a) issue_part(s + W3_NBUF - 2, std::integral_constant<int, 2 + decltype(K)::value>{});
    });
    w3_wait(S1);
    if (do_bias) bias_add();
    if (s + 1 < ns) {
      // slab s + 1 becomes visible; behind this barrier nobody reads slab s - 1 any more: its buffer takes slab s + NBUF - 1
      W3_PARK_BEGIN();
      w3_wait_dma(min(W3_NBUF - 3, ns - s - 2));
      if (!(WMZ_W3_ABL & 8)) __builtin_amdgcn_s_barrier();
      W3_PARK_END();
      if (s + 2 == ns && rem < W3_SL) zero_tail(nb);
    }
    // rows 16..31 | reads rows 0..15 of slab s + 1 (behind the last slab: of a buffer nobody needs -- the values are dropped) |
    // the first half of slab s + NBUF - 1's DMA into the buffer the barrier has just freed
    const bool dma_b = s + W3_NBUF - 1 < ns;
    w3_mma_read<0, 4>(acc, S1, S0, smem0 + nb * W3_BUF, ca, aa, do_bias, bv, base + bias_a, [&](auto K) {
      if (dma_b) issue_part(s + W3_NBUF - 1, K);
    });
    w3_wait(S0);
    if (do_bias) bias_add();
    cb = nb;
  }

  W3_STAMP(3);
#ifdef WMZ_W3_STAMPS
  if (tid == 0) w3_stamps[blockIdx.x * 8 + 5] = park_sum;
#endif
  // D: row (n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (k') = lane&31
  const int l31 = lane & 31, hh = lane >> 5;
  const long nblk256 = ((long)P.N * P.K + 255) >> 8;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kc = k0 + wk + 32 * j + l31;
      if (kc >= P.K) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wn + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (n < P.N && (!(WMZ_W3_ABL & 4) || P.M < 0)) {
          const long q = (long)n * P.K + kc;
          ws[(((q >> 8) * P.nsplit + split) << 8) + (q & 255)] = acc[i][j][reg];
        }
      }
    }
  if (P.dbias != nullptr && bk == 0) {                 // the four row classes of a column meet in LDS (the ring is done with)
    __syncthreads();
    float* red = reinterpret_cast<float*>(w3_smem);
    if (tid < W3_T) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(tid >> 6) * W3_T + 4 * (tid & 63) + e] = bsum4[e];
    }
    __syncthreads();
    if (tid < W3_T && n0 + tid < P.N)
      ws[nblk256 * P.nsplit * 256 + (long)split * P.N + n0 + tid] = (red[tid] + red[W3_T + tid]) + (red[2 * W3_T + tid] + red[3 * W3_T + tid]);
  }
  W3_STAMP(4);
}

// dW[n, k] += sum over the slices; dbias[n] += sum over the slices (fixed order: deterministic).  A workgroup owns 256
// consecutive floats of dW (64 lanes x float4 = one KB per wave instruction); its sixteen waves take the slices s = w,
// w + 16, ..: every load is a full coalesced KB and all of a wave's loads are in flight together; the partial sums meet
// in LDS.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ ws_base, RedBatch B) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXB; ++i) if (i < B.n && (int)blockIdx.x >= B.p[i].first) pi = i;
  const RedProb R = B.p[pi];
  const float* __restrict__ ws = ws_base + R.wsoff;
  float* __restrict__ dW = R.dW;
  float* __restrict__ dbias = R.dbias;
  const int nsplit = R.nsplit, N = R.N, nblk_w = R.nblk_w, overwrite = R.overwrite;
  const long NK = R.NK;
  const int blk = (int)blockIdx.x - R.first;
  __shared__ f32x4 red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blk < nblk_w) {
    const long q = ((long)blk * 64 + lane) * 4;
    f32x4 a = (f32x4)(0.f);
    if (q < NK) {
      const float* base = ws + ((long)blk * nsplit << 8) + lane * 4;       // this block's [slice][256] run
      int s = wave;
      for (; s + 48 < nsplit; s += 64) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + ((long)(s + 16 * u) << 8));
        a += (v[0] + v[1]) + (v[2] + v[3]);
      }
      for (; s < nsplit; s += 16) a += *reinterpret_cast<const f32x4*>(base + ((long)s << 8));
    }
    red[wave][lane] = a;
    __syncthreads();
    if (wave == 0 && q < NK) {
      f32x4 t = red[0][lane];
#pragma unroll
      for (int w = 1; w < 16; ++w) t += red[w][lane];
      if (R.taps == 0) {
        f32x4* d = reinterpret_cast<f32x4*>(dW + q);
        *d = overwrite ? t : *d + t;
      } else {
        const int K = R.taps * R.cin_p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long qe = q + e;
          const int n = (int)(qe / K), k = (int)(qe - (long)n * K);
          const int tap = k / R.cin_p, c = k - tap * R.cin_p;
          if (n < R.co && c < R.ci) {
            float* d = dW + ((long)n * R.ci + c) * R.taps + tap;
            *d = overwrite ? t[e] : *d + t[e];
          }
        }
      }
    }
  } else if (dbias != nullptr) {
    // bias: 64 entries per workgroup, the slices spread over the 16 waves like above (a thread per entry walking all the
    // slices serially costs one memory round trip per few slices: 20+ us for 128 slices)
    const long n = ((long)blk - nblk_w) * 64 + lane;
    const float* wb = ws + ((long)nblk_w * nsplit << 8);
    float a = 0.f;
    if (n < N) {
      int s = wave;
      for (; s + 48 < nsplit; s += 64) {
        const float v0 = wb[(long)s * N + n], v1 = wb[(long)(s + 16) * N + n];
        const float v2 = wb[(long)(s + 32) * N + n], v3 = wb[(long)(s + 48) * N + n];
        a += (v0 + v1) + (v2 + v3);
      }
      for (; s < nsplit; s += 16) a += wb[(long)s * N + n];
    }
    float* redf = reinterpret_cast<float*>(red);
    redf[wave * 64 + lane] = a;
    __syncthreads();
    if (wave == 0 && n < (R.taps == 0 ? N : R.co)) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += redf[w * 64 + lane];
      dbias[n] = overwrite ? t : dbias[n] + t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&f)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&f)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = v[e];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&f)[4]) {
  const s16x4 v = *reinterpret_cast<const s16x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = bf16_bits_to_f32((unsigned short)v[e]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&f)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&f)[4]) {
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = f[e];
  *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&f)[4]) {
  s16x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (short)f32_to_bf16_bits(f[e]);
  *reinterpret_cast<s16x4*>(p) = v;
}

// one wave per row; lanes own interleaved 4-element column groups
template <typename T>
__global__ __launch_bounds__(NT) void ln_stats_kernel(const T* __restrict__ X, long ldx, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int M, int K, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  for (int m = row0; m < M; m += gridDim.x * (NT / 64)) {
    const T* x = X + (long)m * ldx;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float f[4];
      load4<T>(x + k, f);
      s += (f[0] + f[1]) + (f[2] + f[3]);
    }
    s = wave_sum(s);
    const float mu = s / (float)K;
    float q = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float f[4];
      load4<T>(x + k, f);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = f[e] - mu; q += d * d; }
    }
    q = wave_sum(q);
    if (lane == 0) { mean[m] = mu; rstd[m] = rsqrtf(q / (float)K + eps); }
  }
}

// dx = rstd * (g*dy - mean_k(g*dy) - xhat * mean_k(g*dy*xhat)) + skip ;  dgamma += sum_m dy*xhat ; dbeta += sum_m dy
// KMAX4 = K / 256 rounded up: per-lane column groups kept in registers.
// 16 waves per workgroup: the dgamma / dbeta partials of 16 rows-in-flight meet in LDS, so each column receives a quarter of
// the same-address float atomics a 4-wave workgroup would send (those serialise in L2 and all arrive at the end).
template <typename T, int KG>
__global__ __launch_bounds__(1024 / KG) void ln_bwd_kernel(const T* __restrict__ X, long ldx, const T* __restrict__ DY, long lddy,
                                                    const T* __restrict__ SKIP, long ldskip, const T* __restrict__ SKIP2,
                                                    long ldskip2, const float* __restrict__ gamma,
                                                    T* __restrict__ DX, long lddx, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int M, int K, float eps) {
  constexpr int NTL = 1024 / KG;                        // 16 / 8 / 4 waves: the LDS partials stay at 32 KB
  __shared__ float red[2][NTL / 64][KG * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gsum[KG][4], bsum[KG][4], gam[KG][4];
#pragma unroll
  for (int c = 0; c < KG; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gsum[c][e] = 0.f; bsum[c][e] = 0.f;
      const int k = c * 256 + lane * 4 + e;
      gam[c][e] = k < K ? gamma[k] : 0.f;
    }
  // TWO rows per wave and iteration: both rows' loads (x, dy, the skips) are in flight before the first reduction (one row in
  // flight: 72 us per call at 65 536 x 384, two: 58, four: 84 -- 154 registers, a third of the waves).  The second row of a
  // wave's last pair may lie past M: it is clamped for the loads, its dy zeroed (exact zeros into the column sums), no store.
  constexpr int U = 2;
  const int stride = gridDim.x * (NTL / 64);
  for (int m0 = blockIdx.x * (NTL / 64) + wave; m0 < M; m0 += U * stride) {
    float xv[U][KG][4], dy[U][KG][4], sk[U][KG][4];
    int mr[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      live[u] = m0 + u * stride < M;
      mr[u] = live[u] ? m0 + u * stride : m0;
#pragma unroll
      for (int c = 0; c < KG; ++c) {
        const int k0 = c * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[u][c][e] = 0.f; dy[u][c][e] = 0.f; sk[u][c][e] = 0.f; }
        if (k0 < K) {
          load4<T>(X + (long)mr[u] * ldx + k0, xv[u][c]);
          load4<T>(DY + (long)mr[u] * lddy + k0, dy[u][c]);
          if (!live[u]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dy[u][c][e] = 0.f;
          }
          if (SKIP) load4<T>(SKIP + (long)mr[u] * ldskip + k0, sk[u][c]);
          if (SKIP2) {
            float s2[4];
            load4<T>(SKIP2 + (long)mr[u] * ldskip2 + k0, s2);
#pragma unroll
            for (int e = 0; e < 4; ++e) sk[u][c][e] += s2[e];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KG; ++c) s += (xv[u][c][0] + xv[u][c][1]) + (xv[u][c][2] + xv[u][c][3]);
      s = wave_sum(s);
      const float mu = s / (float)K;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < KG; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = c * 256 + lane * 4 + e;
          const float d = k < K ? xv[u][c][e] - mu : 0.f;
          q += d * d;
        }
      q = wave_sum(q);
      const float rs = rsqrtf(q / (float)K + eps);
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int c = 0; c < KG; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = c * 256 + lane * 4 + e;
          const float xh = k < K ? (xv[u][c][e] - mu) * rs : 0.f;
          xv[u][c][e] = xh;
          const float gd = gam[c][e] * dy[u][c][e];
          c1 += gd;
          c2 += gd * xh;
          gsum[c][e] += dy[u][c][e] * xh;
          bsum[c][e] += dy[u][c][e];
        }
      c1 = wave_sum(c1);
      c2 = wave_sum(c2);
      c1 /= (float)K;
      c2 /= (float)K;
      if (live[u]) {
#pragma unroll
        for (int c = 0; c < KG; ++c) {
          const int k0 = c * 256 + lane * 4;
          if (k0 < K) {
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = rs * (gam[c][e] * dy[u][c][e] - c1 - xv[u][c][e] * c2) + sk[u][c][e];
            store4<T>(DX + (long)mr[u] * lddx + k0, out);
          }
        }
      }
    }
  }
  // per-workgroup reduction over the 4 waves, then one atomic per column
#pragma unroll
  for (int c = 0; c < KG; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave][c * 256 + lane * 4 + e] = gsum[c][e];
      red[1][wave][c * 256 + lane * 4 + e] = bsum[c][e];
    }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += NTL) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < NTL / 64; ++w) { a += red[0][w][k]; b += red[1][w][k]; }
    atomicAdd(dgamma + k, a);
    atomicAdd(dbeta + k, b);
  }
}

// The same sums for FEW slices (<= 8) of plain [N, K] gradients: one WAVE per 256-float block (all of its slices' loads in
// flight at once), four blocks per workgroup -- the 16-wave form above spends a 1024-thread workgroup on two or three KB here.
// Same summation order (slice 0, 1, ..): the same bits.
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ ws_base, RedBatch B, int nred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = (int)blockIdx.x * 4 + wave;
  if (g >= nred) return;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXB; ++i) if (i < B.n && g >= B.p[i].first) pi = i;
  const RedProb R = B.p[pi];
  const float* __restrict__ ws = ws_base + R.wsoff;
  const int nsplit = R.nsplit, blk = g - R.first;
  if (blk < R.nblk_w) {
    const long q = ((long)blk * 64 + lane) * 4;
    if (q >= R.NK) return;
    const float* base = ws + ((long)blk * nsplit << 8) + lane * 4;
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = u < nsplit ? *reinterpret_cast<const f32x4*>(base + ((long)u << 8)) : (f32x4)(0.f);
    f32x4 t = v[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) if (u < nsplit) t += v[u];
    f32x4* d = reinterpret_cast<f32x4*>(R.dW + q);
    *d = R.overwrite ? t : *d + t;
  } else if (R.dbias != nullptr) {
    const long n = ((long)blk - R.nblk_w) * 64 + lane;
    if (n >= R.N) return;
    const float* wb = ws + ((long)R.nblk_w * nsplit << 8);
    float t = 0.f;
    for (int u = 0; u < nsplit; ++u) t += wb[(long)u * R.N + n];
    R.dbias[n] = R.overwrite ? t : R.dbias[n] + t;
  }
}

}  // namespace

extern "C" int wmz_linear_wgrad(const void* dC, long ldc, const void* A, long lda, float* dW, float* dbias, int M, int N,
                                int K, const float* ln_gamma, const float* ln_beta, const float* ln_mean,
                                const float* ln_rstd, int gelu_in, int dtype, void* stream) {
  WMZ_REQUIRE(dC && A && dW, "wmz_linear_wgrad: null tensor");
  WMZ_REQUIRE(M > 0 && N > 0 && K > 0, "wmz_linear_wgrad: bad shape");
  WMZ_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldc % 8 == 0 && lda % 8 == 0, "wmz_linear_wgrad: N, K and row strides must be multiples of 8");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_linear_wgrad: bad dtype %d", dtype);
  const bool ln = ln_gamma != nullptr;
  WMZ_REQUIRE(!ln || (ln_beta && ln_mean && ln_rstd), "wmz_linear_wgrad: LayerNorm prologue needs gamma, beta, mean, rstd");
  WMZ_REQUIRE(!(ln && gelu_in), "wmz_linear_wgrad: LayerNorm and GELU prologues are exclusive");
  WgParams P;
  P.dC = dC; P.ldc = ldc; P.A = A; P.lda = lda; P.dW = dW; P.dbias = dbias; P.M = M; P.N = N; P.K = K;
  P.gamma = ln_gamma; P.beta = ln_beta; P.mean = ln_mean; P.rstd = ln_rstd; P.gelu_in = gelu_in;
  P.Hi = P.Wi = P.Cin = P.Ho = P.Wo = P.KW = P.cstride = P.cpad = 0;
  P.nbn = wmz_cdiv(N, WG_BN); P.nbk = wmz_cdiv(K, WG_BK); P.nsplit = 0; P.direct = 0;
  const int tiles = P.nbn * P.nbk;
  constexpr int wg_target = 256;             // ~one workgroup per CU (measured optimum)
  int split = wmz_cdiv(wg_target, tiles);    // ~one workgroup per CU: every extra split is another 64 KB of float atomics
  const int max_split = wmz_cdiv(M, 4 * WG_MS);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  P.rows_per_wg = wmz_cdiv(wmz_cdiv(M, split), WG_MS) * WG_MS;
  split = wmz_cdiv(M, P.rows_per_wg);
  dim3 grid((unsigned)(tiles * split)), block(NT);
  hipStream_t st = (hipStream_t)stream;
  const int pro = ln ? 1 : (gelu_in ? 2 : 0);
#define WMZ_WG(T, PRO) hipLaunchKernelGGL((wgrad_kernel<T, PRO>), grid, block, 0, st, P)
  if (dtype == WMZ_BF16) { if (pro == 1) WMZ_WG(bf16_t, 1); else if (pro == 2) WMZ_WG(bf16_t, 2); else WMZ_WG(bf16_t, 0); }
  else { if (pro == 1) WMZ_WG(float, 1); else if (pro == 2) WMZ_WG(float, 2); else WMZ_WG(float, 0); }
#undef WMZ_WG
  WMZ_LAUNCH_CHECK("wmz_linear_wgrad");
  return WMZ_OK;
}

// slices of M for a wgrad of these sizes (shared by the launch and by the workspace query)
// target of workgroups per problem (set by the entry points around their own launches; host-side, single stream of calls):
// a batch of n problems fills the chip together, and every slice a problem does NOT have is 2 x N x K x 4 bytes of partial
// tiles not written and read back
static thread_local int g_wgrad_target = 256;
static int wgrad_split(int M, int N, int K, int ms, int* rows_per_wg) {
  const int tiles = wmz_cdiv(N, WG_BN) * wmz_cdiv(K, WG_BK);
  const int target = g_wgrad_target;         // measured: 512 slices double the partial-tile traffic and lose
  int split = wmz_cdiv(target, tiles);         // two resident workgroups per CU
  const int max_split = wmz_cdiv(M, 4 * ms);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  *rows_per_wg = wmz_cdiv(wmz_cdiv(M, split), ms) * ms;
  return wmz_cdiv(M, *rows_per_wg);
}

extern "C" long wmz_linear_wgrad_workspace_floats(int M, int N, int K, int dtype) {
  int rows;
  const int split = wgrad_split(M, N, K, dtype == WMZ_BF16 ? 64 : 32, &rows);
  return (long)split * ((((long)N * K + 255) >> 8) * 256 + N);
}

namespace {
// fills problem i of a batch; returns the workspace floats it needs
long wg_batch_add(WgBatch& B, RedBatch& R, int i, const void* dC, long ldc, const void* A, long lda, float* dW, float* dbias,
                  int M, int N, int K, const float* g, const float* b, const float* mean, const float* rstd, int gelu_in,
                  int overwrite, int dtype, long wsoff, int a_tiled = 0) {
  WgParams& P = B.p[i];
  P.a_tiled = a_tiled;
  P.dC = dC; P.ldc = ldc; P.A = A; P.lda = lda; P.dW = dW; P.dbias = dbias; P.M = M; P.N = N; P.K = K;
  P.gamma = g; P.beta = b; P.mean = mean; P.rstd = rstd; P.gelu_in = gelu_in;
  P.Hi = P.Wi = P.Cin = P.Ho = P.Wo = P.KW = P.cstride = P.cpad = 0;
  P.nbn = wmz_cdiv(N, WG_BN); P.nbk = wmz_cdiv(K, WG_BK);
  P.direct = 0;
  P.nsplit = wgrad_split(M, N, K, dtype == WMZ_BF16 ? 64 : 32, &P.rows_per_wg);
  B.first[i + 1] = B.first[i] + P.nbn * P.nbk * P.nsplit;
  B.wsoff[i] = wsoff;
  const long NK = (long)N * K;
  RedProb& Q = R.p[i];
  Q.dW = dW; Q.dbias = dbias; Q.wsoff = wsoff; Q.NK = NK; Q.nsplit = P.nsplit; Q.N = N;
  Q.nblk_w = wmz_cdiv(NK, 256); Q.overwrite = overwrite;
  Q.taps = 0; Q.cin_p = 0; Q.co = 0; Q.ci = 0;
  Q.first = i == 0 ? 0 : R.p[i - 1].first + R.p[i - 1].nblk_w + (R.p[i - 1].dbias != nullptr ? wmz_cdiv(R.p[i - 1].N, 64) : 0);
  return wmz_linear_wgrad_workspace_floats(M, N, K, dtype);
}
int wg_batch_launch(const WgBatch& B, const RedBatch& R, int pro, float* workspace, int dtype, hipStream_t st) {
  const int n = B.n;
  dim3 grid((unsigned)B.first[n]), block(NT);
#define WMZ_WG2(T, PRO) hipLaunchKernelGGL((wgrad2_kernel<T, PRO>), grid, block, 0, st, B, workspace)
  if (dtype == WMZ_BF16) { if (pro == 1) WMZ_WG2(bf16_t, 1); else if (pro == 2) WMZ_WG2(bf16_t, 2); else if (pro == 3) WMZ_WG2(bf16_t, 3); else WMZ_WG2(bf16_t, 0); }
  else { if (pro == 1) WMZ_WG2(float, 1); else if (pro == 2) WMZ_WG2(float, 2); else if (pro == 3) WMZ_WG2(float, 3); else WMZ_WG2(float, 0); }
#undef WMZ_WG2
  const RedProb& L = R.p[n - 1];
  const int nred = L.first + L.nblk_w + (L.dbias != nullptr ? wmz_cdiv(L.N, 64) : 0);
  bool small = true;
  for (int i = 0; i < n; ++i) small = small && R.p[i].nsplit <= 8 && R.p[i].taps == 0;
  if (nred > 0 && small) hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)wmz_cdiv(nred, 4)), dim3(256), 0, st, workspace, R, nred);
  else if (nred > 0) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nred), dim3(1024), 0, st, workspace, R);
  return WMZ_OK;
}
}  // namespace

int wmz_wgrad_reduce_launch(const float* workspace, float* dW, float* dbias, long NK, int nsplit, int N, int overwrite, int taps,
                            int cin_p, int co, int ci, hipStream_t stream) {
  RedBatch R;
  R.n = 1;
  RedProb& Q = R.p[0];
  Q.dW = dW; Q.dbias = dbias; Q.wsoff = 0; Q.NK = NK; Q.nsplit = nsplit; Q.N = N; Q.nblk_w = wmz_cdiv(NK, 256); Q.overwrite = overwrite;
  Q.taps = taps; Q.cin_p = cin_p; Q.co = co; Q.ci = ci; Q.first = 0;
  const int nred = Q.nblk_w + (dbias != nullptr ? wmz_cdiv(N, 64) : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nred), dim3(1024), 0, stream, workspace, R);
  return WMZ_OK;
}

extern "C" int wmz_linear_wgrad_ws(const void* dC, long ldc, const void* A, long lda, float* dW, float* dbias, int M, int N,
                                   int K, const float* ln_gamma, const float* ln_beta, const float* ln_mean,
                                   const float* ln_rstd, int gelu_in, int overwrite, float* workspace,
                                   long workspace_floats, int dtype, void* stream) {
  WMZ_REQUIRE(dC && A && dW && workspace, "wmz_linear_wgrad_ws: null tensor");
  WMZ_REQUIRE(M > 0 && N > 0 && K > 0, "wmz_linear_wgrad_ws: bad shape");
  WMZ_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldc % 8 == 0 && lda % 8 == 0, "wmz_linear_wgrad_ws: N, K and row strides must be multiples of 8");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_linear_wgrad_ws: bad dtype %d", dtype);
  const bool ln = ln_gamma != nullptr;
  WMZ_REQUIRE(!ln || (ln_beta && ln_mean && ln_rstd), "wmz_linear_wgrad_ws: LayerNorm prologue needs gamma, beta, mean, rstd");
  WMZ_REQUIRE(!(ln && gelu_in), "wmz_linear_wgrad_ws: LayerNorm and GELU prologues are exclusive");
  WMZ_REQUIRE(workspace_floats >= wmz_linear_wgrad_workspace_floats(M, N, K, dtype), "wmz_linear_wgrad_ws: workspace too small (%ld floats)", workspace_floats);
  WgBatch B;
  RedBatch R;
  B.n = R.n = 1;
  B.first[0] = 0;
  wg_batch_add(B, R, 0, dC, ldc, A, lda, dW, dbias, M, N, K, ln_gamma, ln_beta, ln_mean, ln_rstd, gelu_in, overwrite, dtype, 0);
  wg_batch_launch(B, R, ln ? 1 : (gelu_in ? 2 : 0), workspace, dtype, (hipStream_t)stream);
  WMZ_LAUNCH_CHECK("wmz_linear_wgrad_ws");
  return WMZ_OK;
}

static int wgrad_batch_plain(int n, const void* const* dC, const long* ldc, const void* const* A, const long* lda,
                             float* const* dW, float* const* dbias, const int* M, const int* N, const int* K,
                             const int* overwrite, const int* a_tiled, float* workspace, long workspace_floats,
                             int dtype, void* stream, bool single_v3);

#ifdef WMZ_W3_STAMPS
extern "C" int wmz_debug_w3_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w3_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" int wmz_linear_wgrad_batch(int n, const void* const* dC, const long* ldc, const void* const* A, const long* lda,
                                      float* const* dW, float* const* dbias, const int* M, const int* N, const int* K,
                                      const int* overwrite, const int* a_tiled, float* workspace, long workspace_floats,
                                      int dtype, void* stream) {
  return wgrad_batch_plain(n, dC, ldc, A, lda, dW, dbias, M, N, K, overwrite, a_tiled, workspace, workspace_floats, dtype, stream, false);
}

// 256-wide tiles (wgrad3_kernel) pay off for a bf16 problem that fills at least WMZ_W3_FILL16 / 16 of its tiles and has rows to slice
#ifndef WMZ_W3_FILL16
#define WMZ_W3_FILL16 6         // 128 x 384 (the published width's to_q / to_out) rides along: 12.83 -> 12.65 ms a step
#endif
static bool wgrad3_eligible(int M, int N, int K, int dtype) {
  const int tn = wmz_cdiv(N, W3_T), tk = wmz_cdiv(K, W3_T);
  return dtype == WMZ_BF16 && (long)N * K * 16 >= (long)tn * tk * W3_T * W3_T * WMZ_W3_FILL16 && M >= 4 * W3_MS;
}

static int wgrad_batch_plain(int n, const void* const* dC, const long* ldc, const void* const* A, const long* lda,
                             float* const* dW, float* const* dbias, const int* M, const int* N, const int* K,
                             const int* overwrite, const int* a_tiled, float* workspace, long workspace_floats,
                             int dtype, void* stream, bool single_v3) {
  WMZ_REQUIRE(n >= 1 && n <= WG_MAXB, "wmz_linear_wgrad_batch: 1 .. %d problems per call (got %d)", WG_MAXB, n);
  WMZ_REQUIRE(dC && ldc && A && lda && dW && dbias && M && N && K && overwrite && workspace, "wmz_linear_wgrad_batch: null table");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_linear_wgrad_batch: bad dtype %d", dtype);
  WgBatch B;
  RedBatch R;
  B.n = R.n = n;
  B.first[0] = 0;
  long off = 0;
  struct TargetGuard { int saved; ~TargetGuard() { g_wgrad_target = saved; } } guard{g_wgrad_target};
  g_wgrad_target = n >= 3 ? 96 : (n == 2 ? 128 : 256);       // measured at five problems: 96-128 best (2.12-2.14 vs 2.17 ms a step)
  for (int i = 0; i < n; ++i) {
    WMZ_REQUIRE(dC[i] && A[i] && dW[i] && M[i] > 0 && N[i] > 0 && K[i] > 0, "wmz_linear_wgrad_batch: bad problem %d", i);
    WMZ_REQUIRE(N[i] % 8 == 0 && K[i] % 8 == 0 && ldc[i] % 8 == 0 && lda[i] % 8 == 0, "wmz_linear_wgrad_batch: problem %d: N, K and row strides must be multiples of 8", i);
    const int tiled = a_tiled != nullptr && a_tiled[i] != 0;
    WMZ_REQUIRE(!tiled || (dtype == WMZ_BF16 && K[i] == 256 && M[i] % 32 == 0),
                "wmz_linear_wgrad_batch: problem %d: a tiled A is bf16, 256 wide, whole 32-row tiles", i);
    off += wg_batch_add(B, R, i, dC[i], ldc[i], A[i], lda[i], dW[i], dbias[i], M[i], N[i], K[i], nullptr, nullptr, nullptr,
                        nullptr, 0, overwrite[i], dtype, off, tiled);
  }
  WMZ_REQUIRE(workspace_floats >= off, "wmz_linear_wgrad_batch: workspace too small (%ld floats needed)", off);
  // bf16 problems that fill at least half of a 256 x 256 tile (the fused path's five: 256 x 256, 256 x 128, 128 x 256): version 3
  bool v3 = dtype == WMZ_BF16 && (n >= 2 || single_v3);
  int tiles3 = 0;
  for (int i = 0; i < n && v3; ++i) {
    v3 = wgrad3_eligible(M[i], N[i], K[i], dtype);
    tiles3 += wmz_cdiv(N[i], W3_T) * wmz_cdiv(K[i], W3_T);
  }
  if (v3 && tiles3 <= 256) {
    long off3 = 0;
    const int per = 256 / tiles3;                    // ~one workgroup per CU in all (8 waves, 128 KB of LDS: one fits)
    for (int i = 0; i < n; ++i) {
      WgParams& P = B.p[i];
      P.nbn = wmz_cdiv(N[i], W3_T); P.nbk = wmz_cdiv(K[i], W3_T);
      int split = per < 1 ? 1 : per;
      const int max_split = M[i] / (4 * W3_MS);
      if (split > max_split) split = max_split;
      P.rows_per_wg = wmz_cdiv(wmz_cdiv(M[i], split), W3_MS) * W3_MS;
      P.nsplit = wmz_cdiv(M[i], P.rows_per_wg);
      P.direct = 0;
      B.first[i + 1] = B.first[i] + P.nbn * P.nbk * P.nsplit;
      B.wsoff[i] = off3;
      RedProb& Q = R.p[i];
      Q.nsplit = P.nsplit; Q.wsoff = off3;
      Q.first = i == 0 ? 0 : R.p[i - 1].first + R.p[i - 1].nblk_w + (R.p[i - 1].dbias != nullptr ? wmz_cdiv(R.p[i - 1].N, 64) : 0);
      off3 += (long)P.nsplit * ((((long)N[i] * K[i] + 255) >> 8) * 256 + N[i]);
    }
    if (off3 <= workspace_floats) {
      hipStream_t st = (hipStream_t)stream;
      hipLaunchKernelGGL(wgrad3_kernel, dim3((unsigned)B.first[n]), dim3(W3_NT), 0, st, B, workspace);
      const RedProb& Lp = R.p[n - 1];
      const int nred = Lp.first + Lp.nblk_w + (Lp.dbias != nullptr ? wmz_cdiv(Lp.N, 64) : 0);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nred), dim3(1024), 0, st, workspace, R);
      WMZ_LAUNCH_CHECK("wmz_linear_wgrad_batch");
      return WMZ_OK;
    }
    // (workspace sized for version 2's slicing only: fall through to it -- rebuild the tables)
    off = 0;
    for (int i = 0; i < n; ++i) {
      const int tiled = a_tiled != nullptr && a_tiled[i] != 0;
      off += wg_batch_add(B, R, i, dC[i], ldc[i], A[i], lda[i], dW[i], dbias[i], M[i], N[i], K[i], nullptr, nullptr, nullptr,
                          nullptr, 0, overwrite[i], dtype, off, tiled);
    }
  }
  wg_batch_launch(B, R, 0, workspace, dtype, (hipStream_t)stream);
  WMZ_LAUNCH_CHECK("wmz_linear_wgrad_batch");
  return WMZ_OK;
}

// The same with an optional LayerNorm prologue PER PROBLEM (tables of n pointers, NULL entries = plain): the four weight
// gradients of a transformer layer on the op-by-op path -- to_qkv and the feed-forward's first GEMM behind their PreNorm, to_out
// and the second GEMM plain -- by one launch pair.
extern "C" int wmz_linear_wgrad_batch_ln(int n, const void* const* dC, const long* ldc, const void* const* A, const long* lda,
                                         float* const* dW, float* const* dbias, const int* M, const int* N, const int* K,
                                         const int* overwrite, const float* const* ln_gamma, const float* const* ln_beta,
                                         const float* const* ln_mean, const float* const* ln_rstd, float* workspace,
                                         long workspace_floats, int dtype, void* stream) {
  WMZ_REQUIRE(n >= 1 && n <= WG_MAXB, "wmz_linear_wgrad_batch_ln: 1 .. %d problems per call (got %d)", WG_MAXB, n);
  WMZ_REQUIRE(dC && ldc && A && lda && dW && dbias && M && N && K && overwrite && workspace && ln_gamma && ln_beta && ln_mean && ln_rstd,
              "wmz_linear_wgrad_batch_ln: null table");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_linear_wgrad_batch_ln: bad dtype %d", dtype);
  // Many rows (the published widths' training step: 65 536 tokens): the plain problems that suit the 256-wide tiles leave as
  // their own launch pair on wgrad3_kernel, the rest (LayerNorm prologue, narrow shapes) as before.  One stream, in order: the
  // second pair may reuse the workspace.
  if (dtype == WMZ_BF16 && n >= 2) {
    int e_idx[WG_MAXB], r_idx[WG_MAXB], ne = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
      const bool plain = ln_gamma[i] == nullptr;
      if (plain && M[i] >= 16384 && wgrad3_eligible(M[i], N[i], K[i], dtype)) e_idx[ne++] = i; else r_idx[nr++] = i;
    }
    if (ne > 0 && nr > 0) {
      const void* sdC[WG_MAXB]; const void* sA[WG_MAXB]; long sldc[WG_MAXB], slda[WG_MAXB]; float* sdW[WG_MAXB]; float* sdb[WG_MAXB];
      int sM[WG_MAXB], sN[WG_MAXB], sK[WG_MAXB], sov[WG_MAXB];
      const float* sg[WG_MAXB]; const float* sb[WG_MAXB]; const float* sm[WG_MAXB]; const float* sr[WG_MAXB];
      auto gather = [&](const int* idx, int cnt) {
        for (int j = 0; j < cnt; ++j) {
          const int i = idx[j];
          sdC[j] = dC[i]; sA[j] = A[i]; sldc[j] = ldc[i]; slda[j] = lda[i]; sdW[j] = dW[i]; sdb[j] = dbias[i];
          sM[j] = M[i]; sN[j] = N[i]; sK[j] = K[i]; sov[j] = overwrite[i];
          sg[j] = ln_gamma[i]; sb[j] = ln_beta[i]; sm[j] = ln_mean[i]; sr[j] = ln_rstd[i];
        }
      };
      gather(e_idx, ne);
      int rc = wgrad_batch_plain(ne, sdC, sldc, sA, slda, sdW, sdb, sM, sN, sK, sov, nullptr, workspace, workspace_floats, dtype, stream, true);
      if (rc != WMZ_OK) return rc;
      gather(r_idx, nr);
      return wmz_linear_wgrad_batch_ln(nr, sdC, sldc, sA, slda, sdW, sdb, sM, sN, sK, sov, sg, sb, sm, sr, workspace, workspace_floats, dtype, stream);
    }
    if (ne == n)
      return wgrad_batch_plain(n, dC, ldc, A, lda, dW, dbias, M, N, K, overwrite, nullptr, workspace, workspace_floats, dtype, stream, true);
  }
  WgBatch B;
  RedBatch R;
  B.n = R.n = n;
  B.first[0] = 0;
  long off = 0;
  struct TargetGuard { int saved; ~TargetGuard() { g_wgrad_target = saved; } } guard{g_wgrad_target};
  g_wgrad_target = n >= 3 ? 96 : (n == 2 ? 128 : 256);
  bool any_ln = false;
  int tiles_total = 0;
  for (int i = 0; i < n; ++i) {
    WMZ_REQUIRE(dC[i] && A[i] && dW[i] && M[i] > 0 && N[i] > 0 && K[i] > 0, "wmz_linear_wgrad_batch_ln: bad problem %d", i);
    WMZ_REQUIRE(N[i] % 8 == 0 && K[i] % 8 == 0 && ldc[i] % 8 == 0 && lda[i] % 8 == 0, "wmz_linear_wgrad_batch_ln: problem %d: N, K and row strides must be multiples of 8", i);
    const bool ln = ln_gamma[i] != nullptr;
    WMZ_REQUIRE(!ln || (ln_beta[i] && ln_mean[i] && ln_rstd[i]), "wmz_linear_wgrad_batch_ln: problem %d: the LayerNorm prologue needs gamma, beta, mean, rstd", i);
    any_ln = any_ln || ln;
    off += wg_batch_add(B, R, i, dC[i], ldc[i], A[i], lda[i], dW[i], dbias[i], M[i], N[i], K[i], ln ? ln_gamma[i] : nullptr,
                        ln ? ln_beta[i] : nullptr, ln ? ln_mean[i] : nullptr, ln ? ln_rstd[i] : nullptr, 0, overwrite[i], dtype, off, 0);
    tiles_total += B.p[i].nbn * B.p[i].nbk;
  }
  // Few rows (config 5: 3 072 per GPU) and the batch's tiles alone cover half the chip: no slices of M at all -- every tile has
  // ONE workgroup, which adds its registers straight into dW (no partial tiles written and read back, no reduction launch;
  // at 2-6 slices the reduction kernel ran at a few hundred GB/s and cost as much as the GEMMs).
  bool same_m = true;
  for (int i = 1; i < n; ++i) same_m = same_m && M[i] == M[0];
  if (same_m && M[0] <= 8192 && tiles_total >= 128) {
    // (1 .. 4 slices measured the same step time at config 5, 3.05-3.16 ms: the launches sit on a side branch of the graph)
    int split = wmz_cdiv(256, tiles_total);
    const int ms = dtype == WMZ_BF16 ? 64 : 32;
    const int max_split = M[0] / 1024 > 0 ? M[0] / 1024 : 1;
    if (split > max_split) split = max_split;
    if (split > 8) split = 8;
    off = 0;
    for (int i = 0; i < n; ++i) {
      WgParams& P = B.p[i];
      P.rows_per_wg = wmz_cdiv(wmz_cdiv(M[i], split), ms) * ms;
      P.nsplit = wmz_cdiv(M[i], P.rows_per_wg);
      P.direct = P.nsplit == 1 ? (overwrite[i] ? 2 : 1) : 0;
      B.first[i + 1] = B.first[i] + P.nbn * P.nbk * P.nsplit;
      B.wsoff[i] = off;
      RedProb& Q = R.p[i];
      Q.nsplit = P.nsplit; Q.wsoff = off;
      if (P.direct) { Q.nblk_w = 0; Q.dbias = nullptr; }
      Q.first = i == 0 ? 0 : R.p[i - 1].first + R.p[i - 1].nblk_w + (R.p[i - 1].dbias != nullptr ? wmz_cdiv(R.p[i - 1].N, 64) : 0);
      off += (long)P.nsplit * ((((long)N[i] * K[i] + 255) >> 8) * 256 + N[i]);
    }
    WMZ_REQUIRE(workspace_floats >= off, "wmz_linear_wgrad_batch_ln: workspace too small (%ld floats needed)", off);
  }
  WMZ_REQUIRE(workspace_floats >= off, "wmz_linear_wgrad_batch_ln: workspace too small (%ld floats needed)", off);
  wg_batch_launch(B, R, any_ln ? 1 : 0, workspace, dtype, (hipStream_t)stream);
  WMZ_LAUNCH_CHECK("wmz_linear_wgrad_batch_ln");
  return WMZ_OK;
}

extern "C" int wmz_conv2d_nhwc_wgrad(const void* x, const void* dy, float* dW, float* dbias, int B, int Hi, int Wi, int Cin,
                                     int Cout, int KH, int KW, int stride, int pad, int dtype, void* stream) {
  WMZ_REQUIRE(x && dy && dW, "wmz_conv2d_nhwc_wgrad: null tensor");
  WMZ_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "wmz_conv2d_nhwc_wgrad: bad shape");
  WMZ_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "wmz_conv2d_nhwc_wgrad: Cin and Cout must be multiples of 8 (zero-pad)");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_conv2d_nhwc_wgrad: bad dtype %d", dtype);
  WgParams P;
  P.Hi = Hi; P.Wi = Wi; P.Cin = Cin; P.KW = KW; P.cstride = stride; P.cpad = pad;
  P.Ho = (Hi + 2 * pad - KH) / stride + 1;
  P.Wo = (Wi + 2 * pad - KW) / stride + 1;
  WMZ_REQUIRE(P.Ho > 0 && P.Wo > 0, "wmz_conv2d_nhwc_wgrad: empty output");
  P.dC = dy; P.ldc = Cout; P.A = x; P.lda = 0; P.dW = dW; P.dbias = dbias;
  P.M = B * P.Ho * P.Wo; P.N = Cout; P.K = KH * KW * Cin;
  P.gamma = P.beta = P.mean = P.rstd = nullptr; P.gelu_in = 0;
  P.nbn = wmz_cdiv(P.N, WG_BN); P.nbk = wmz_cdiv(P.K, WG_BK); P.nsplit = 0; P.direct = 0;
  const int tiles = P.nbn * P.nbk;
  int split = wmz_cdiv(256, tiles);
  const int max_split = wmz_cdiv(P.M, 4 * WG_MS);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  P.rows_per_wg = wmz_cdiv(wmz_cdiv(P.M, split), WG_MS) * WG_MS;
  split = wmz_cdiv(P.M, P.rows_per_wg);
  dim3 grid((unsigned)(tiles * split)), block(NT);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 3>), grid, block, 0, st, P);
  else hipLaunchKernelGGL((wgrad_kernel<float, 3>), grid, block, 0, st, P);
  WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_wgrad");
  return WMZ_OK;
}

// The same gradient by the two-stage reduction of wmz_linear_wgrad_ws (partial tiles in a caller-owned workspace, summed in a
// fixed order): no float atomics -- the split-K atomics above are chains of ~30 same-address adds per element of dW (93 us per
// call on the VQ-AE's 3x3 layers) --, and `overwrite` saves the caller the zero fill.
extern "C" long wmz_conv2d_nhwc_wgrad_workspace_floats(int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride,
                                                       int pad, int dtype) {
  const int Ho = (Hi + 2 * pad - KH) / stride + 1, Wo = (Wi + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const long base = wmz_linear_wgrad_workspace_floats(B * Ho * Wo, Cout, KH * KW * Cin, dtype);
  if (wmz_convw_supported(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, dtype)) {
    const long direct = wmz_convw_workspace_floats(B, Hi, Wi, Cin, Cout);
    return direct > base ? direct : base;
  }
  return base;
}
extern "C" int wmz_conv2d_nhwc_wgrad_ws(const void* x, const void* dy, float* dW, float* dbias, int B, int Hi, int Wi, int Cin,
                                        int Cout, int KH, int KW, int stride, int pad, int overwrite, int conv_layout_co,
                                        int conv_layout_ci, float* workspace, long workspace_floats, int dtype, void* stream) {
  WMZ_REQUIRE(x && dy && dW && workspace, "wmz_conv2d_nhwc_wgrad_ws: null tensor");
  WMZ_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "wmz_conv2d_nhwc_wgrad_ws: bad shape");
  WMZ_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "wmz_conv2d_nhwc_wgrad_ws: Cin and Cout must be multiples of 8 (zero-pad)");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_conv2d_nhwc_wgrad_ws: bad dtype %d", dtype);
  const int Ho = (Hi + 2 * pad - KH) / stride + 1, Wo = (Wi + 2 * pad - KW) / stride + 1;
  WMZ_REQUIRE(Ho > 0 && Wo > 0, "wmz_conv2d_nhwc_wgrad_ws: empty output");
  WMZ_REQUIRE((long)B * Ho * Wo < (1L << 31), "wmz_conv2d_nhwc_wgrad_ws: too many output pixels");
  const int M = B * Ho * Wo, K = KH * KW * Cin;
  WMZ_REQUIRE(workspace_floats >= wmz_linear_wgrad_workspace_floats(M, Cout, K, dtype), "wmz_conv2d_nhwc_wgrad_ws: workspace too small");
  if (wmz_convw_supported(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, dtype) && workspace_floats >= wmz_convw_workspace_floats(B, Hi, Wi, Cin, Cout))
    return wmz_convw_launch(x, dy, dW, dbias, B, Hi, Wi, Cin, Cout, overwrite, conv_layout_co, conv_layout_ci, workspace, workspace_floats,
                            (hipStream_t)stream);
  WgBatch Bt;
  RedBatch R;
  Bt.n = R.n = 1;
  Bt.first[0] = 0;
  wg_batch_add(Bt, R, 0, dy, Cout, x, 0, dW, dbias, M, Cout, K, nullptr, nullptr, nullptr, nullptr, 0, overwrite, dtype, 0);
  WgParams& P = Bt.p[0];
  P.Hi = Hi; P.Wi = Wi; P.Cin = Cin; P.KW = KW; P.cstride = stride; P.cpad = pad; P.Ho = Ho; P.Wo = Wo;
  if (conv_layout_co > 0) {
    WMZ_REQUIRE(conv_layout_co <= Cout && conv_layout_ci > 0 && conv_layout_ci <= Cin, "wmz_conv2d_nhwc_wgrad_ws: bad nn.Conv2d layout sizes");
    R.p[0].taps = KH * KW; R.p[0].cin_p = Cin; R.p[0].co = conv_layout_co; R.p[0].ci = conv_layout_ci;
  }
  wg_batch_launch(Bt, R, 3, workspace, dtype, (hipStream_t)stream);
  WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_wgrad_ws");
  return WMZ_OK;
}

// n <= 6 conv weight gradients (any kernel size / stride / padding) by ONE launch pair of the implicit-im2col kernel: the small
// layers of a VQ-AE training step (1x1, 2x2 / stride 2, 3x3 / stride 2, the 3-channel conv_1) are 13 launch pairs otherwise, on a
// side branch of the step's hipGraph whose nodes the host enqueues LAST (~10 us a node).  Workspace: the sum of the problems'
// wmz_conv2d_nhwc_wgrad_workspace_floats is enough.  HOST tables of n entries; dbias[i] may be NULL.
extern "C" int wmz_conv2d_nhwc_wgrad_batch(int n, const void* const* x, const void* const* dy, float* const* dW, float* const* dbias,
                                           const int* B, const int* Hi, const int* Wi, const int* Cin, const int* Cout, const int* KH,
                                           const int* KW, const int* stride, const int* pad, const int* overwrite,
                                           const int* conv_layout_co, const int* conv_layout_ci, float* workspace,
                                           long workspace_floats, int dtype, void* stream) {
  WMZ_REQUIRE(n >= 1 && n <= WG_MAXB, "wmz_conv2d_nhwc_wgrad_batch: 1..%d problems per call (got %d)", WG_MAXB, n);
  WMZ_REQUIRE(x && dy && dW && dbias && B && Hi && Wi && Cin && Cout && KH && KW && stride && pad && overwrite && conv_layout_co &&
              conv_layout_ci && workspace, "wmz_conv2d_nhwc_wgrad_batch: null table");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_conv2d_nhwc_wgrad_batch: bad dtype %d", dtype);
  WgBatch Bt;
  RedBatch R;
  Bt.n = R.n = n;
  Bt.first[0] = 0;
  long off = 0;
  struct TargetGuard { int saved; ~TargetGuard() { g_wgrad_target = saved; } } guard{g_wgrad_target};
  g_wgrad_target = n >= 3 ? 96 : (n == 2 ? 128 : 256);
  for (int i = 0; i < n; ++i) {
    WMZ_REQUIRE(x[i] && dy[i] && dW[i] && B[i] > 0 && Hi[i] > 0 && Wi[i] > 0 && Cin[i] > 0 && Cout[i] > 0 && KH[i] > 0 && KW[i] > 0 &&
                stride[i] > 0 && pad[i] >= 0, "wmz_conv2d_nhwc_wgrad_batch: bad problem %d", i);
    WMZ_REQUIRE(Cin[i] % 8 == 0 && Cout[i] % 8 == 0, "wmz_conv2d_nhwc_wgrad_batch: problem %d: Cin and Cout must be multiples of 8", i);
    const int Ho = (Hi[i] + 2 * pad[i] - KH[i]) / stride[i] + 1, Wo = (Wi[i] + 2 * pad[i] - KW[i]) / stride[i] + 1;
    WMZ_REQUIRE(Ho > 0 && Wo > 0 && (long)B[i] * Ho * Wo < (1L << 31), "wmz_conv2d_nhwc_wgrad_batch: problem %d: bad output size", i);
    const int M = B[i] * Ho * Wo, K = KH[i] * KW[i] * Cin[i];
    off += wg_batch_add(Bt, R, i, dy[i], Cout[i], x[i], 0, dW[i], dbias[i], M, Cout[i], K, nullptr, nullptr, nullptr, nullptr, 0,
                        overwrite[i], dtype, off, 0);
    WgParams& P = Bt.p[i];
    P.Hi = Hi[i]; P.Wi = Wi[i]; P.Cin = Cin[i]; P.KW = KW[i]; P.cstride = stride[i]; P.cpad = pad[i]; P.Ho = Ho; P.Wo = Wo;
    if (conv_layout_co[i] > 0) {
      WMZ_REQUIRE(conv_layout_co[i] <= Cout[i] && conv_layout_ci[i] > 0 && conv_layout_ci[i] <= Cin[i], "wmz_conv2d_nhwc_wgrad_batch: problem %d: bad nn.Conv2d layout sizes", i);
      R.p[i].taps = KH[i] * KW[i]; R.p[i].cin_p = Cin[i]; R.p[i].co = conv_layout_co[i]; R.p[i].ci = conv_layout_ci[i];
    }
  }
  WMZ_REQUIRE(workspace_floats >= off, "wmz_conv2d_nhwc_wgrad_batch: workspace too small (%ld floats needed)", off);
  wg_batch_launch(Bt, R, 3, workspace, dtype, (hipStream_t)stream);
  WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_wgrad_batch");
  return WMZ_OK;
}

/* != 0: wmz_conv2d_nhwc_wgrad_ws takes the direct 3x3 kernel (csrc/conv_wgrad.hip) for this layer -- a launch of its own */
extern "C" int wmz_conv2d_nhwc_wgrad_is_direct(int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype) {
  return wmz_convw_supported(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad, dtype);
}

extern "C" int wmz_layernorm_stats(const void* x, long ldx, float* mean, float* rstd, int M, int K, float eps, int dtype,
                                   void* stream) {
  WMZ_REQUIRE(x && mean && rstd, "wmz_layernorm_stats: null tensor");
  WMZ_REQUIRE(M > 0 && K > 0 && K % 4 == 0, "wmz_layernorm_stats: bad shape (K %% 4 == 0 required)");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_layernorm_stats: bad dtype %d", dtype);
  const int grid = wmz_cdiv(M, 4) < 2048 ? wmz_cdiv(M, 4) : 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16) hipLaunchKernelGGL(ln_stats_kernel<bf16_t>, dim3(grid), dim3(NT), 0, st, (const bf16_t*)x, ldx, mean, rstd, M, K, eps);
  else hipLaunchKernelGGL(ln_stats_kernel<float>, dim3(grid), dim3(NT), 0, st, (const float*)x, ldx, mean, rstd, M, K, eps);
  WMZ_LAUNCH_CHECK("wmz_layernorm_stats");
  return WMZ_OK;
}

extern "C" int wmz_layernorm_bwd(const void* x, long ldx, const void* dyhat, long lddy, const void* skip, long ldskip,
                                 const void* skip2, long ldskip2, const float* gamma, void* dx, long lddx, float* dgamma,
                                 float* dbeta, int M, int K, float eps, int dtype, void* stream) {
  WMZ_REQUIRE(x && dyhat && gamma && dx && dgamma && dbeta, "wmz_layernorm_bwd: null tensor");
  WMZ_REQUIRE(M > 0 && K > 0 && K % 4 == 0, "wmz_layernorm_bwd: bad shape (K %% 4 == 0 required)");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_layernorm_bwd: bad dtype %d", dtype);
  if (K > 1024) { wmz_set_error("wmz_layernorm_bwd: K=%d > 1024 not built", K); return WMZ_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const int kg = wmz_cdiv(K, 256) <= 1 ? 1 : (wmz_cdiv(K, 256) == 2 ? 2 : 4);
  const int rows_per_wg = 16 / kg * 4;
  const int grid = wmz_cdiv(M, rows_per_wg) < 256 * kg ? wmz_cdiv(M, rows_per_wg) : 256 * kg;
#define WMZ_LNB(T, KG) hipLaunchKernelGGL((ln_bwd_kernel<T, KG>), dim3(grid), dim3(1024 / KG), 0, st, (const T*)x, ldx, (const T*)dyhat, lddy, \
                                          (const T*)skip, ldskip, (const T*)skip2, ldskip2, gamma, (T*)dx, lddx, dgamma, dbeta, M, K, eps)
  if (dtype == WMZ_BF16) { if (kg == 1) WMZ_LNB(bf16_t, 1); else if (kg == 2) WMZ_LNB(bf16_t, 2); else WMZ_LNB(bf16_t, 4); }
  else { if (kg == 1) WMZ_LNB(float, 1); else if (kg == 2) WMZ_LNB(float, 2); else WMZ_LNB(float, 4); }
#undef WMZ_LNB
  WMZ_LAUNCH_CHECK("wmz_layernorm_bwd");
  return WMZ_OK;
}
