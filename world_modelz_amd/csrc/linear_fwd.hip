// nn.Linear family with fused PreNorm LayerNorm prologue and bias / GELU / residual epilogue
// (local_3d_attention.py:11-31, :46-53, :159-161; main.py:31).
//
//   C[M,N] = act( LN?(A)[M,K] @ Wt[N,K]^T + bias ) + residual
//
// 128x128 output tile per workgroup, 4 waves as 2x2, each wave 2x2 MFMA 32x32 blocks.  A and Wt are both
// K-contiguous, so both tiles are staged row-wise into 128-byte-row LDS images (XOR-swizzled per row pair:
// conflict-free ds_read_b128 fragment reads) with the next K-slab prefetched into registers while the
// current one is multiplied.  The LayerNorm statistics of the workgroup's 128 rows are computed in a
// prologue (two-pass, fp32) and applied while the A slab is written to LDS.
#include "wmz_common.h"

namespace {

constexpr int BN = 128, ROWB = 128, CPR = 8, NT = 256;

struct LinParams {
  const void* A; long lda;
  const void* Wt;
  const float* bias;
  const void* res; long ldr;
  void* C; long ldc;
  void* C2; long ldc2;                       // optional second output: GELU(C) (C then holds the pre-activation); no residual
  void* An; long ldan;                       // optional, LayerNorm prologue: the normalised rows LN(A) [M, K] as the GEMM consumed them
  int M, N, K;
  const float* gamma; const float* beta; float eps;
  const float* mean; const float* rstd;      // optional precomputed LayerNorm statistics (wmz_layernorm_stats)
  int flags, out_f32;
  int nbn;
  int rpb; long bstride;                     // A rows in blocks: row m lives at A + (m / rpb) * bstride + (m % rpb) * lda (rpb 0: plain)
  int dma;                                   // BM = 64, bf16, no prologue: the K loop as an LDS-DMA ring (linear_launch decides)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void lgkm_wait_for3(s16x8& a, s16x8& b, s16x8& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

// address of A row m (blocked rows: the last frame of every clip, x[:, -1], read in place)
template <typename T> __device__ __forceinline__ const T* a_row(const LinParams& P, int m) {
  const T* A = reinterpret_cast<const T*>(P.A);
  if (P.rpb == 0) return A + (long)m * P.lda;
  const int blk = m / P.rpb;
  return A + (long)blk * P.bstride + (long)(m - blk * P.rpb) * P.lda;
}

__device__ __forceinline__ int swz128(int r) { return ((r >> 1) << 4) & 112; }

__device__ __forceinline__ float gelu_erf(float v) { return wmz_gelu(v); }

template <typename T> __device__ __forceinline__ void chunk_to_f32(const i32x4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const i32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __int_as_float(c[i]);
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const i32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf16_bits_to_f32((unsigned short)((unsigned)c[i] & 0xFFFFu));          // (the unit's 16-bit format: wmz_common.h)
    f[2 * i + 1] = bf16_bits_to_f32((unsigned short)((unsigned)c[i] >> 16));
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
  for (int i = 0; i < 4; ++i)
    c[i] = (int)((unsigned)f32_to_bf16_bits(f[2 * i]) | ((unsigned)f32_to_bf16_bits(f[2 * i + 1]) << 16));
  return c;
}

// PRO: 0 = A as is, 1 = LayerNorm(A) over K, 2 = GELU(A) (FeedForward second GEMM reading the saved pre-activation)
// BM = 128: 4 waves as 2 x 2, each 64 x 64.  BM = 64 (small-M GEMMs that would otherwise leave CUs idle: the last-frame logits,
// M = B*H*W; config 5's 3 072 rows per GPU): 4 waves side by side, each 64 rows x 32 columns.
template <typename T, int PRO, int BM>
__global__ __launch_bounds__(NT, 2) void linear_kernel(LinParams P) {
  constexpr bool LN = PRO == 1;
  constexpr int NJ = BM == 128 ? 2 : 1;      // 32-column MFMA blocks per wave
  constexpr int AI = BM / 32;                // A rows per thread per slab
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = CPR * EPC;              // 64 (bf16) / 32 (f32)
  constexpr int KSTEPS = BK / 16;
  constexpr bool DMA_FORM = PRO == 0 && BM == 64 && sizeof(T) == 2;       // (a ring of three slab pairs instead of one)
  constexpr int SLAB = (BM + BN) * ROWB;
  __shared__ __attribute__((aligned(1024))) char tiles[(DMA_FORM ? 3 : 1) * SLAB];   // A slab | B slab; reused by the epilogue
  char* As = tiles;
  char* Bs = tiles + BM * ROWB;
  __shared__ float mean_s[BM], rstd_s[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lid / P.nbn, bn = lid - bm * P.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const T* A = reinterpret_cast<const T*>(P.A);
  const T* Wt = reinterpret_cast<const T*>(P.Wt);
  const int K = P.K;

  if constexpr (LN) {
   if (P.mean != nullptr) {
    // statistics supplied by the caller (training: computed once, reused by the backward): no extra pass over A
    for (int r = tid; r < BM; r += NT) {
      const int gm = m0 + r < P.M ? m0 + r : P.M - 1;
      mean_s[r] = P.mean[gm];
      rstd_s[r] = P.rstd[gm];
    }
   } else {
    // 8 lanes per row, 32 rows per sweep; two passes (mean, then centred sum of squares)
    const int sub = tid & 7;
    for (int r = tid >> 3; r < BM; r += NT / 8) {
      const int gm = m0 + r;
      float sum = 0.f;
      if (gm < P.M) {
        const T* row = a_row<T>(P, gm);
        for (int c = sub; c * EPC < K; c += 8) {
          float f[EPC];
          chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(row + c * EPC), f);
#pragma unroll
          for (int i = 0; i < EPC; ++i) sum += f[i];
        }
      }
      sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
      const float mean = sum / (float)K;
      float sq = 0.f;
      if (gm < P.M) {
        const T* row = a_row<T>(P, gm);
        for (int c = sub; c * EPC < K; c += 8) {
          float f[EPC];
          chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(row + c * EPC), f);
#pragma unroll
          for (int i = 0; i < EPC; ++i) { const float d = f[i] - mean; sq += d * d; }
        }
      }
      sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
      if (sub == 0) { mean_s[r] = mean; rstd_s[r] = rsqrtf(sq / (float)K + P.eps); }
    }
   }
  }

  // this thread's 4 chunks of each slab: rows r_i = (tid>>3) + 32 i, chunk column cc = tid & 7.
  // TWO register sets: the loads of slabs s + 1 and s + 2 are in flight while slab s is multiplied.  With few workgroups per CU
  // (M = 3 072 rows at config 5: one) the K loop is a chain of memory round trips -- one slab ahead it ran at ~2 us per 64-wide
  // slab, 32 us for a 3 072 x 1 024 x 512 GEMM.  The loads are UNCONDITIONAL (rows, weight rows and the chunk column clamped
  // into the matrices): a branch around a load makes the compiler's wait-count pass fall back to vmcnt(0), which drains both
  // sets.  Clamped rows only feed output rows / columns that are never stored; a chunk past K is zeroed in the WEIGHT slab
  // when it is written to LDS (finite x 0).
  const int cc = tid & 7, rr = tid >> 3;
  i32x4 ras[2][AI], rbs[2][4];
  const T* arow[AI];
  const T* brow[4];
#pragma unroll
  for (int i = 0; i < AI; ++i) arow[i] = a_row<T>(P, min(m0 + rr + 32 * i, P.M - 1));
#pragma unroll
  for (int i = 0; i < 4; ++i) brow[i] = Wt + (long)min(n0 + rr + 32 * i, P.N - 1) * K;
  f32x4 lng[2][LN ? EPC / 4 : 1], lnb[2][LN ? EPC / 4 : 1];      // LN: the slab's gamma / beta of this thread's chunk column ride with it
  auto fetch = [&](auto setc, int k0) {
    constexpr int S = decltype(setc)::value;
    const int k = min(k0 + cc * EPC, K - EPC);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < AI) ras[S][i < AI ? i : 0] = *reinterpret_cast<const i32x4*>(arow[i < AI ? i : 0] + k);
      rbs[S][i] = *reinterpret_cast<const i32x4*>(brow[i] + k);
    }
    if constexpr (LN) {
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q) {
        lng[S][q] = *reinterpret_cast<const f32x4*>(P.gamma + k + 4 * q);
        lnb[S][q] = *reinterpret_cast<const f32x4*>(P.beta + k + 4 * q);
      }
    }
  };
  auto stash = [&](auto setc, int k0) {
    constexpr int S = decltype(setc)::value;
    const int k = k0 + cc * EPC;
    const bool kok = k < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 32 * i;
      const int off = r * ROWB + ((cc << 4) ^ swz128(r));
      *reinterpret_cast<i32x4*>(Bs + off) = kok ? rbs[S][i] : (i32x4)(0);
      if (i >= AI) continue;
      i32x4 va = ras[S][i < AI ? i : 0];
      if constexpr (LN) {
        float f[EPC];
        chunk_to_f32<T>(va, f);
        const float mu = mean_s[r], rs = rstd_s[r];
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = (f[e] - mu) * rs * lng[S][e >> 2][e & 3] + lnb[S][e >> 2][e & 3];
        va = f32_to_chunk<T>(f);
        // (training: the rows as normalised here are what the weight gradient multiplies -- column tile 0 keeps them, so the
        //  backward's GEMM reads a plain operand instead of re-normalising per output tile)
        if (P.An != nullptr && bn == 0 && kok && m0 + r < P.M)
          *reinterpret_cast<i32x4*>(reinterpret_cast<T*>(P.An) + (long)(m0 + r) * P.ldan + k) = va;
      } else if constexpr (PRO == 2) {
        float f[EPC];
        chunk_to_f32<T>(va, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = gelu_erf(f[e]);
        va = f32_to_chunk<T>(f);
      }
      *reinterpret_cast<i32x4*>(As + off) = va;
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x16)(0.f);

  const int wr = BM == 128 ? (wave >> 1) * 64 : 0, wc = BM == 128 ? (wave & 1) * 64 : wave * 32;
  const int l31 = lane & 31, hh = lane >> 5;

  auto compute = [&]() {
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        Frag8<T> af[2], bf[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wr + 32 * i + l31;
          const char* row = As + r * ROWB;
          const int sw = swz128(r);
          const int b0 = (kk * 16 + hh * 8) * (int)sizeof(T);
          if constexpr (sizeof(T) == 2) {
            af[i].v = *reinterpret_cast<const s16x8*>(row + (b0 ^ sw));
          } else {
            const f32x4 x = *reinterpret_cast<const f32x4*>(row + (b0 ^ sw));
            const f32x4 y = *reinterpret_cast<const f32x4*>(row + ((b0 + 16) ^ sw));
#pragma unroll
            for (int e = 0; e < 4; ++e) { af[i].v[e] = x[e]; af[i].v[4 + e] = y[e]; }
          }
          if (i < NJ) {
            const int rn = wc + 32 * i + l31;
            const char* rowb = Bs + rn * ROWB;
            const int swb = swz128(rn);
            if constexpr (sizeof(T) == 2) {
              bf[i < NJ ? i : 0].v = *reinterpret_cast<const s16x8*>(rowb + (b0 ^ swb));
            } else {
              const f32x4 x = *reinterpret_cast<const f32x4*>(rowb + (b0 ^ swb));
              const f32x4 y = *reinterpret_cast<const f32x4*>(rowb + ((b0 + 16) ^ swb));
#pragma unroll
              for (int e = 0; e < 4; ++e) { bf[i < NJ ? i : 0].v[e] = x[e]; bf[i < NJ ? i : 0].v[4 + e] = y[e]; }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) mma32(acc[i][j], af[i], bf[j]);
      }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  bool k_done = false;
  if constexpr (DMA_FORM) {
    if (P.dma) {
      // ---- small-M GEMMs (config 5: 3 072 rows per GPU; the last-frame logits): with one workgroup per CU the register-staged loop
      // below is a chain of memory round trips (~2 us per 64-wide slab, 16 us for K = 512).  Here the slabs travel by LDS-DMA into a
      // ring of three (two in flight while one is multiplied), the XOR swizzle applied on the SOURCE side (lane p of a 1 KB piece
      // fetches the chunk that belongs at LDS position p), one raw barrier per slab; the fragment reads are inline asm with counted
      // waits -- a compiler-visible LDS read behind a pending LDS-DMA gets a vmcnt(0) in front, i.e. drains the prefetch.
      using S2 = std::integral_constant<int, 2>;
      const int lr = lane >> 3, lc = lane & 7;
      const char* asrc[2];
      const char* bsrc[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 8 * (wave + 4 * q) + lr;
        asrc[q] = reinterpret_cast<const char*>(a_row<T>(P, min(m0 + r, P.M - 1))) + ((lc ^ (swz128(r) >> 4)) << 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 8 * (wave + 4 * q) + lr;
        bsrc[q] = reinterpret_cast<const char*>(Wt + (long)min(n0 + r, P.N - 1) * K) + ((lc ^ (swz128(r) >> 4)) << 4);
      }
      auto issue = [&](auto bufc, int sl) {
        constexpr int B = decltype(bufc)::value;
        char* dst = tiles + B * SLAB;
        const long ko = (long)sl * ROWB;
#pragma unroll
        for (int q = 0; q < 2; ++q)
          __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + ko), (lptr_t)(dst + (wave + 4 * q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[q] + ko), (lptr_t)(dst + BM * ROWB + (wave + 4 * q) * 1024), 16, 0, 0);
      };
      // fragment addresses: row l31 (+ 32 for the second A block; + 32 wave for this wave's weight rows: the swizzle of row r + 32 is
      // that of r), byte (32 kk + 16 hh) ^ swizzle -- the XOR is not an immediate, so one address per k-step
      unsigned aA[KSTEPS], aB[KSTEPS];
      {
        const unsigned base = lds_addr(tiles) + (unsigned)l31 * ROWB;
        const int t = (hh * 16) ^ swz128(l31);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
          aA[kk] = base + (unsigned)((kk * 32) ^ t);
          aB[kk] = aA[kk] + (unsigned)(BM * ROWB + wave * 32 * ROWB);
        }
      }
      auto step = [&](auto bufc, int sl, int nsl) {
        constexpr int B = decltype(bufc)::value;
        if (sl + 1 < nsl) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // (six requests per wave and slab)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (sl + 2 < nsl) issue(std::integral_constant<int, (B + 2) % 3>{}, sl + 2);
        s16x8 fa[KSTEPS], fb[KSTEPS], fw[KSTEPS];
        static_for<KSTEPS>([&](auto kc) {
          constexpr int kk = decltype(kc)::value;
          fa[kk] = ds_read_b128_asm<B * SLAB>(aA[kk]);
          fb[kk] = ds_read_b128_asm<B * SLAB + 32 * ROWB>(aA[kk]);
          fw[kk] = ds_read_b128_asm<B * SLAB>(aB[kk]);
        });
        static_for<KSTEPS>([&](auto kc) {
          constexpr int kk = decltype(kc)::value;
          lgkm_wait_for3<3 * (KSTEPS - 1 - kk)>(fa[kk], fb[kk], fw[kk]);
          Frag8<T> a0, a1, w;
          a0.v = fa[kk]; a1.v = fb[kk]; w.v = fw[kk];
          mma32(acc[0][0], a0, w);
          mma32(acc[1][0], a1, w);
        });
      };
      const int nsl = K / BK;
      issue(S0{}, 0);
      if (nsl > 1) issue(S1{}, 1);
      for (int sl = 0; sl < nsl; sl += 3) {
        step(S0{}, sl, nsl);
        if (sl + 1 < nsl) step(S1{}, sl + 1, nsl);
        if (sl + 2 < nsl) step(S2{}, sl + 2, nsl);
      }
      k_done = true;
    }
  }
  if (!k_done) {
  fetch(S0{}, 0);
  fetch(S1{}, BK);                                       // (a slab past K re-reads the last chunk column; it is never stashed)
  for (int k0 = 0; k0 < K; k0 += 2 * BK) {
    __syncthreads();
    stash(S0{}, k0);
    __syncthreads();
    fetch(S0{}, k0 + 2 * BK);
    compute();
    // (no branch around the second half: a slab past K has its weight chunks zeroed and adds nothing -- a conditional region
    //  with loads inside costs the counted waits)
    __syncthreads();
    stash(S1{}, k0 + BK);
    __syncthreads();
    fetch(S1{}, k0 + 3 * BK);
    compute();
  }
  }

  // epilogue: D col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const bool gelu = (P.flags & WMZ_LIN_GELU) != 0;
  const bool dgelu = (P.flags & WMZ_LIN_DGELU) != 0;
  const T* R = reinterpret_cast<const T*>(P.res);
  if constexpr (sizeof(T) == 2) {
    // (fp32 outputs -- the logits -- leave the same way, as two 16-byte stores per chunk, when there is no residual operand)
    const bool f32_staged = P.out_f32 && R == nullptr && P.C2 == nullptr && (P.ldc % 4) == 0;
    if (f32_staged || (!P.out_f32 && (P.ldc % 8) == 0 && (R == nullptr || (P.ldr % 8) == 0) && (P.C2 == nullptr || (P.ldc2 % 8) == 0))) {
      // 16-bit outputs: a lane owns ONE column, so direct stores would be 2 bytes each.  Stage the fp32 tile through LDS
      // (BM / 2 rows per round: it fits the bytes the slabs occupied) and leave as whole 16-byte row chunks; the residual /
      // gelu' operand is read the same way and applied in fp32 before the single rounding.
      constexpr int RR = BM / 2;                                       // rows per round
      float* stage = reinterpret_cast<float*>(tiles);                 // [RR][128] fp32
#pragma unroll 1
      for (int round = 0; round < 2; ++round) {
        __syncthreads();
        if (BM == 64 || (wave >> 1) == round) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (BM == 64 && i != round) continue;                     // BM = 64: every wave holds rows 32 i .. 32 i + 31 in acc[i]
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const int cl = wc + 32 * j + l31;
              const int col = n0 + cl;
              const float bv = (P.bias && col < P.N) ? P.bias[col] : 0.f;
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                const int rl = (BM == 64 ? 0 : 32 * i) + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                float v = acc[i][j][reg] + bv;
                if (gelu) v = gelu_erf(v);
                stage[rl * BN + cl] = v;
              }
            }
          }
        }
        __syncthreads();
        // RR rows x 16 chunks of 8 columns, RR / 16 per thread
#pragma unroll
        for (int it = 0; it < RR / 16; ++it) {
          const int idx = tid + it * NT;
          const int rl = idx >> 4, ch = idx & 15;
          const int row = m0 + round * RR + rl, col = n0 + ch * 8;
          if (row >= P.M || col >= P.N) continue;
          const f32x4 a = *reinterpret_cast<const f32x4*>(stage + rl * BN + ch * 8);
          const f32x4 b = *reinterpret_cast<const f32x4*>(stage + rl * BN + ch * 8 + 4);
          float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
          if (f32_staged) {
            float* dstf = reinterpret_cast<float*>(P.C) + (long)row * P.ldc + col;
            if (col + 8 <= P.N) {
              *reinterpret_cast<f32x4*>(dstf) = a;
              *reinterpret_cast<f32x4*>(dstf + 4) = b;
            } else {
              for (int e = 0; e < 8 && col + e < P.N; ++e) dstf[e] = f[e];
            }
            continue;
          }
          T* dst = reinterpret_cast<T*>(P.C) + (long)row * P.ldc + col;
          if (col + 8 <= P.N) {
            if (R) {
              float r8[8];
              chunk_to_f32<T>(*reinterpret_cast<const i32x4*>(R + (long)row * P.ldr + col), r8);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                if (dgelu) f[e] *= wmz_dgelu(r8[e]);
                else f[e] += r8[e];
              }
            }
            *reinterpret_cast<i32x4*>(dst) = f32_to_chunk<T>(f);
            if (P.C2) {                                                // the activation next to the pre-activation
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] = gelu_erf(f[e]);
              *reinterpret_cast<i32x4*>(reinterpret_cast<T*>(P.C2) + (long)row * P.ldc2 + col) = f32_to_chunk<T>(f);
            }
          } else {
            for (int e = 0; e < 8 && col + e < P.N; ++e) {
              float v = f[e];
              if (R) {
                const float rv = Elem<T>::to_f32(R[(long)row * P.ldr + col + e]);
                if (dgelu) v *= wmz_dgelu(rv);
                else v += rv;
              }
              dst[e] = Elem<T>::from_f32(v);
              if (P.C2) reinterpret_cast<T*>(P.C2)[(long)row * P.ldc2 + col + e] = Elem<T>::from_f32(gelu_erf(v));
            }
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wc + 32 * j + l31;
      if (col >= P.N) continue;
      const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wr + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (row >= P.M) continue;
        float v = acc[i][j][reg] + bv;
        if (gelu) v = gelu_erf(v);
        if (R) {
          const float rv = Elem<T>::to_f32(R[(long)row * P.ldr + col]);
          if (dgelu) {   // rv is the saved pre-activation z: multiply by gelu'(z)
            v *= wmz_dgelu(rv);
          } else {
            v += rv;
          }
        }
        if (P.out_f32) reinterpret_cast<float*>(P.C)[(long)row * P.ldc + col] = v;
        else reinterpret_cast<T*>(P.C)[(long)row * P.ldc + col] = Elem<T>::from_f32(v);
        if (P.C2) reinterpret_cast<T*>(P.C2)[(long)row * P.ldc2 + col] = Elem<T>::from_f32(gelu_erf(v));
      }
    }
}

}  // namespace

extern "C" int WMZ_FN(wmz_linear_fwd)(const void* A, long lda, const void* Wt, const float* bias, const void* residual,
                              long ldr, void* C, long ldc, int M, int N, int K, const float* ln_gamma,
                              const float* ln_beta, float ln_eps, int flags, int out_f32, int dtype, void* stream) {
  return WMZ_FN(wmz_linear_fwd_stats)(A, lda, Wt, bias, residual, ldr, C, ldc, M, N, K, ln_gamma, ln_beta, nullptr, nullptr, ln_eps,
                              flags, out_f32, dtype, stream);
}

static int g_linear_dma = 1;
#ifndef WMZ_OP16_F16
extern "C" int wmz_debug_linear_knobs(int dma) { g_linear_dma = dma; return WMZ_OK; }      // (development A/B: 0 = the register-staged loop)
#endif

static int linear_launch(LinParams P, const float* ln_gamma, int flags, int dtype, hipStream_t st) {
  const bool ln = ln_gamma != nullptr;
  const bool gin = (flags & WMZ_LIN_GELU_IN) != 0;
  WMZ_REQUIRE(!(ln && gin), "wmz_linear_fwd: LayerNorm and GELU prologues are exclusive");
  P.nbn = wmz_cdiv(P.N, BN);
  // small-M GEMMs (the last-frame logits: M = B*H*W; config 5's 3 072 tokens per GPU): 64-row tiles, so that the grid covers
  // the chip (a 16-bit output then leaves by per-lane stores: fine at these sizes)
  const bool small = (long)wmz_cdiv(P.M, 128) * P.nbn < (P.out_f32 ? 192 : 320) && (dtype == kOp16Dtype || (!ln && !gin));
  const int bm = small ? 64 : 128;
  const uintptr_t al = (uintptr_t)P.A | (uintptr_t)P.Wt | (uintptr_t)(P.lda * 2) | (uintptr_t)(P.bstride * 2);
  P.dma = (small && dtype == kOp16Dtype && !ln && !gin && P.K % 64 == 0 && (al & 15) == 0 && g_linear_dma) ? 1 : 0;
  dim3 grid((unsigned)(wmz_cdiv(P.M, bm) * P.nbn)), block(NT);
  if (dtype == kOp16Dtype) {
    if (small && ln) hipLaunchKernelGGL((linear_kernel<bf16_t, 1, 64>), grid, block, 0, st, P);
    else if (small && gin) hipLaunchKernelGGL((linear_kernel<bf16_t, 2, 64>), grid, block, 0, st, P);
    else if (small) hipLaunchKernelGGL((linear_kernel<bf16_t, 0, 64>), grid, block, 0, st, P);
    else if (ln) hipLaunchKernelGGL((linear_kernel<bf16_t, 1, 128>), grid, block, 0, st, P);
    else if (gin) hipLaunchKernelGGL((linear_kernel<bf16_t, 2, 128>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((linear_kernel<bf16_t, 0, 128>), grid, block, 0, st, P);
  } else {
    if (small) hipLaunchKernelGGL((linear_kernel<float, 0, 64>), grid, block, 0, st, P);
    else if (ln) hipLaunchKernelGGL((linear_kernel<float, 1, 128>), grid, block, 0, st, P);
    else if (gin) hipLaunchKernelGGL((linear_kernel<float, 2, 128>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((linear_kernel<float, 0, 128>), grid, block, 0, st, P);
  }
  WMZ_LAUNCH_CHECK("wmz_linear_fwd");
  return WMZ_OK;
}

extern "C" int WMZ_FN(wmz_linear_fwd_stats)(const void* A, long lda, const void* Wt, const float* bias, const void* residual,
                                    long ldr, void* C, long ldc, int M, int N, int K, const float* ln_gamma,
                                    const float* ln_beta, const float* ln_mean, const float* ln_rstd, float ln_eps,
                                    int flags, int out_f32, int dtype, void* stream) {
  WMZ_REQUIRE(A && Wt && C, "wmz_linear_fwd: null tensor");
  WMZ_REQUIRE((ln_mean == nullptr) == (ln_rstd == nullptr), "wmz_linear_fwd: ln_mean and ln_rstd go together");
  WMZ_REQUIRE(ln_mean == nullptr || ln_gamma != nullptr, "wmz_linear_fwd: statistics without a LayerNorm prologue");
  WMZ_REQUIRE(M > 0 && N > 0 && K > 0, "wmz_linear_fwd: bad shape M=%d N=%d K=%d", M, N, K);
  WMZ_REQUIRE(K % 8 == 0 && lda % 8 == 0, "wmz_linear_fwd: K and lda must be multiples of 8 (K=%d lda=%ld)", K, lda);
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == kOp16Dtype, "wmz_linear_fwd: bad dtype %d", dtype);
  WMZ_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "wmz_linear_fwd: ln_gamma and ln_beta go together");
  LinParams P;
  P.A = A; P.lda = lda; P.Wt = Wt; P.bias = bias; P.res = residual; P.ldr = ldr; P.C = C; P.ldc = ldc;
  P.M = M; P.N = N; P.K = K; P.gamma = ln_gamma; P.beta = ln_beta; P.eps = ln_eps; P.flags = flags;
  P.mean = ln_mean; P.rstd = ln_rstd;
  P.out_f32 = out_f32;
  P.rpb = 0; P.bstride = 0;
  P.C2 = nullptr; P.ldc2 = 0;
  P.An = nullptr; P.ldan = 0;
  return linear_launch(P, ln_gamma, flags, dtype, (hipStream_t)stream);
}

#ifndef WMZ_OP16_F16      // (the training forward's GEMMs: the bfloat16 / fp32 unit only)
// FeedForward's first GEMM in training (local_3d_attention.py:24-25: Linear -> GELU): Z = LN?(A) Wt^T + bias AND H = GELU(Z),
// both in the activation dtype, from one accumulator (H rounds GELU of the fp32 sum, as the inference epilogue does).  The
// second GEMM and its weight gradient then read H as it is -- with the activation recomputed in their loaders every column
// tile of the output repeats the erf of its whole operand panel (dim 384: three times; the 512-wide mlp of config 5: four).
extern "C" int wmz_linear_fwd_gelu_pair(const void* A, long lda, const void* Wt, const float* bias, void* Z, long ldz, void* H,
                                        long ldh, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                                        const float* ln_mean, const float* ln_rstd, float ln_eps, int dtype, void* stream) {
  WMZ_REQUIRE(A && Wt && Z && H, "wmz_linear_fwd_gelu_pair: null tensor");
  WMZ_REQUIRE((ln_mean == nullptr) == (ln_rstd == nullptr), "wmz_linear_fwd_gelu_pair: ln_mean and ln_rstd go together");
  WMZ_REQUIRE(ln_mean == nullptr || ln_gamma != nullptr, "wmz_linear_fwd_gelu_pair: statistics without a LayerNorm prologue");
  WMZ_REQUIRE(M > 0 && N > 0 && K > 0, "wmz_linear_fwd_gelu_pair: bad shape M=%d N=%d K=%d", M, N, K);
  WMZ_REQUIRE(K % 8 == 0 && lda % 8 == 0, "wmz_linear_fwd_gelu_pair: K and lda must be multiples of 8 (K=%d lda=%ld)", K, lda);
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == kOp16Dtype, "wmz_linear_fwd_gelu_pair: bad dtype %d", dtype);
  WMZ_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "wmz_linear_fwd_gelu_pair: ln_gamma and ln_beta go together");
  LinParams P;
  P.A = A; P.lda = lda; P.Wt = Wt; P.bias = bias; P.res = nullptr; P.ldr = 0; P.C = Z; P.ldc = ldz;
  P.M = M; P.N = N; P.K = K; P.gamma = ln_gamma; P.beta = ln_beta; P.eps = ln_eps; P.flags = 0;
  P.mean = ln_mean; P.rstd = ln_rstd;
  P.out_f32 = 0;
  P.rpb = 0; P.bstride = 0;
  P.C2 = H; P.ldc2 = ldh;
  P.An = nullptr; P.ldan = 0;
  return linear_launch(P, ln_gamma, 0, dtype, (hipStream_t)stream);
}

// The training forward's PreNorm GEMM in full: C = LN(A) Wt^T + bias, optionally H = GELU(C) next to it (FeedForward's first
// GEMM) and optionally An = LN(A) [M, K] as the GEMM consumed it -- the operand of the layer's weight gradient, which then is a
// plain GEMM (the LayerNorm prologue in the weight gradient re-normalises the operand panel once per output column tile and
// keeps it off the 256-wide tiles of wgrad3_kernel).  Statistics: supplied (ln_mean / ln_rstd) or computed by the prologue.
extern "C" int wmz_linear_fwd_train(const void* A, long lda, const void* Wt, const float* bias, void* C, long ldc, void* H, long ldh,
                                    void* An, long ldan, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                                    const float* ln_mean, const float* ln_rstd, float ln_eps, int dtype, void* stream) {
  WMZ_REQUIRE(A && Wt && C && ln_gamma && ln_beta, "wmz_linear_fwd_train: null tensor (a LayerNorm prologue is required)");
  WMZ_REQUIRE((ln_mean == nullptr) == (ln_rstd == nullptr), "wmz_linear_fwd_train: ln_mean and ln_rstd go together");
  WMZ_REQUIRE(M > 0 && N > 0 && K > 0, "wmz_linear_fwd_train: bad shape M=%d N=%d K=%d", M, N, K);
  WMZ_REQUIRE(K % 8 == 0 && lda % 8 == 0 && (An == nullptr || ldan % 8 == 0), "wmz_linear_fwd_train: K, lda, ldan must be multiples of 8");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == kOp16Dtype, "wmz_linear_fwd_train: bad dtype %d", dtype);
  LinParams P;
  P.A = A; P.lda = lda; P.Wt = Wt; P.bias = bias; P.res = nullptr; P.ldr = 0; P.C = C; P.ldc = ldc;
  P.M = M; P.N = N; P.K = K; P.gamma = ln_gamma; P.beta = ln_beta; P.eps = ln_eps; P.flags = 0;
  P.mean = ln_mean; P.rstd = ln_rstd;
  P.out_f32 = 0;
  P.rpb = 0; P.bstride = 0;
  P.C2 = H; P.ldc2 = ldh;
  P.An = An; P.ldan = ldan;
  return linear_launch(P, ln_gamma, 0, dtype, (hipStream_t)stream);
}

#endif  // WMZ_OP16_F16

// logit_proj on the LAST FRAME of every clip (main.py:35-36: x[:, -1] -> nn.Linear), read in place: A's rows come in blocks of
// rows_per_block (= H*W) rows lda apart, the blocks block_stride apart (= S*H*W*D for the last plane of each clip).
extern "C" int WMZ_FN(wmz_linear_fwd_blocked)(const void* A, long lda, int rows_per_block, long block_stride, const void* Wt,
                                      const float* bias, void* C, long ldc, int M, int N, int K, int out_f32, int dtype,
                                      void* stream) {
  WMZ_REQUIRE(A && Wt && C, "wmz_linear_fwd_blocked: null tensor");
  WMZ_REQUIRE(M > 0 && N > 0 && K > 0 && rows_per_block > 0, "wmz_linear_fwd_blocked: bad shape M=%d N=%d K=%d rows_per_block=%d", M, N, K, rows_per_block);
  WMZ_REQUIRE(K % 8 == 0 && lda % 8 == 0 && block_stride % 8 == 0, "wmz_linear_fwd_blocked: K, lda, block_stride must be multiples of 8");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == kOp16Dtype, "wmz_linear_fwd_blocked: bad dtype %d", dtype);
  LinParams P;
  P.A = A; P.lda = lda; P.Wt = Wt; P.bias = bias; P.res = nullptr; P.ldr = 0; P.C = C; P.ldc = ldc;
  P.M = M; P.N = N; P.K = K; P.gamma = nullptr; P.beta = nullptr; P.eps = 0.f; P.flags = 0;
  P.mean = nullptr; P.rstd = nullptr;
  P.out_f32 = out_f32;
  P.rpb = rows_per_block; P.bstride = block_stride;
  P.C2 = nullptr; P.ldc2 = 0;
  P.An = nullptr; P.ldan = 0;
  return linear_launch(P, nullptr, 0, dtype, (hipStream_t)stream);
}
