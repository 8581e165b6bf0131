// One launch per transformer layer boundary for everything that is per-token (bf16 speed path):
//
//   HEAD:  x1 = o Wout^T + bout + x                       (to_out + residual,        local_3d_attention.py:50-53, :160)
//          x2 = W2 GELU(W1 LN2(x1) + b1) + b2 + x1        (PreNorm(FeedForward) + x, :11-31, :161)
//   TAIL:  q' = Wq' x2 ,  k'|v' = Wkv' LN1'(x2) + bkv'    (NEXT layer's to_q on the raw stream and to_k|to_v on
//                                                          LayerNorm(x): quirk Q1, :16-17, :46-48, :106-108)
//
// Design (MI355X): a workgroup = 8 waves = 128 tokens; each wave owns 16 tokens end to end, so the only thing the waves
// share is the weight stream.  Every GEMM is computed TRANSPOSED, D^T[features x 16 tokens] = W[features x K] . act^T, with
// MFMA 16x16x32 bf16: A = weight rows, B = the wave's activation rows -- both K-contiguous row fragments, no transposes --
// and the token on the lane.  Consequences: the residual stream x1/x2 stays in fp32 REGISTERS across the whole chain
// (64 VGPRs; two waves per SIMD, so one wave's VALU epilogue (LayerNorm, GELU, packing) overlaps the other's MFMAs);
// LayerNorm statistics are lane-local sums plus two shuffles; each lane holds 4 consecutive features per accumulator
// block, so activations are exchanged through a per-wave 8 KB LDS buffer with 8-byte accesses (no workgroup barrier),
// and every HBM store is a whole row.
// The layer's 512 KB of weights are pre-packed on the host in exactly the order the kernel consumes them (16 KB slabs of
// [32-deep k-step][feature][32 k], 16-byte chunks swizzled against bank conflicts) and streamed by LDS-DMA (global_load_lds) into a 4-slot
// LDS ring with 3 slabs in flight: counted vmcnt + ONE raw s_barrier per slab (= 16 MFMAs per wave).  The feed-forward is
// walked 64 hidden units at a time (W1 rows -> GELU -> W2 columns), so its activation never exists in full.
#include "wmz_common.h"
#include <stdlib.h>

namespace {

constexpr int FT = 16;          // tokens per wave
constexpr int FW = 8;           // waves per workgroup (two per SIMD: one's epilogue overlaps the other's MFMAs)
constexpr int NTHR = FW * 64;
constexpr int SLAB = 16384;     // bytes per weight slab
constexpr int RING = 4;         // LDS ring slots: 3 slabs (48 KB) of the weight stream stay in flight
constexpr int VECB = 8192;      // the layer's bias / LayerNorm vectors (2048 fp32), staged once per workgroup
constexpr int ACTB = 8192;      // per-wave activation buffer (16 tokens x 256 features bf16)
constexpr int ZCB = 2048;       // per-wave feed-forward chunk buffer (16 tokens x 64 features bf16)
constexpr int MC = 64;          // feed-forward hidden chunk: W1 rows / W2 columns streamed MC at a time

struct FusedParams {
  const bf16_t* o;      // [ntok, I]   attention output               (HEAD)
  const bf16_t* x;      // [ntok, D]   residual stream in
  bf16_t* xo;           // [ntok, D]   residual stream out            (HEAD)
  bf16_t* q;            // [ntok, I]                                   (TAIL)
  bf16_t* kv;           // [ntok, 2I]                                  (TAIL)
  const char* wpack;    // packed bf16 weights in streaming order (+ RING-1 slabs of padding)
  const float* vec;     // packed fp32 vectors: bout[D] g2[D] be2[D] b1[M] b2[D] g1n[D] be1n[D] bkv[2I]
  int ntok;
  float eps;
  // EMBED variant (first layer): x is produced in-kernel from the token grid (local_3d_attention.py:140-157)
  const int64_t* z; const float *emb, *pos_s, *pos_h, *pos_w; int S, H, W, num_classes;
  int dbg;              // ablation switches (timing experiments only): 1 = skip MFMA loop, 2 = skip weight DMA + waits
  // trailing-planes variant: token t of the compact output grid reads row (t / rows_out) * rows_in + row0 + t % rows_out
  // of x (or of the token grid z); rows_out == 0: identity
  int rows_out, rows_in, row0;
};
__device__ __forceinline__ long src_row(const FusedParams& P, long t) {
  if (P.rows_out == 0) return t;
  const int c = (int)t / P.rows_out;
  return (long)c * P.rows_in + P.row0 + ((int)t - c * P.rows_out);
}

// 16-byte-chunk XOR swizzle of an activation row: conflict-free ds_read_b128 of 32 token rows at one k-chunk
template <int ROWB> __device__ __forceinline__ int aswz(int token) {
  if constexpr (ROWB >= 256) return (token & 15) << 4;
  else return ((token >> 1) & 7) << 4;          // 128-byte rows: two rows per 256-byte bank line
}
__device__ __forceinline__ float gelu_erf(float v) { return wmz_gelu(v); }

// ---- weight stream: RING-slot LDS ring filled by LDS-DMA (global_load_lds), RING-1 slabs in flight.
struct WStream {
  int dbg;
  const char* src;     // global address of the next slab to ISSUE (this lane's 16 bytes of piece 0 of its wave)
  char* ring;          // LDS ring base + this wave's 4 KB quarter
  int issue_slot;      // ring slot the next issued slab goes to
  int cur;             // ring slot of the slab being multiplied
};

__device__ __forceinline__ void ws_issue(WStream& ws) {
  if (ws.dbg & 2) return;
  char* dst = ws.ring + ws.issue_slot * SLAB;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws.src + i * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  ws.src += SLAB;
  ws.issue_slot = ws.issue_slot == RING - 1 ? 0 : ws.issue_slot + 1;
}

// Before multiplying a slab: this wave's eighth of it has landed (all but the 2*(RING-2) youngest VMEM ops done), then
// one barrier: every piece landed, and every wave is done with the previous slab, whose slot is refilled right away.
__device__ __forceinline__ void ws_acquire(WStream& ws) {
  if (!(ws.dbg & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  static_assert(2 * (RING - 2) == 4, "vmcnt literal above follows RING (2 LDS-DMA pieces per wave per slab)");
  __builtin_amdgcn_s_barrier();
  ws_issue(ws);
}
__device__ __forceinline__ void ws_release(WStream& ws) { ws.cur = ws.cur == RING - 1 ? 0 : ws.cur + 1; }

// acc^T[N x 16 tokens] += W[N x K] . act^T over K/32 k-steps of the stream (MFMA 16x16x32: A = weight rows, B = the
// wave's token rows).  act: this wave's LDS buffer, rows = tokens, AROWB bytes per row.
template <int N, int K, int AROWB>
__device__ __forceinline__ void gemm_stage(f32x4 (&acc)[N / 16], const char* act, const char* ring0, WStream& ws, int li,
                                           int g) {
  constexpr int NB = N / 16;
  constexpr int KPS = SLAB / (N * 64);          // 32-deep k-steps per slab
  constexpr int NSLAB = (K / 32) / KPS;
  static_assert(KPS >= 1 && NSLAB * KPS * 32 == K, "stage depth must be a whole number of slabs");
  const char* arow = act + li * AROWB;
  const int asw = aswz<AROWB>(li);
  const int wchunk = (g ^ ((0 - (li >> 2)) & 3)) * 16;   // physical 16-byte chunk of this lane's weight row (host swizzle)
#pragma unroll 1
  for (int s = 0; s < NSLAB; ++s) {
    ws_acquire(ws);
    const char* slab = ring0 + ws.cur * SLAB;
    if (ws.dbg & 1) { ws_release(ws); continue; }
#pragma unroll
    for (int t = 0; t < KPS; ++t) {
      const int ks = s * KPS + t;
      Frag8<bf16_t> bf;
      bf.v = *reinterpret_cast<const s16x8*>(arow + (((ks * 4 + g) << 4) ^ asw));
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        Frag8<bf16_t> af;
        af.v = *reinterpret_cast<const s16x8*>(slab + t * (N * 64) + (16 * b + li) * 64 + wchunk);
        mma16(acc[b], af, bf);
      }
    }
    ws_release(ws);
  }
}

// accumulator block b of lane group g holds features n0 .. n0+3 of the lane's token
__device__ __forceinline__ int quad_n0(int b, int g) { return 16 * b + 4 * g; }

template <int NB>
__device__ __forceinline__ void add_vec(f32x4 (&acc)[NB], const float* vec, int g) {   // vec offset to feature 0 of acc
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(vec + quad_n0(b, g));
    acc[b] += v;
  }
}

// acc += the lane's token row of a global [ntok, NB*16] bf16 tensor (8-byte loads in the accumulator layout)
template <int NB>
__device__ __forceinline__ void add_row_global(f32x4 (&acc)[NB], const bf16_t* src, long tok, bool ok, int g) {
  const bf16_t* row = src + tok * (NB * 16);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    s16x4 pk = (s16x4)(0);
    if (ok) pk = *reinterpret_cast<const s16x4*>(row + quad_n0(b, g));
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[b][r] += bf16_bits_to_f32((unsigned short)pk[r]);
  }
}

// write the wave's [16 tokens x NB*16 features] tile (bf16) into columns n_off.. of an LDS buffer with ROWF-feature rows
template <int NB, int ROWF, typename F>
__device__ __forceinline__ void tile_to_lds(char* act, const f32x4 (&acc)[NB], int n_off, int li, int g, F f) {
  constexpr int ROWB = ROWF * 2;
  char* row = act + li * ROWB;
  const int sw = aswz<ROWB>(li);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n0 = n_off + quad_n0(b, g);
    s16x4 pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(f(acc[b][r]));
    *reinterpret_cast<s16x4*>(row + ((n0 * 2) ^ sw)) = pk;
  }
}

// LayerNorm of the token held by this lane quartet (features split over the 4 lane groups) -> bf16 -> LDS buffer
template <int NB>
__device__ __forceinline__ void ln_to_lds(char* act, const f32x4 (&acc)[NB], const float* gamma, const float* beta,
                                          float eps, int li, int g) {
  constexpr int NF = NB * 16, ROWB = NF * 2;
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) s += (acc[b][0] + acc[b][1]) + (acc[b][2] + acc[b][3]);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float mean = s / (float)NF;
  float q = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = acc[b][r] - mean; q = fmaf(d, d, q); }
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  const float rstd = rsqrtf(q / (float)NF + eps);
  char* row = act + li * ROWB;
  const int sw = aswz<ROWB>(li);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n0 = quad_n0(b, g);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + n0);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + n0);
    s16x4 pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits((acc[b][r] - mean) * rstd * gm[r] + be[r]);
    *reinterpret_cast<s16x4*>(row + ((n0 * 2) ^ sw)) = pk;
  }
}

// global [tok0.., F] bf16 rows -> the wave's LDS buffer (coalesced 16-byte chunks); rows >= ntok are zero
template <int F>
__device__ __forceinline__ void rows_to_lds(char* act, const bf16_t* src, long tok0, int ntok, int lane) {
  constexpr int ROWB = F * 2, CPR = ROWB / 16, TOT = FT * CPR;
#pragma unroll
  for (int i = 0; i < TOT / 64; ++i) {
    const int idx = lane + 64 * i;
    const int r = idx / CPR, c = idx - r * CPR;
    i32x4 v = (i32x4)(0);
    if (tok0 + r < ntok) v = *reinterpret_cast<const i32x4*>(src + (tok0 + r) * F + c * 8);
    *reinterpret_cast<i32x4*>(act + r * ROWB + ((c << 4) ^ aswz<ROWB>(r))) = v;
  }
}
// same, split: issue the global loads early (registers), write them to LDS late (T14: the GEMM in between hides them)
template <int F>
__device__ __forceinline__ void rows_fetch(i32x4 (&regs)[FT * (F * 2 / 16) / 64], const bf16_t* src, long tok0, int ntok, int lane,
                                           const FusedParams& P) {
  constexpr int CPR = F * 2 / 16, TOT = FT * CPR;
#pragma unroll
  for (int i = 0; i < TOT / 64; ++i) {
    const int idx = lane + 64 * i;
    const int r = idx / CPR, c = idx - r * CPR;
    regs[i] = (i32x4)(0);
    if (tok0 + r < ntok) regs[i] = *reinterpret_cast<const i32x4*>(src + src_row(P, tok0 + r) * F + c * 8);
  }
}
template <int F>
__device__ __forceinline__ void rows_put(char* act, const i32x4 (&regs)[FT * (F * 2 / 16) / 64], int lane) {
  constexpr int ROWB = F * 2, CPR = ROWB / 16, TOT = FT * CPR;
#pragma unroll
  for (int i = 0; i < TOT / 64; ++i) {
    const int idx = lane + 64 * i;
    const int r = idx / CPR, c = idx - r * CPR;
    *reinterpret_cast<i32x4*>(act + r * ROWB + ((c << 4) ^ aswz<ROWB>(r))) = regs[i];
  }
}
// the wave's LDS buffer -> global rows, whole rows per store instruction
template <int F>
__device__ __forceinline__ void lds_to_rows(bf16_t* dst, const char* act, long tok0, int ntok, int lane) {
  constexpr int ROWB = F * 2, CPR = ROWB / 16, TOT = FT * CPR;
#pragma unroll
  for (int i = 0; i < TOT / 64; ++i) {
    const int idx = lane + 64 * i;
    const int r = idx / CPR, c = idx - r * CPR;
    const i32x4 v = *reinterpret_cast<const i32x4*>(act + r * ROWB + ((c << 4) ^ aswz<ROWB>(r)));
    if (tok0 + r < ntok) *reinterpret_cast<i32x4*>(dst + (tok0 + r) * F + c * 8) = v;
  }
}
// read the lane's token row back from LDS into the accumulator layout (fp32)
template <int NB, bool ADD>
__device__ __forceinline__ void lds_to_acc(f32x4 (&acc)[NB], const char* act, int li, int g) {
  constexpr int ROWB = NB * 16 * 2;
  const char* row = act + li * ROWB;
  const int sw = aswz<ROWB>(li);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const s16x4 pk = *reinterpret_cast<const s16x4*>(row + ((quad_n0(b, g) * 2) ^ sw));
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[b][r] = (ADD ? acc[b][r] : 0.f) + bf16_bits_to_f32((unsigned short)pk[r]);
  }
}

template <int N> __device__ __forceinline__ void zero_acc(f32x4 (&acc)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) acc[b] = (f32x4)(0.f);
}

// token + 3-axis position embedding of the wave's 16 tokens -> LDS buffer (bf16, the tail's operand) and x_out (bf16)
template <int F>
__device__ __forceinline__ void embed_rows(char* act, const FusedParams& P, long tok0, int lane) {
  constexpr int ROWB = F * 2, CPR = ROWB / 16, TOT = FT * CPR;
#pragma unroll
  for (int i = 0; i < TOT / 64; ++i) {
    const int idx = lane + 64 * i;
    const int r = idx / CPR, c = idx - r * CPR;
    const long t = tok0 + r;
    i32x4 v = (i32x4)(0);
    if (t < P.ntok) {
      const long ts = src_row(P, t);
      const int w = (int)(ts % P.W), h = (int)((ts / P.W) % P.H), s = (int)((ts / ((long)P.W * P.H)) % P.S);
      long tk = P.z[ts];
      tk = tk < 0 ? 0 : (tk >= P.num_classes ? P.num_classes - 1 : tk);
      float f[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int col = c * 8 + 4 * q;
        const f32x4 e = *reinterpret_cast<const f32x4*>(P.emb + tk * F + col);
        const f32x4 a = *reinterpret_cast<const f32x4*>(P.pos_s + (long)s * F + col);
        const f32x4 b = *reinterpret_cast<const f32x4*>(P.pos_h + (long)h * F + col);
        const f32x4 d = *reinterpret_cast<const f32x4*>(P.pos_w + (long)w * F + col);
        const f32x4 y = e + ((a + b) + d);
#pragma unroll
        for (int k = 0; k < 4; ++k) f[4 * q + k] = y[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        v[k] = (int)((unsigned)f32_to_bf16_bits(f[2 * k]) | ((unsigned)f32_to_bf16_bits(f[2 * k + 1]) << 16));
      *reinterpret_cast<i32x4*>(P.xo + t * F + c * 8) = v;
    }
    *reinterpret_cast<i32x4*>(act + r * ROWB + ((c << 4) ^ aswz<ROWB>(r))) = v;
  }
}

template <int D, int I, int M, bool HEAD, bool TAIL>
__global__ __launch_bounds__(NTHR, 2) void layer_fused_kernel(FusedParams P) {
  static_assert(D == 256 && M == 256 && I == 128, "built for the default denoiser widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  char* actA = smem + wave * ACTB;
  char* zc = smem + FW * ACTB + wave * ZCB;
  float* vecs = reinterpret_cast<float*>(smem + FW * (ACTB + ZCB));
  const char* ring0 = smem + FW * (ACTB + ZCB) + VECB;
  WStream ws;
  ws.ring = smem + FW * (ACTB + ZCB) + VECB + wave * 2048;
  ws.src = P.wpack + wave * 2048 + lane * 16;
  ws.issue_slot = 0;
  ws.cur = 0;
  ws.dbg = P.dbg;
  const long tok0 = (long)blockIdx.x * (FT * FW) + wave * FT;
  const bool tok_ok = tok0 + li < P.ntok;
  // per-feature vectors: one coalesced copy into LDS (reading them from L2 inside the epilogues exposed ~150 dependent
  // load latencies per wave)
  *reinterpret_cast<f32x4*>(vecs + tid * 4) = *reinterpret_cast<const f32x4*>(P.vec + tid * 4);
  static_assert(NTHR * 4 == 7 * 256 + 256, "vector block is 2048 floats");
  const float* v_bout = vecs;
  const float* v_g2 = v_bout + D;
  const float* v_be2 = v_g2 + D;
  const float* v_b1 = v_be2 + D;
  const float* v_b2 = v_b1 + M;
  const float* v_g1n = v_b2 + D;
  const float* v_be1n = v_g1n + D;
  const float* v_bkv = v_be1n + D;

  __syncthreads();                                   // vectors visible
#pragma unroll
  for (int i = 0; i < RING - 1; ++i) ws_issue(ws);   // prime: RING-1 slabs in flight

  f32x4 xr[D / 16];                          // the residual stream of this lane's token (1/4 of its features), fp32
  if constexpr (HEAD) {
    rows_to_lds<I>(actA, P.o, tok0, P.ntok, lane);
    i32x4 xpre[FT * (D * 2 / 16) / 64];
    rows_fetch<D>(xpre, P.x, tok0, P.ntok, lane, P);                      // residual rows: in flight under the first GEMM
    zero_acc(xr);
    gemm_stage<D, I, I * 2>(xr, actA, ring0, ws, li, g);               // o Wout^T
    add_vec<D / 16>(xr, v_bout, g);
    rows_put<D>(actA, xpre, lane);                                     // (o tile is dead) x rows -> LDS, coalesced
    lds_to_acc<D / 16, true>(xr, actA, li, g);                         // + x            -> x1
    ln_to_lds<D / 16>(actA, xr, v_g2, v_be2, P.eps, li, g);            // LN2(x1) -> actA
#pragma unroll 1
    for (int c = 0; c < M / MC; ++c) {                                 // feed-forward, MC hidden units at a time
      f32x4 z[MC / 16];
      zero_acc(z);
      gemm_stage<MC, D, D * 2>(z, actA, ring0, ws, li, g);             // W1[c] LN2(x1)
      add_vec<MC / 16>(z, v_b1 + c * MC, g);
      tile_to_lds<MC / 16, MC>(zc, z, 0, li, g, [](float v) { return gelu_erf(v); });
      gemm_stage<D, MC, MC * 2>(xr, zc, ring0, ws, li, g);             // x1 += W2[:, c] GELU(.)
    }
    add_vec<D / 16>(xr, v_b2, g);                                      //                 -> x2
    tile_to_lds<D / 16, D>(actA, xr, 0, li, g, [](float v) { return v; });
    lds_to_rows<D>(P.xo, actA, tok0, P.ntok, lane);
  } else {
    if (P.z != nullptr) embed_rows<D>(actA, P, tok0, lane);          // first layer: x = embedding, also written to x_out
    else {
      i32x4 xpre[FT * (D * 2 / 16) / 64];
      rows_fetch<D>(xpre, P.x, tok0, P.ntok, lane, P);
      rows_put<D>(actA, xpre, lane);
    }
    lds_to_acc<D / 16, false>(xr, actA, li, g);
  }
  if constexpr (TAIL) {
    {                                                                  // to_q on the raw stream (actA = x2, bf16)
      f32x4 qa[I / 16];
      zero_acc(qa);
      gemm_stage<I, D, D * 2>(qa, actA, ring0, ws, li, g);
      tile_to_lds<I / 16, I>(actA, qa, 0, li, g, [](float v) { return v; });
      lds_to_rows<I>(P.q, actA, tok0, P.ntok, lane);
    }
    ln_to_lds<D / 16>(actA, xr, v_g1n, v_be1n, P.eps, li, g);          // LN1'(x2) -> actA
    f32x4 ka[I / 16], va[I / 16];
    zero_acc(ka);
    zero_acc(va);
    gemm_stage<I, D, D * 2>(ka, actA, ring0, ws, li, g);               // to_k
    gemm_stage<I, D, D * 2>(va, actA, ring0, ws, li, g);               // to_v
    add_vec<I / 16>(ka, v_bkv, g);
    add_vec<I / 16>(va, v_bkv + I, g);
    tile_to_lds<I / 16, 2 * I>(actA, ka, 0, li, g, [](float v) { return v; });
    tile_to_lds<I / 16, 2 * I>(actA, va, I, li, g, [](float v) { return v; });
    lds_to_rows<2 * I>(P.kv, actA, tok0, P.ntok, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the padding slabs still in flight target this workgroup's LDS
}

}  // namespace

static int fused_launch(FusedParams& P, int ntok, int D, int I, int M, int has_head, int has_tail, void* stream);

extern "C" int wmz_layer_fused_fwd(const void* o, const void* x, void* x_out, void* q_out, void* kv_out,
                                   const void* wpack, const float* vec, int ntok, int D, int I, int M, int has_head,
                                   int has_tail, float eps, void* stream) {
  return wmz_layer_fused_fwd_planes(o, x, x_out, q_out, kv_out, wpack, vec, 1, 1, 1, ntok, D, I, M, has_head, has_tail, eps,
                                    stream);
}

extern "C" int wmz_layer_fused_fwd_planes(const void* o, const void* x, void* x_out, void* q_out, void* kv_out,
                                          const void* wpack, const float* vec, int B, int planes_out, int planes_in, int HW,
                                          int D, int I, int M, int has_head, int has_tail, float eps, void* stream) {
  WMZ_REQUIRE(B > 0 && HW > 0 && planes_out > 0 && planes_out <= planes_in, "wmz_layer_fused_fwd: bad plane counts");
  WMZ_REQUIRE((long)B * planes_in * HW < (1L << 31), "wmz_layer_fused_fwd: token count overflows int");
  const int ntok = B * planes_out * HW;
  WMZ_REQUIRE(x && wpack && vec && ntok > 0, "wmz_layer_fused_fwd: bad arguments");
  WMZ_REQUIRE(has_head || has_tail, "wmz_layer_fused_fwd: nothing to do");
  WMZ_REQUIRE(!has_head || (o && x_out), "wmz_layer_fused_fwd: head needs o and x_out");
  WMZ_REQUIRE(!has_tail || (q_out && kv_out), "wmz_layer_fused_fwd: tail needs q_out and kv_out");
  FusedParams P;
  P.o = (const bf16_t*)o; P.x = (const bf16_t*)x; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.ntok = ntok; P.eps = eps;
  P.z = nullptr; P.emb = P.pos_s = P.pos_h = P.pos_w = nullptr; P.S = P.H = P.W = P.num_classes = 0;
  P.rows_out = P.rows_in = P.row0 = 0;
  if (planes_out != planes_in) { P.rows_out = planes_out * HW; P.rows_in = planes_in * HW; P.row0 = (planes_in - planes_out) * HW; }
  return fused_launch(P, ntok, D, I, M, has_head, has_tail, stream);
}

extern "C" int wmz_embed_qkv_fused_fwd(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                       const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                       const float* vec, int B, int S, int H, int W, int D, int I, int M, int num_classes,
                                       float eps, void* stream) {
  return wmz_embed_qkv_fused_fwd_planes(z, emb, pos_s, pos_h, pos_w, x_out, q_out, kv_out, wpack, vec, B, S, H, W, S, D, I, M,
                                        num_classes, eps, stream);
}

extern "C" int wmz_embed_qkv_fused_fwd_planes(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                              const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                              const float* vec, int B, int S, int H, int W, int planes_out, int D, int I,
                                              int M, int num_classes, float eps, void* stream) {
  WMZ_REQUIRE(planes_out > 0 && planes_out <= S, "wmz_embed_qkv_fused_fwd: bad plane count");
  WMZ_REQUIRE((long)B * S * H * W < (1L << 31), "wmz_embed_qkv_fused_fwd: token count overflows int");
  WMZ_REQUIRE(z && emb && pos_s && pos_h && pos_w && x_out && q_out && kv_out && wpack && vec, "wmz_embed_qkv_fused_fwd: null tensor");
  WMZ_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && num_classes > 0, "wmz_embed_qkv_fused_fwd: bad shape");
  FusedParams P;
  P.o = nullptr; P.x = nullptr; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.ntok = B * planes_out * H * W; P.eps = eps;
  P.rows_out = P.rows_in = P.row0 = 0;
  if (planes_out != S) { P.rows_out = planes_out * H * W; P.rows_in = S * H * W; P.row0 = (S - planes_out) * H * W; }
  P.z = z; P.emb = emb; P.pos_s = pos_s; P.pos_h = pos_h; P.pos_w = pos_w; P.S = S; P.H = H; P.W = W; P.num_classes = num_classes;
  return fused_launch(P, P.ntok, D, I, M, 0, 1, stream);
}

static int fused_launch(FusedParams& P, int ntok, int D, int I, int M, int has_head, int has_tail, void* stream) {
  if (!(D == 256 && I == 128 && M == 256)) {
    wmz_set_error("wmz_layer_fused_fwd: built for dim 256 / inner 128 / mlp 256 (got %d/%d/%d); use the unfused path", D, I, M);
    return WMZ_ERR_UNSUPPORTED;
  }
  static const int dbg_env = getenv("WMZ_FUSED_DBG") ? atoi(getenv("WMZ_FUSED_DBG")) : 0;
  P.dbg = dbg_env;
  const size_t smem = FW * (ACTB + ZCB) + VECB + RING * SLAB;
  dim3 grid((unsigned)wmz_cdiv(ntok, FT * FW)), block(NTHR);
  hipStream_t st = (hipStream_t)stream;
#define WMZ_FUSED(H, T)                                                                                            \
  do {                                                                                                             \
    auto kern = layer_fused_kernel<256, 128, 256, H, T>;                                                           \
    static bool attr = false;                                                                                      \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL(kern, grid, block, smem, st, P);                                                            \
  } while (0)
  if (has_head && has_tail) WMZ_FUSED(true, true);
  else if (has_head) WMZ_FUSED(true, false);
  else WMZ_FUSED(false, true);
#undef WMZ_FUSED
  WMZ_LAUNCH_CHECK("wmz_layer_fused_fwd");
  return WMZ_OK;
}
