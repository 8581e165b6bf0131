// One launch per transformer layer boundary for everything that is per-token (bf16 speed path):
//
//   HEAD:  x1 = o Wout^T + bout + x                       (to_out + residual,        local_3d_attention.py:50-53, :160)
//          x2 = W2 GELU(W1 LN2(x1) + b1) + b2 + x1        (PreNorm(FeedForward) + x, :11-31, :161)
//   TAIL:  q' = Wq' x2 ,  k'|v' = Wkv' LN1'(x2) + bkv'    (NEXT layer's to_q on the raw stream and to_k|to_v on
//                                                          LayerNorm(x): quirk Q1, :16-17, :46-48, :106-108)
//
// Design (MI355X): a workgroup = 4 waves = 128 tokens; each wave owns 32 tokens end to end, so the only thing the waves
// share is the weight stream.  Every GEMM is computed TRANSPOSED, D^T[features x 32 tokens] = W[features x K] . act^T, with
// MFMA 32x32x16 bf16: A = weight rows, B = the wave's activation rows -- both K-contiguous row fragments, no transposes --
// and the token on the lane.  Consequences: the residual stream x1/x2 stays in fp32 REGISTERS across the whole chain
// (128 accumulator VGPRs, one wave per SIMD, 512-register budget); LayerNorm statistics are lane-local sums plus one
// cross-half shuffle; each lane holds 4 consecutive features per register quad, so activations are exchanged through two
// per-wave 16 KB LDS buffers with 8-byte accesses (no workgroup barrier), and every HBM store is a whole 512-byte row.
// The layer's 512 KB of weights are pre-packed on the host in exactly the order the kernel consumes them (16 KB slabs of
// [k-step][feature][16 k], half-swizzled against bank conflicts) and streamed global -> registers -> 2-slab LDS ring, one
// barrier per slab (= 16 MFMAs per wave).
#include "wmz_common.h"

namespace {

constexpr int FT = 32;          // tokens per wave
constexpr int FW = 4;           // waves per workgroup
constexpr int NTHR = FW * 64;
constexpr int SLAB = 16384;     // bytes per weight slab
constexpr int ACTB = 16384;     // bytes per per-wave activation buffer (32 tokens x 256 features bf16)

struct FusedParams {
  const bf16_t* o;      // [ntok, I]   attention output               (HEAD)
  const bf16_t* x;      // [ntok, D]   residual stream in
  bf16_t* xo;           // [ntok, D]   residual stream out            (HEAD)
  bf16_t* q;            // [ntok, I]                                   (TAIL)
  bf16_t* kv;           // [ntok, 2I]                                  (TAIL)
  const char* wpack;    // packed bf16 weights in streaming order (+ 2 slabs of padding)
  const float* vec;     // packed fp32 vectors: bout[D] g2[D] be2[D] b1[M] b2[D] g1n[D] be1n[D] bkv[2I]
  int ntok;
  float eps;
};

__device__ __forceinline__ int aswz(int token) { return (token & 15) << 4; }
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// weight stream: ring of two 16 KB slabs; `pre` holds the slab after the one being multiplied
struct WStream {
  const char* src;     // next global slab to fetch
  char* ring;          // LDS, 2 * SLAB
  int cur;             // ring index holding the current slab
};

__device__ __forceinline__ void ws_fetch(i32x4 (&pre)[4], const char* src, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) pre[i] = *reinterpret_cast<const i32x4*>(src + (tid + NTHR * i) * 16);
}
__device__ __forceinline__ void ws_put(char* dst, const i32x4 (&pre)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<i32x4*>(dst + (tid + NTHR * i) * 16) = pre[i];
}

// acc^T[N x 32 tokens] += W[N x K] . act^T, consuming K/16 k-steps of the stream.  act: this wave's LDS buffer,
// rows = tokens, ROWB = K*2 bytes, 16-byte chunks XOR-swizzled by aswz(token).
template <int N, int K>
__device__ __forceinline__ void gemm_stage(f32x16 (&acc)[N / 32], const char* act, WStream& ws, i32x4 (&pre)[4], int tid,
                                           int l31, int hh) {
  constexpr int NB = N / 32;
  constexpr int KPS = SLAB / (N * 32);          // k-steps per slab
  constexpr int NSLAB = (K / 16) / KPS;
  constexpr int ROWB = K * 2;
  const char* arow = act + l31 * ROWB;
  const int asw = aswz(l31);
  const int whalf = (hh ^ ((l31 >> 3) & 1)) * 16; // physical half of this lane's weight row (pre-swizzled on the host)
#pragma unroll 1
  for (int s = 0; s < NSLAB; ++s) {
    const char* slab = ws.ring + ws.cur * SLAB;
#pragma unroll
    for (int t = 0; t < KPS; ++t) {
      const int ks = s * KPS + t;
      Frag8<bf16_t> bf;
      bf.v = *reinterpret_cast<const s16x8*>(arow + (((ks * 2 + hh) << 4) ^ asw));
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        Frag8<bf16_t> af;
        af.v = *reinterpret_cast<const s16x8*>(slab + t * (N * 32) + (32 * b + l31) * 32 + whalf);
        mma32(acc[b], af, bf);
      }
    }
    // hand-over: the prefetched slab goes into the other ring slot (everyone left it one barrier ago), the fetch after
    // next is issued, and one barrier publishes the new slab and retires the current one
    ws_put(ws.ring + (ws.cur ^ 1) * SLAB, pre, tid);
    ws.src += SLAB;
    ws_fetch(pre, ws.src, tid);
    __syncthreads();
    ws.cur ^= 1;
  }
}

// feature quad of accumulator block b, register group g4 (regs 4g4..4g4+3): n0 .. n0+3
__device__ __forceinline__ int quad_n0(int b, int g4, int hh) { return 32 * b + 8 * g4 + 4 * hh; }

template <int NB>
__device__ __forceinline__ void add_vec(f32x16 (&acc)[NB], const float* vec, int hh) {   // vec already offset to feature 0 of acc
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(vec + quad_n0(b, g4, hh));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[b][4 * g4 + r] += v[r];
      if (g4 == 3) __builtin_amdgcn_sched_barrier(0);   // keep the loads of later blocks from piling up in VGPRs
    }
}

// write the wave's [32 tokens x NB*32 features] tile (bf16) into columns n_off.. of its LDS buffer (rows of ROWF
// features, row-major, swizzled)
template <int NB, int ROWF, typename F>
__device__ __forceinline__ void tile_to_lds(char* act, const f32x16 (&acc)[NB], int n_off, int l31, int hh, F f) {
  constexpr int ROWB = ROWF * 2;
  char* row = act + l31 * ROWB;
  const int sw = aswz(l31);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n0 = n_off + quad_n0(b, g4, hh);
      s16x4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(f(acc[b][4 * g4 + r], n0 + r));
      *reinterpret_cast<s16x4*>(row + ((n0 * 2) ^ sw)) = pk;
      __builtin_amdgcn_sched_barrier(0);
    }
}

// LayerNorm of the token held by this lane pair (features split over hh): returns mean, rstd
template <int NB>
__device__ __forceinline__ void ln_stats(const f32x16 (&acc)[NB], float eps, float& mean, float& rstd) {
  constexpr int NF = NB * 32;
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[b][r];
  s += __shfl_xor(s, 32);
  mean = s / (float)NF;
  float q = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = acc[b][r] - mean; q += d * d; }
  q += __shfl_xor(q, 32);
  rstd = rsqrtf(q / (float)NF + eps);
}

// LN(acc) -> bf16 -> LDS buffer
template <int NB>
__device__ __forceinline__ void ln_to_lds(char* act, const f32x16 (&acc)[NB], const float* gamma, const float* beta,
                                          float eps, int l31, int hh) {
  float mean, rstd;
  ln_stats<NB>(acc, eps, mean, rstd);
  constexpr int ROWB = NB * 32 * 2;
  char* row = act + l31 * ROWB;
  const int sw = aswz(l31);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n0 = quad_n0(b, g4, hh);
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + n0);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + n0);
      s16x4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits((acc[b][4 * g4 + r] - mean) * rstd * g[r] + be[r]);
      *reinterpret_cast<s16x4*>(row + ((n0 * 2) ^ sw)) = pk;
      if (g4 == 3) __builtin_amdgcn_sched_barrier(0);
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
    *reinterpret_cast<i32x4*>(act + r * ROWB + ((c << 4) ^ aswz(r))) = v;
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
    const i32x4 v = *reinterpret_cast<const i32x4*>(act + r * ROWB + ((c << 4) ^ aswz(r)));
    if (tok0 + r < ntok) *reinterpret_cast<i32x4*>(dst + (tok0 + r) * F + c * 8) = v;
  }
}
// read the lane's token row back from LDS into the accumulator layout (fp32)
template <int NB>
__device__ __forceinline__ void lds_to_acc(f32x16 (&acc)[NB], const char* act, int l31, int hh, bool add) {
  constexpr int ROWB = NB * 32 * 2;
  const char* row = act + l31 * ROWB;
  const int sw = aswz(l31);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const s16x4 pk = *reinterpret_cast<const s16x4*>(row + ((quad_n0(b, g4, hh) * 2) ^ sw));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = bf16_bits_to_f32((unsigned short)pk[r]);
        acc[b][4 * g4 + r] = add ? acc[b][4 * g4 + r] + v : v;
      }
      if (g4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int D, int I, int M, bool HEAD, bool TAIL>
__global__ __launch_bounds__(NTHR, 1) void layer_fused_kernel(FusedParams P) {
  static_assert(D == 256 && M == 256 && I == 128, "built for the default denoiser widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  char* actA = smem + wave * ACTB;
  char* actB = smem + FW * ACTB + wave * ACTB;
  WStream ws;
  ws.ring = smem + 2 * FW * ACTB;
  ws.src = P.wpack;
  ws.cur = 0;
  const long tok0 = (long)blockIdx.x * (FT * FW) + wave * FT;
  const float* v_bout = P.vec;
  const float* v_g2 = v_bout + D;
  const float* v_be2 = v_g2 + D;
  const float* v_b1 = v_be2 + D;
  const float* v_b2 = v_b1 + M;
  const float* v_g1n = v_b2 + D;
  const float* v_be1n = v_g1n + D;
  const float* v_bkv = v_be1n + D;

  // prime the weight ring: slab 0 -> ring[0], slab 1 in flight
  i32x4 pre[4];
  ws_fetch(pre, ws.src, tid);
  ws_put(ws.ring, pre, tid);
  ws.src += SLAB;
  ws_fetch(pre, ws.src, tid);

  f32x16 xr[D / 32];                         // the residual stream of this lane's token, fp32, lives here
  if constexpr (HEAD) {
    rows_to_lds<I>(actA, P.o, tok0, P.ntok, lane);
    rows_to_lds<D>(actB, P.x, tok0, P.ntok, lane);
    __syncthreads();                         // ring[0] visible (the act buffers are wave-private)
#pragma unroll
    for (int b = 0; b < D / 32; ++b) xr[b] = (f32x16)(0.f);
    gemm_stage<D, I>(xr, actA, ws, pre, tid, l31, hh);            // o Wout^T
    add_vec<D / 32>(xr, v_bout, hh);
    lds_to_acc<D / 32>(xr, actB, l31, hh, true);                   // + x            -> x1
    ln_to_lds<D / 32>(actA, xr, v_g2, v_be2, P.eps, l31, hh);      // LN2(x1) -> actA
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {                          // W1 LN2(x1) in two 128-feature halves
      f32x16 z[M / 64];
#pragma unroll
      for (int b = 0; b < M / 64; ++b) z[b] = (f32x16)(0.f);
      gemm_stage<M / 2, D>(z, actA, ws, pre, tid, l31, hh);
      add_vec<M / 64>(z, v_b1 + half * (M / 2), hh);
      tile_to_lds<M / 64, M>(actB, z, half * (M / 2), l31, hh, [](float v, int) { return gelu_erf(v); });
    }
    gemm_stage<D, M>(xr, actB, ws, pre, tid, l31, hh);             // x1 += W2 GELU(.)
    add_vec<D / 32>(xr, v_b2, hh);                                 //                 -> x2
    tile_to_lds<D / 32, D>(actA, xr, 0, l31, hh, [](float v, int) { return v; });
    lds_to_rows<D>(P.xo, actA, tok0, P.ntok, lane);
  } else {
    rows_to_lds<D>(actA, P.x, tok0, P.ntok, lane);
    __syncthreads();
    lds_to_acc<D / 32>(xr, actA, l31, hh, false);
  }
  if constexpr (TAIL) {
    // actA holds x2 (bf16): to_q on the raw stream; LN1'(x2) -> actB for to_k | to_v
    ln_to_lds<D / 32>(actB, xr, v_g1n, v_be1n, P.eps, l31, hh);
    {
      f32x16 qa[I / 32];
#pragma unroll
      for (int b = 0; b < I / 32; ++b) qa[b] = (f32x16)(0.f);
      gemm_stage<I, D>(qa, actA, ws, pre, tid, l31, hh);
      tile_to_lds<I / 32, I>(actA, qa, 0, l31, hh, [](float v, int) { return v; });
      lds_to_rows<I>(P.q, actA, tok0, P.ntok, lane);
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {                          // to_k, then to_v, into the two column halves
      f32x16 ka[I / 32];
#pragma unroll
      for (int b = 0; b < I / 32; ++b) ka[b] = (f32x16)(0.f);
      gemm_stage<I, D>(ka, actB, ws, pre, tid, l31, hh);
      add_vec<I / 32>(ka, v_bkv + half * I, hh);
      tile_to_lds<I / 32, 2 * I>(actA, ka, half * I, l31, hh, [](float v, int) { return v; });
    }
    lds_to_rows<2 * I>(P.kv, actA, tok0, P.ntok, lane);
  }
}

}  // namespace

extern "C" int wmz_layer_fused_fwd(const void* o, const void* x, void* x_out, void* q_out, void* kv_out,
                                   const void* wpack, const float* vec, int ntok, int D, int I, int M, int has_head,
                                   int has_tail, float eps, void* stream) {
  WMZ_REQUIRE(x && wpack && vec && ntok > 0, "wmz_layer_fused_fwd: bad arguments");
  WMZ_REQUIRE(has_head || has_tail, "wmz_layer_fused_fwd: nothing to do");
  WMZ_REQUIRE(!has_head || (o && x_out), "wmz_layer_fused_fwd: head needs o and x_out");
  WMZ_REQUIRE(!has_tail || (q_out && kv_out), "wmz_layer_fused_fwd: tail needs q_out and kv_out");
  if (!(D == 256 && I == 128 && M == 256)) {
    wmz_set_error("wmz_layer_fused_fwd: built for dim 256 / inner 128 / mlp 256 (got %d/%d/%d); use the unfused path", D, I, M);
    return WMZ_ERR_UNSUPPORTED;
  }
  FusedParams P;
  P.o = (const bf16_t*)o; P.x = (const bf16_t*)x; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.ntok = ntok; P.eps = eps;
  const size_t smem = 2 * FW * ACTB + 2 * SLAB;
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
