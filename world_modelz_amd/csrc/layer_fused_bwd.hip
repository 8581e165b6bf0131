// Backward of everything per-token between two attention launches (bf16 speed path, default widths), the counterpart of
// layer_fused.hip on the same machinery (fused_common.h: LDS-DMA weight ring, MFMA 32x32x16 with the token on the lane,
// the accumulator of one GEMM packed into the B operand of the next, lane-local LayerNorm):
//
//   qkv_bwd_kernel  (one layer's attention inputs; local_3d_attention.py:46-48, :106-108, quirk Q1)
//       dx = res + dq Wq + LNbwd(dk Wk' + dv Wv'; x)          ' = LayerNorm gamma folded into the weight
//       xhat = (x - mean) rstd                                 the operand of the to_k | to_v weight gradient
//   ff_bwd_kernel   (one layer's to_out + feed-forward; :11-31, :50-53, :160-161)
//       g = GELU(z), dz = (dy W2) GELU'(z)                     z: the pre-activation the fused forward saved (tiled)
//       dx1 = dy + LNbwd(dz W1'; x1),  do = dx1 Wout,  xhat1 = (x1 - mean) rstd
//
// Replaces, per layer, five GEMM launches, two LayerNorm-backward launches and the pre-activation recompute of the
// op-by-op backward (linear_fwd.hip / linear_bwd.hip), whose [ntok, 256] intermediates each made a round trip through HBM.
// The weight gradients stay separate GEMMs over the token axis (linear_bwd.hip); the kernels write their operands (g, dz,
// xhat, xhat1) so that those run without prologue arithmetic, and the LayerNorm affine gradients follow from the weight
// gradients themselves (wmz_ln_affine_grads below) -- no reduction over tokens happens here.
//
// Layouts.  Row-major [ntok, F] tensors in and out.  A lane (token t = lane & 31, half h = lane >> 5) owns, in every
// 128-feature group of a row, the 64 contiguous features h*64..: one 16-byte load per 8 features going in (issued early,
// not tracked by the compiler: counted waits), the wave's 8 KB LDS image going out (whole 256-byte runs per row).  The
// weight streams are packed for that ownership by wmz_layer_fused_bwd_pack (layer_fused.hip).
#include "fused_common.h"

namespace {

struct BwdParams {
  const char* wpack;
  int ntok;
  // qkv backward
  const bf16_t* dq; long lddq;          // [ntok, I]
  const bf16_t* dkv; long lddkv;        // dk | dv: column halves of [ntok, 2I]
  const bf16_t* x;                      // [ntok, D]  the block's input (row-major copy of the stream)
  const float* st;                      // [2, ntok]  its LayerNorm statistics: means, then reciprocal standard deviations
  const bf16_t* res;                    // [ntok, D]  gradient arriving through the residual connection, or null
  bf16_t* dx;                           // [ntok, D]
  bf16_t* xhat;                         // [ntok, D]  or null
  // feed-forward + to_out backward
  const bf16_t* dy;                     // [ntok, D]  gradient w.r.t. the block's output (LASTDY: only the clips' LAST planes,
                                        //            [ntok / dy_S, D]; every other row's gradient is zero)
  int dy_S, dy_HW;                      // LASTDY: planes per clip, tokens per plane
  const bf16_t* zero_row;               // LASTDY: D zeros
  const bf16_t* zt;                     // tiled pre-activation (layer_fused.hip FusedParams::zt)
  const bf16_t* x1;                     // [ntok, D]  the feed-forward block's input
  const float* st_ff;                   // [2, ntok]
  bf16_t *g, *dz, *xhat1, *dx1, *dout;  // [ntok, M], [ntok, M], [ntok, D], [ntok, D], [ntok, I]
};

__device__ __forceinline__ float gload_f32_untracked(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// the lane's share of a row: k-step s = features (s >> 3) * 128 + h * 64 + (s & 7) * 8 .. +7 (row points at feature h * 64)
// untracked 16-byte load at a compile-time byte offset from ONE base address (a separate 64-bit address per load costs a
// register pair each, alive until the load issues)
template <int OFF>
__device__ __forceinline__ s16x8 gload_untracked_at(const bf16_t* p) {
  static_assert(OFF >= 0 && OFF < 4096, "13-bit signed instruction offset");
  s16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
  return v;
}
// fragments FIRST .. FIRST+N-1 of the lane's share of a row
template <int FIRST, int N, int KS>
__device__ __forceinline__ void load_row_part(Frag8<bf16_t> (&f)[KS], const bf16_t* row, WStream& ws) {
  static_for<N>([&](auto ic) {
    constexpr int s = FIRST + decltype(ic)::value;
    f[s].v = gload_untracked_at<((s >> 3) * 128 + (s & 7) * 8) * 2>(row);
  });
  ws_extra(ws, N);
}
template <int KS>
__device__ __forceinline__ void load_row(Frag8<bf16_t> (&f)[KS], const bf16_t* row, WStream& ws) {
  load_row_part<0, KS, KS>(f, row, ws);
}
template <int KS>
__device__ __forceinline__ void pin(Frag8<bf16_t> (&f)[KS]) {
#pragma unroll
  for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(f[s].v));
}

// eight fragments (one 128-feature group of the lane's token) -> rows of a [ntok, rowf] tensor at column col0
__device__ __forceinline__ void store_group(char* stg, bf16_t* dst, int rowf, long tok0, int ntok, int col0,
                                            const Frag8<bf16_t>* b, int lane, WStream& ws) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 8; ++s) stage_put(stg, b[s].v, t, h * 8 + s);
  if (rowf == 128) stage_flush<128>(stg, dst, tok0, ntok, col0, lane);
  else stage_flush<256>(stg, dst, tok0, ntok, col0, lane);
  ws_extra(ws, 8);
}

// LayerNorm backward without the affine (it lives in the weights), in place on the lane's half of the row:
//   acc <- rstd * (acc - mean(acc) - xhat * mean(acc * xhat)),   xhat = x * rstd - mean * rstd
// x as the bf16 fragments of load_row.  xhat (the weight-gradient operand) leaves through the wave's LDS image, fragment by
// fragment as it is computed.  Register budget: the accumulator tuples (128) + x (64) leave ~60 for everything else and
// NOTHING may spill (see the kernels), so: the second pass recomputes xhat instead of keeping the first pass's values
// (the empty asms make its inputs opaque to common-subexpression elimination; the compiler would otherwise hold -- and
// spill -- 256 values), and the scheduler is fenced block by block.
// XIN: xb already holds xhat (the forward stored the normalised rows): nothing is recomputed and nothing leaves.
template <bool XIN, typename Mid>
__device__ __forceinline__ void ln_bwd_inplace(f32x16 (&acc)[8], Frag8<bf16_t> (&xb)[16], float mean, float rstd, char* stg,
                                               bf16_t* xhat_out, long tok0, int ntok, int lane, WStream& ws, Mid mid) {
  const int t = lane & 31, h = lane >> 5;
  float mr = -mean * rstd;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float xv = bf16_bits_to_f32((unsigned short)xb[2 * b + (i >> 3)].v[i & 7]);
      const float xh = XIN ? xv : fmaf(xv, rstd, mr);
      s1 += acc[b][i];
      s2 = fmaf(acc[b][i], xh, s2);
    }
    WMZ_FENCE();
  }
  s1 = wave_halves_sum(s1) * (1.f / 256.f);
  s2 = wave_halves_sum(s2) * (1.f / 256.f);
  asm volatile("" : "+v"(mr), "+v"(rstd), "+v"(s1), "+v"(s2));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int b = 4 * j + bb;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        asm volatile("" : "+v"(xb[2 * b + m].v));
        s16x8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = bf16_bits_to_f32((unsigned short)xb[2 * b + m].v[e]);
          const float xh = XIN ? xv : fmaf(xv, rstd, mr);
          acc[b][8 * m + e] = rstd * (acc[b][8 * m + e] - s1 - xh * s2);
          if constexpr (!XIN) pk[e] = (short)f32_to_bf16_bits(xh);
        }
        asm volatile("" : "+v"(acc[b]));                    // the update of acc HERE (left alone it sinks to its use in the
        if constexpr (!XIN) stage_put(stg, pk, t, h * 8 + 2 * bb + m);   // next GEMM and every xhat stays alive -- in scratch -- until then)
      }
      WMZ_FENCE();
    }
    if constexpr (!XIN) {
      stage_flush<256>(stg, xhat_out, tok0, ntok, 128 * j, lane);
      ws_extra(ws, 8);
    }
    if (j == 0) mid();                                      // the first half of xb is dead from here on
  }
}

__device__ __forceinline__ void ws_init(WStream& ws, char* smem, const char* wpack, int wave, int lane) {
  ws.ring = smem + wave * (SLAB / FW);
  ws.src = wpack + wave * (SLAB / FW) + lane * 16;
  ws.half = 0;
  ws.issue_slot = 0;
  ws.cur = 0;
  ws.dbg = 0;                 // no ablation switches here: the kernels must stay straight-line code (tools/check_untracked.py)
  ws.tot = ws.t1 = ws.t2 = ws.t3 = 0;
  ws.all = 0;
  ws.wave = wave;
  ws.probe = 0;
  ws.ts = nullptr;
}

template <int D, int I, bool RES, bool XIN>
__global__ __launch_bounds__(NTHR, 8 / FW) void qkv_bwd_kernel(BwdParams P) {
  static_assert(D == 256 && I == 128, "built for the default denoiser widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const char* ring0 = smem;
  char* stg = smem + RING * SLAB + wave * 8192;
  WStream ws;
  ws_init(ws, smem, P.wpack, wave, lane);
  const long tok0 = (long)blockIdx.x * (TW * FW) + wave * TW;
  const long tok = tok0 + (lane & 31);
  const long tokc = tok < P.ntok ? tok : P.ntok - 1;                  // clamped: loads are unconditional, stores predicated
#pragma unroll
  for (int i = 0; i < RING - 1; ++i) ws_issue(ws);

  // Register budget (256 per lane at two waves per SIMD, and NOTHING may spill: scratch traffic would count in vmcnt and
  // break the counted waits): the accumulator is 128, a GEMM's operand fragments in flight 40, so at most 64 more are
  // ever requested ahead -- the next operand while a GEMM runs.
  Frag8<bf16_t> dkb[I / 16], dvb[I / 16];
  load_row<I / 16>(dkb, P.dkv + tokc * P.lddkv + h * 64, ws);
  const int m_k = ws.all;
  load_row<I / 16>(dvb, P.dkv + tokc * P.lddkv + I + h * 64, ws);
  const int m_v = ws.all;
  float mean = gload_f32_untracked(P.st + tokc);
  float rstd = gload_f32_untracked(P.st + (long)P.ntok + tokc);
  ws_extra(ws, 2);

  f32x16 acc[D / 32];
  zero_acc(acc);
  vm_wait_since(ws, m_k);
  pin(dkb);
  gemm_stage<D / 32, I / 16>(acc, dkb, ring0, ws, lane);               // dk Wk'
  Frag8<bf16_t> xb[D / 16];
  const bf16_t* xrow = P.x + tokc * D + h * 64;
  load_row_part<0, 8, D / 16>(xb, xrow, ws);                           // first half of x: in flight under the second GEMM
  vm_wait_since(ws, m_v);
  pin(dvb);
  gemm_stage<D / 32, I / 16>(acc, dvb, ring0, ws, lane);               // + dv Wv'   = d LN(x)
  load_row_part<8, 8, D / 16>(xb, xrow, ws);
  const int m_x = ws.all;
  vm_wait_since(ws, m_x);
  pin(xb);
  asm volatile("" : "+v"(mean), "+v"(rstd));                           // (older than xb: landed)
  Frag8<bf16_t> dqb[I / 16];
  int m_q = 0;
  ln_bwd_inplace<XIN>(acc, xb, mean, rstd, stg, P.xhat, tok0, P.ntok, lane, ws, [&]() {
    load_row<I / 16>(dqb, P.dq + tokc * P.lddq + h * 64, ws);          // in flight under the second half of the LayerNorm pass
    m_q = ws.all;
  });
  vm_wait_since(ws, m_q);
  pin(dqb);
  const bf16_t* rrow = RES ? P.res + tokc * D + h * 64 : nullptr;
  if constexpr (RES) load_row_part<0, 8, D / 16>(xb, rrow, ws);
  gemm_stage<D / 32, I / 16>(acc, dqb, ring0, ws, lane);               // + dq Wq
  if constexpr (RES) {
    load_row_part<8, 8, D / 16>(xb, rrow, ws);
    const int m_r = ws.all;
    vm_wait_since(ws, m_r);
    pin(xb);
    add_bop<D / 32>(acc, xb);
  }
  bop_from_acc<D / 32>(xb, acc);
  store_group(stg, P.dx, D, tok0, P.ntok, 0, &xb[0], lane, ws);
  store_group(stg, P.dx, D, tok0, P.ntok, 128, &xb[8], lane, ws);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the padding slabs still in flight target this workgroup's LDS
}

template <int D, int I, int M, bool LASTDY, bool XIN>
__global__ __launch_bounds__(NTHR, 8 / FW) void ff_bwd_kernel(BwdParams P) {
  static_assert(D == 256 && I == 128 && M == 256, "built for the default denoiser widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, t = lane & 31;
  const char* ring0 = smem;
  char* stg = smem + RING * SLAB + wave * 8192;
  WStream ws;
  ws_init(ws, smem, P.wpack, wave, lane);
  const long tok0 = (long)blockIdx.x * (TW * FW) + wave * TW;
  const long tok = tok0 + t;
  const long tokc = tok < P.ntok ? tok : P.ntok - 1;
  const long tok0c = tok0 < P.ntok ? tok0 : (P.ntok >= 32 ? P.ntok - 32 : 0);   // z tiles: whole tiles only (ntok % 32 == 0)
#pragma unroll
  for (int i = 0; i < RING - 1; ++i) ws_issue(ws);

  // LASTDY: the loss reads the last plane only (main.py:37), so the last layer's dy is zero everywhere else -- those rows
  // are read from one zero row instead of from a [ntok, D] tensor of zeros somebody had to write
  const bf16_t* dyrow;
  if constexpr (LASTDY) {
    const int per = P.dy_S * P.dy_HW, tt = (int)tokc;
    const int b = tt / per, rr = tt - b * per - (per - P.dy_HW);
    dyrow = rr >= 0 ? P.dy + ((long)b * P.dy_HW + rr) * D + h * 64 : P.zero_row + h * 64;
  } else {
    dyrow = P.dy + tokc * D + h * 64;
  }
  Frag8<bf16_t> dyb[D / 16];
  load_row<D / 16>(dyb, dyrow, ws);
  const int m_dy = ws.all;
  const bf16_t* ztile = P.zt + tok0c * M + lane * 8;
  Frag8<bf16_t> zf[2][2];
  int m_z[2];
  auto fetch_z = [&](int c) {
    zf[c & 1][0].v = gload_untracked(ztile + (2 * c) * 512);
    zf[c & 1][1].v = gload_untracked(ztile + (2 * c + 1) * 512);
    ws_extra(ws, 2);
    m_z[c & 1] = ws.all;
  };
  fetch_z(0);
  vm_wait_since(ws, m_dy);
  pin(dyb);

  // ---- pass 1, 32 hidden units at a time: dg_c = W2[:, c]^T dy, then g_c = GELU(z_c), dz_c = dg_c GELU'(z_c)
  Frag8<bf16_t> dzb[M / 16];
#pragma unroll
  for (int c = 0; c < M / MC; ++c) {
    f32x16 dg[1];
    zero_acc(dg);
    gemm_stage<1, D / 16>(dg, dyb, ring0, ws, lane);
    if (c + 1 < M / MC) fetch_z(c + 1);
    vm_wait_since(ws, m_z[c & 1]);
    pin(zf[c & 1]);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float gv[8], dv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float gg, dd;
        wmz_gelu_fast_both(bf16_bits_to_f32((unsigned short)zf[c & 1][m].v[e]), gg, dd);
        gv[e] = gg;
        dv[e] = dg[0][8 * m + e] * dd;
      }
      Frag8<bf16_t> gf;
      pack8(gf, gv);
      pack8(dzb[2 * c + m], dv);
      stage_put(stg, gf.v, t, (c & 3) * 4 + 2 * h + m);               // hidden 32c + 16h + 8m .. of this 128-wide group
    }
    if ((c & 3) == 3) { stage_flush<256>(stg, P.g, tok0, P.ntok, (c >> 2) * 128, lane); ws_extra(ws, 8); }
  }
  // dz rows: the same ownership (hidden 32c + 16h + 8m = fragment 2c + m)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int m = 0; m < 2; ++m) stage_put(stg, dzb[2 * (4 * j + cc) + m].v, t, cc * 4 + 2 * h + m);
    stage_flush<256>(stg, P.dz, tok0, P.ntok, 128 * j, lane);
    ws_extra(ws, 8);
  }

  // ---- pass 2: d LN2(x1) = W1'^T dz, LayerNorm backward, + dy
  f32x16 acc[D / 32];
  zero_acc(acc);
  Frag8<bf16_t> xb[D / 16];
  gemm_stage<D / 32, M / 16>(acc, dzb, ring0, ws, lane);
  load_row<D / 16>(xb, P.x1 + tokc * D + h * 64, ws);
  float mean = gload_f32_untracked(P.st_ff + tokc);
  float rstd = gload_f32_untracked(P.st_ff + (long)P.ntok + tokc);
  ws_extra(ws, 2);
  const int m_x = ws.all;
  vm_wait_since(ws, m_x);
  pin(xb);
  asm volatile("" : "+v"(mean), "+v"(rstd));
  ln_bwd_inplace<XIN>(acc, xb, mean, rstd, stg, P.xhat1, tok0, P.ntok, lane, ws, []() {});
  load_row<D / 16>(xb, dyrow, ws);                                     // the residual path (L2-hot: this wave read it above)
  const int m_r = ws.all;
  vm_wait_since(ws, m_r);
  pin(xb);
  add_bop<D / 32>(acc, xb);                                            // dx1
  bop_from_acc<D / 32>(xb, acc);
  store_group(stg, P.dx1, D, tok0, P.ntok, 0, &xb[0], lane, ws);
  store_group(stg, P.dx1, D, tok0, P.ntok, 128, &xb[8], lane, ws);

  // ---- do = dx1 Wout
  f32x16 oa[I / 32];
  zero_acc(oa);
  gemm_stage<I / 32, D / 16>(oa, xb, ring0, ws, lane);
  Frag8<bf16_t> ob[I / 16];
  bop_from_acc<I / 32>(ob, oa);
  store_group(stg, P.dout, I, tok0, P.ntok, 0, ob, lane, ws);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// dW[N, K] += G[n, k] gamma[k] + s[n] beta[k];  dgamma[k] += sum_n W[n, k] G[n, k];  dbeta[k] += sum_n W[n, k] s[n];
// dbias[n] += s[n] for n >= bias_from (to_k has no bias: the k | v gradient shares one call).  A workgroup owns 64 columns
// x 8 rows: thread (row group rg = tid / 64, column) walks 2 rows, the four row groups meet in LDS, one atomic per column
// (128 workgroups for a 256 x 256 weight: the kernel is pure latency, short chains and many workgroups keep it at ~3 us).
struct LnAffineProb { const float *G, *s, *W, *gamma, *beta; float *dW, *dbias, *dgamma, *dbeta; int N, K, bias_from; };
struct LnAffineBatch { LnAffineProb p[4]; };      // blockIdx.z selects the problem (a layer has two: feed-forward and k | v)
__global__ __launch_bounds__(256) void ln_affine_grads_kernel(LnAffineBatch Bt) {
  const LnAffineProb& Q = Bt.p[blockIdx.z];
  const float* __restrict__ G = Q.G;
  const float* __restrict__ s = Q.s;
  const float* __restrict__ W = Q.W;
  const float* __restrict__ gamma = Q.gamma;
  const float* __restrict__ beta = Q.beta;
  float* __restrict__ dW = Q.dW;
  float* __restrict__ dbias = Q.dbias;
  float* __restrict__ dgamma = Q.dgamma;
  float* __restrict__ dbeta = Q.dbeta;
  const int N = Q.N, K = Q.K, bias_from = Q.bias_from;
  if ((int)blockIdx.x * 64 >= K || (int)blockIdx.y * 8 >= N) return;       // (the grid covers the largest problem of the batch)
  __shared__ float red[2][4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  const int n0 = blockIdx.y * 8 + rg * 2;
  float ag = 0.f, ab = 0.f;
  if (k < K) {
    const float gk = gamma[k], bk = beta[k];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + i;
      if (n < N) {
        const float g = G[(long)n * K + k], w = W[(long)n * K + k], sn = s[n];
        dW[(long)n * K + k] += fmaf(g, gk, sn * bk);
        ag = fmaf(w, g, ag);
        ab = fmaf(w, sn, ab);
      }
    }
  }
  red[0][rg][c] = ag;
  red[1][rg][c] = ab;
  __syncthreads();
  if (rg == 0 && k < K) {
    atomicAdd(dgamma + k, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
    atomicAdd(dbeta + k, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
  }
  if (dbias != nullptr && blockIdx.x == 0 && threadIdx.x < 8) {
    const int n = blockIdx.y * 8 + (int)threadIdx.x;
    if (n < N && n >= bias_from) dbias[n - bias_from] += s[n];
  }
}

}  // namespace

extern "C" int wmz_qkv_fused_bwd(const void* dq, long lddq, const void* dkv, long lddkv, const void* x, const float* ln_stats,
                                 const void* res, void* dx, void* xhat_out, const void* wpack, int ntok, int D, int I,
                                 void* stream) {
  WMZ_REQUIRE(dq && dkv && x && ln_stats && dx && wpack && ntok > 0, "wmz_qkv_fused_bwd: null tensor");
  if (!(D == 256 && I == 128)) {
    wmz_set_error("wmz_qkv_fused_bwd: built for dim 256 / inner 128 (got %d/%d); use the per-op backward", D, I);
    return WMZ_ERR_UNSUPPORTED;
  }
  WMZ_REQUIRE(lddq >= I && lddkv >= 2 * I && lddq % 8 == 0 && lddkv % 8 == 0, "wmz_qkv_fused_bwd: bad row strides");
  // whole 32-token wave tiles: a wave's row stores are then all issued or all skipped, which the counted waits rely on
  WMZ_REQUIRE(ntok % 32 == 0, "wmz_qkv_fused_bwd: ntok must be a multiple of 32");
  BwdParams P = {};
  P.wpack = (const char*)wpack; P.ntok = ntok;
  P.dq = (const bf16_t*)dq; P.lddq = lddq; P.dkv = (const bf16_t*)dkv; P.lddkv = lddkv;
  P.x = (const bf16_t*)x; P.st = ln_stats; P.res = (const bf16_t*)res; P.dx = (bf16_t*)dx; P.xhat = (bf16_t*)xhat_out;
  const size_t smem = RING * SLAB + FW * 8192;
  const dim3 grid((unsigned)wmz_cdiv(ntok, TW * FW)), block(NTHR);
#define WMZ_QKV_BWD(RES_, XIN_)                                                                                            \
  do {                                                                                                                     \
    auto kern = qkv_bwd_kernel<256, 128, RES_, XIN_>;                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)stream, P);                                                   \
  } while (0)
  const bool xin = xhat_out == nullptr;       // x already holds the normalised rows (WMZ_FUSED_XRM_NORMALISED forward)
  if (res != nullptr) { if (xin) WMZ_QKV_BWD(true, true); else WMZ_QKV_BWD(true, false); }
  else { if (xin) WMZ_QKV_BWD(false, true); else WMZ_QKV_BWD(false, false); }
#undef WMZ_QKV_BWD
  WMZ_LAUNCH_CHECK("wmz_qkv_fused_bwd");
  return WMZ_OK;
}

extern "C" int wmz_ff_fused_bwd(const void* dy, const void* z_tiled, const void* x1, const float* ln_stats, void* g_out,
                                void* dz_out, void* xhat_out, void* dx1_out, void* do_out, const void* wpack, int ntok, int D,
                                int I, int M, int dy_last_planes, int dy_plane_tokens, const void* zero_row, void* stream) {
  WMZ_REQUIRE(dy && z_tiled && x1 && ln_stats && g_out && dz_out && dx1_out && do_out && wpack && ntok > 0,
              "wmz_ff_fused_bwd: null tensor");
  if (!(D == 256 && I == 128 && M == 256)) {
    wmz_set_error("wmz_ff_fused_bwd: built for dim 256 / inner 128 / mlp 256 (got %d/%d/%d); use the per-op backward", D, I, M);
    return WMZ_ERR_UNSUPPORTED;
  }
  WMZ_REQUIRE(ntok % 32 == 0, "wmz_ff_fused_bwd: the tiled pre-activation needs whole 32-token tiles");
  const bool lastdy = dy_last_planes > 0;
  WMZ_REQUIRE(!lastdy || (dy_plane_tokens > 0 && zero_row && ntok % (dy_last_planes * dy_plane_tokens) == 0),
              "wmz_ff_fused_bwd: last-plane dy needs planes per clip, tokens per plane (ntok a multiple of their product) and a zero row");
  BwdParams P = {};
  P.wpack = (const char*)wpack; P.ntok = ntok;
  P.dy = (const bf16_t*)dy; P.zt = (const bf16_t*)z_tiled; P.x1 = (const bf16_t*)x1; P.st_ff = ln_stats;
  P.dy_S = dy_last_planes; P.dy_HW = dy_plane_tokens; P.zero_row = (const bf16_t*)zero_row;
  P.g = (bf16_t*)g_out; P.dz = (bf16_t*)dz_out; P.xhat1 = (bf16_t*)xhat_out; P.dx1 = (bf16_t*)dx1_out; P.dout = (bf16_t*)do_out;
  const size_t smem = RING * SLAB + FW * 8192;
  const dim3 grid((unsigned)wmz_cdiv(ntok, TW * FW)), block(NTHR);
#define WMZ_FF_BWD(LAST_, XIN_)                                                                                            \
  do {                                                                                                                     \
    auto kern = ff_bwd_kernel<256, 128, 256, LAST_, XIN_>;                                                                 \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)stream, P);                                                   \
  } while (0)
  const bool xin = xhat_out == nullptr;       // x1 already holds the normalised rows (WMZ_FUSED_X1_NORMALISED forward)
  if (lastdy) { if (xin) WMZ_FF_BWD(true, true); else WMZ_FF_BWD(true, false); }
  else { if (xin) WMZ_FF_BWD(false, true); else WMZ_FF_BWD(false, false); }
#undef WMZ_FF_BWD
  WMZ_LAUNCH_CHECK("wmz_ff_fused_bwd");
  return WMZ_OK;
}

extern "C" int wmz_ln_affine_grads(const float* G, const float* s, const float* W, const float* gamma, const float* beta,
                                   float* dW, float* dbias, float* dgamma, float* dbeta, int N, int K, int bias_from,
                                   void* stream) {
  WMZ_REQUIRE(G && s && W && gamma && beta && dW && dgamma && dbeta && N > 0 && K > 0, "wmz_ln_affine_grads: bad arguments");
  WMZ_REQUIRE(bias_from >= 0 && bias_from <= N, "wmz_ln_affine_grads: bad bias_from");
  LnAffineBatch Bt = {};
  Bt.p[0] = LnAffineProb{G, s, W, gamma, beta, dW, dbias, dgamma, dbeta, N, K, bias_from};
  hipLaunchKernelGGL(ln_affine_grads_kernel, dim3((unsigned)wmz_cdiv(K, 64), (unsigned)wmz_cdiv(N, 8), 1), dim3(256), 0,
                     (hipStream_t)stream, Bt);
  WMZ_LAUNCH_CHECK("wmz_ln_affine_grads");
  return WMZ_OK;
}

extern "C" int wmz_ln_affine_grads_batch(int n, const float* const* G, const float* const* s, const float* const* W,
                                         const float* const* gamma, const float* const* beta, float* const* dW,
                                         float* const* dbias, float* const* dgamma, float* const* dbeta, const int* N,
                                         const int* K, const int* bias_from, void* stream) {
  WMZ_REQUIRE(n >= 1 && n <= 4, "wmz_ln_affine_grads_batch: 1 .. 4 problems per call (got %d)", n);
  WMZ_REQUIRE(G && s && W && gamma && beta && dW && dbias && dgamma && dbeta && N && K && bias_from, "wmz_ln_affine_grads_batch: null table");
  LnAffineBatch Bt = {};
  int gx = 0, gy = 0;
  for (int i = 0; i < n; ++i) {
    WMZ_REQUIRE(G[i] && s[i] && W[i] && gamma[i] && beta[i] && dW[i] && dgamma[i] && dbeta[i] && N[i] > 0 && K[i] > 0 &&
                bias_from[i] >= 0 && bias_from[i] <= N[i], "wmz_ln_affine_grads_batch: bad problem %d", i);
    Bt.p[i] = LnAffineProb{G[i], s[i], W[i], gamma[i], beta[i], dW[i], dbias[i], dgamma[i], dbeta[i], N[i], K[i], bias_from[i]};
    gx = gx > wmz_cdiv(K[i], 64) ? gx : wmz_cdiv(K[i], 64);
    gy = gy > wmz_cdiv(N[i], 8) ? gy : wmz_cdiv(N[i], 8);
  }
  hipLaunchKernelGGL(ln_affine_grads_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)n), dim3(256), 0, (hipStream_t)stream, Bt);
  WMZ_LAUNCH_CHECK("wmz_ln_affine_grads_batch");
  return WMZ_OK;
}
