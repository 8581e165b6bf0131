// One launch per transformer layer boundary for everything that is per-token (bf16 speed path):
//
//   HEAD:  x1 = o Wout^T + bout + x                       (to_out + residual,        local_3d_attention.py:50-53, :160)
//          x2 = W2 GELU(W1 LN2(x1) + b1) + b2 + x1        (PreNorm(FeedForward) + x, :11-31, :161)
//   TAIL:  q' = Wq' x2 ,  k'|v' = Wkv' LN1'(x2) + bkv'    (NEXT layer's to_q on the raw stream and to_k|to_v on
//                                                          LayerNorm(x): quirk Q1, :16-17, :46-48, :106-108)
//
// Design (MI355X): a workgroup = 8 waves = 256 tokens, one workgroup per CU; each wave owns 32 tokens end to end, so the
// only thing the waves share is the weight stream.  Every GEMM is computed TRANSPOSED with MFMA 32x32x16 bf16,
// D[32 features x 32 tokens] += W[32 features x 16 k] . act^T[16 k x 32 tokens]: the token sits on the lane, and
//   * the accumulator of one GEMM IS the B operand of the next (after bf16 packing): the order in which a GEMM walks its
//     k axis and the order of its output features are free, so the host packs the weights such that lane (token t,
//     half h) owns the CONTIGUOUS features h*N/2 .. h*N/2+N/2-1 of every activation, in accumulator registers and in
//     operand registers alike.  Activations never touch LDS; the residual stream stays in fp32 registers across the
//     whole chain; LayerNorm is lane-local sums plus one cross-half exchange; global loads / stores are 16-byte pieces
//     of one token row per lane.
//   * the layer's 512 KB of weights are pre-packed in exactly the order the kernel consumes them -- 1 KB pieces, one
//     MFMA A operand each (lane l's 16 bytes at l*16: conflict-free ds_read_b128 without a swizzle) -- and streamed by
//     LDS-DMA into a 4-slot ring of 16 KB slabs, 3 slabs in flight: counted vmcnt + ONE raw s_barrier per slab
//     (= 16 MFMAs of 32 cycles per wave).  Each A fragment read from LDS feeds 32 tokens: the LDS array runs at half the
//     rate that would bound the MFMAs (16-token waves with 16x16x32 sit exactly on that bound).
//   * the feed-forward is walked 32 hidden units at a time (W1 rows -> GELU -> W2 columns), so its activation never
//     exists in full and the chunk's accumulator (16 registers) is converted in place to the next operand.
#include "fused_common.h"
#include "wmz_debug.h"

namespace {

struct FusedParams {
  const bf16_t* o;      // [ntok, I]   attention output               (HEAD)
  const bf16_t* x;      // [ntok, D]   residual stream in
  bf16_t* xo;           // [ntok, D]   residual stream out            (HEAD)
  bf16_t* q;            // [ntok, I]                                   (TAIL)
  bf16_t* kv;           // [2, ntok, I]  k rows, then v rows            (TAIL)
  const char* wpack;    // packed bf16 weights in streaming order (+ RING-1 slabs of padding)
  const float* vec;     // packed fp32 vectors (2048 floats): bout[D] b1'[M] b2[D] bk'[I] bv'[I], LayerNorm affines folded in
  int ntok;
  float eps;
  // EMBED variant (first layer): x is produced in-kernel from the token grid (local_3d_attention.py:140-157)
  const int64_t* z; const float *emb, *pos_s, *pos_h, *pos_w; int S, H, W, num_classes;
  int dbg;              // ablation switches (timing experiments only): 1 = skip MFMA loop, 2 = skip weight DMA + waits
  // trailing-planes variant: token t of the compact output grid reads row (t / rows_out) * rows_in + row0 + t % rows_out
  // of x (or of the token grid z); rows_out == 0: identity
  int rows_out, rows_in, row0;
  // x / x_out in the TILED stream layout (WMZ_FUSED_X_IN_TILED / _OUT_TILED): per 32-token tile [16 chunks][2 halves]
  // [32 tokens][8 features], i.e. lane (t, h)'s k-step s at ((2 s + h) * 32 + t) * 16 bytes -- every load / store
  // instruction of the kernel is then one contiguous KB.  Private to the fused path (layer -> layer); the last layer
  // writes row-major.
  int xflags;
  // training forward (wmz_*_train): what the backward needs, row-major -- x1 (input of the feed-forward block), a row-major
  // copy of x_out beside the tiled one, and k | v as the column halves of one [ntok, 2I] buffer
  bf16_t* x1o; bf16_t* xo_rm; int kv_combined;
  // ... and the LayerNorm statistics the kernel computes anyway (fp32 [2, ntok]: the means, then the reciprocal standard
  // deviations): st_ff of LN2(x1) in front of the feed-forward, st_attn of LN1'(x2) in front of the next layer's k | v
  float* st_ff; float* st_attn;
  // ... and the feed-forward pre-activation z = W1' LN2(x1) + b1' for the fused backward (layer_fused_bwd.hip), in a private
  // TILED layout: per 32-token tile [M/32 chunks][2][64 lanes][8] -- lane (t, h)'s accumulator registers 8j .. 8j+7 of chunk c
  // (hidden units 32c + 16h + 8j ..) at ((2c + j) * 64 + lane) * 16 bytes: every store instruction is one contiguous KB
  bf16_t* zt;
  long long* ts;        // timing probe (wmz_debug_fused_timestamps): workgroup 0 writes s_memtime at stage boundaries
};
// stage-boundary probe: wave w of workgroup 0 stores the shader clock into ts[w * 64 + slot]
#define WMZ_TS(slot)                                                                                      \
  do {                                                                                                    \
    if (P.ts != nullptr && blockIdx.x == 0 && lane == 0) P.ts[wave * 64 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
__device__ __forceinline__ long src_row(const FusedParams& P, long t) {
  if (P.rows_out == 0) return t;
  const int c = (int)t / P.rows_out;
  return (long)c * P.rows_in + P.row0 + ((int)t - c * P.rows_out);
}

// token + 3-axis position embedding of the lane's half row, rounded to bf16 (the value the stream carries)
template <int F>
__device__ __forceinline__ void embed_bop(Frag8<bf16_t> (&bop)[F / 16], const FusedParams& P, long t, bool ok, int h) {
  (void)ok;                                              // t is already clamped to a valid token
  const long ts = src_row(P, t);
  const int w = (int)(ts % P.W), hh = (int)((ts / P.W) % P.H), s = (int)((ts / ((long)P.W * P.H)) % P.S);
  long tk = P.z[ts];
  tk = tk < 0 ? 0 : (tk >= P.num_classes ? P.num_classes - 1 : tk);
  const float* e = P.emb + tk * F + h * (F / 2);
  const float* a = P.pos_s + (long)s * F + h * (F / 2);
  const float* b = P.pos_h + (long)hh * F + h * (F / 2);
  const float* d = P.pos_w + (long)w * F + h * (F / 2);
#pragma unroll
  for (int c = 0; c < F / 16; ++c) {
    float y[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 ev = *reinterpret_cast<const f32x4*>(e + 8 * c + 4 * q);
      const f32x4 av = *reinterpret_cast<const f32x4*>(a + 8 * c + 4 * q);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b + 8 * c + 4 * q);
      const f32x4 dv = *reinterpret_cast<const f32x4*>(d + 8 * c + 4 * q);
      const f32x4 yv = ev + ((av + bv) + dv);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[4 * q + r] = yv[r];
    }
    pack8(bop[c], y);
    if ((c & 1) == 1) WMZ_FENCE();     // 32 loads in flight, not 128
  }
}

// The same for 16-wide planes cut into whole 32-token tiles (tile = two plane rows), cooperatively: the wave walks its 32
// tokens one at a time, all 64 lanes fetching ONE 1 KB embedding row (4 features per lane, coalesced; the lane-per-token
// form above gathers 64 different rows per instruction), adds the position rows (plane and the two plane-row terms
// hoisted, the 16 column rows kept in registers), and drops the bf16 row into the wave's LDS image; the lanes then pick
// up their own half rows.  Same fp32 sum order, e + ((s + h) + w): bit-identical to embed_bop.
template <int F>
__device__ __forceinline__ void embed_coop(Frag8<bf16_t> (&bop)[F / 16], char* stg, const FusedParams& P, long tok0, int lane) {
  static_assert(F == 256, "one 1 KB fp32 row per wave instruction");
  const long last = P.ntok - 1;
  const long ts0 = src_row(P, tok0 < last ? tok0 : last);                 // first token of the tile in the full grid
  const int h0 = (int)((ts0 / 16) % P.H), s = (int)((ts0 / (16L * P.H)) % P.S);
  const int h1 = h0 + 1 < P.H ? h0 + 1 : h0;
  const f32x4 ps = *reinterpret_cast<const f32x4*>(P.pos_s + (long)s * F + lane * 4);
  const f32x4 a0 = ps + *reinterpret_cast<const f32x4*>(P.pos_h + (long)h0 * F + lane * 4);
  const f32x4 a1 = ps + *reinterpret_cast<const f32x4*>(P.pos_h + (long)h1 * F + lane * 4);
  f32x4 pw[16];
#pragma unroll
  for (int w = 0; w < 16; ++w) pw[w] = *reinterpret_cast<const f32x4*>(P.pos_w + (long)w * F + lane * 4);
  const int t = lane & 31, hl = lane >> 5;
  // the tile's 32 token ids: one coalesced load (lane t holds token t's id, clamped), handed out by v_readlane
  int tkv;
  {
    const long tt = tok0 + t < last ? tok0 + t : last;
    long tk = P.z[ts0 + (tt - (tok0 < last ? tok0 : last))];
    tk = tk < 0 ? 0 : (tk >= P.num_classes ? P.num_classes - 1 : tk);
    tkv = (int)tk;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {                                           // 16 tokens (one plane row) per pass
    const f32x4 a = p ? a1 : a0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long tk = __builtin_amdgcn_readlane(tkv, 16 * p + i);
      const f32x4 e = *reinterpret_cast<const f32x4*>(P.emb + tk * F + lane * 4);
      const f32x4 y = e + (a + pw[i]);
      s16x4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(y[r]);
      *reinterpret_cast<s16x4*>(stg + i * 512 + ((((lane >> 1) ^ i) << 4) | ((lane & 1) << 3))) = pk;
    }
    if ((t >> 4) == p) {
      const int tl = t & 15;
#pragma unroll
      for (int c = 0; c < F / 16; ++c)
        bop[c].v = *reinterpret_cast<const s16x8*>(stg + tl * 512 + (((hl * 16 + c) ^ tl) << 4));
    }
  }
}

__device__ __forceinline__ void store_z_tiled(bf16_t* zt, long tok0, int c, const f32x16& z, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = z[8 * j + i];
    Frag8<bf16_t> f;
    pack8(f, y);
    *reinterpret_cast<s16x8*>(zt + tok0 * 256 + ((2 * c + j) * 64 + lane) * 8) = f.v;
  }
}

template <int D, int I, int M, bool HEAD, bool TAIL>
__global__ __launch_bounds__(NTHR, 8 / FW) void layer_fused_kernel(FusedParams P) {
  static_assert(D == 256 && M == 256 && I == 128, "built for the default denoiser widths");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  float* vecs = reinterpret_cast<float*>(smem);
  const char* ring0 = smem + VECB;
  WStream ws;
  ws.ring = smem + VECB + wave * (SLAB / FW);
  ws.src = P.wpack + wave * (SLAB / FW) + lane * 16;
  ws.half = 0;
  ws.issue_slot = 0;
  ws.cur = 0;
  ws.dbg = P.dbg;
  ws.tot = ws.t1 = ws.t2 = ws.t3 = 0;
  ws.all = 0;
  ws.wave = wave;
  ws.probe = 0;
  ws.ts = (P.ts != nullptr && blockIdx.x == 0 && lane == 0) ? P.ts + wave * 64 : nullptr;
  WMZ_TS(0);
  const long tok0 = (long)blockIdx.x * (TW * FW) + wave * TW;        // first token of this wave
  const long tok = tok0 + (lane & 31);
  const long tokc = tok < P.ntok ? tok : P.ntok - 1;                  // clamped: loads are unconditional
  // tiled stream: ntok is a whole number of 32-token tiles, a wave's tile is all valid or all beyond the end -- a wave
  // beyond the end re-reads the last tile and stores nothing
  const bool tile_ok = tok0 < P.ntok;
  const long tok0c = tile_ok ? tok0 : (P.ntok >= 32 ? P.ntok - 32 : 0);
  char* stg = smem + VECB + RING * SLAB + wave * 8192;                // this wave's store-staging image
  static_assert(VECB == 8192 && 8 % FW == 0, "vector block is 2048 floats, 8 / FW KB per wave");
  const float* v_bout = vecs + h * (D / 2);                            // bout[D] b1'[M] b2[D] bk'[I] bv'[I]
  const float* v_b1 = vecs + D + h * (MC / 2);
  const float* v_b2 = vecs + D + M + h * (D / 2);
  const float* v_bk = vecs + 2 * D + M + h * (I / 2);
  const float* v_bv = v_bk + I;

  Frag8<bf16_t> qkvb[I / 16];                // q rows on their way out (TAIL)
  f32x16 xr[D / 32];                         // the residual stream of this lane's token (its half of the features), fp32
  Frag8<bf16_t> xb[D / 16];                  // a D-feature bf16 operand: LN output / the stream as stored
  if constexpr (HEAD) {
    // Everything the chain needs is requested up front, most urgent first, and nothing is waited for as a whole: the o
    // tile and the vectors by LDS-DMA, three weight slabs, then the residual rows (needed only after the first GEMM) by
    // loads the compiler does not track (it would drain the whole queue, weight ring included, at their first use).
    stage_dma128(stg, P.o, I, tok0, P.ntok, lane);                     // o tile -> the wave's LDS image (coalesced)
    vec_dma(vecs, P.vec, wave, lane);
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) ws_issue(ws);                   // prime: RING-1 slabs in flight
    const bool xt = (P.xflags & WMZ_FUSED_X_IN_TILED) != 0;
    {
      const bf16_t* xp = xt ? P.x + src_row(P, tok0c) * D + lane * 8 : P.x + src_row(P, tokc) * D + h * (D / 2);
      const int xs = xt ? 64 * 8 : 8;                                  // elements between consecutive k-steps
#pragma unroll
      for (int s = 0; s < D / 16; ++s) xb[s].v = gload_untracked(xp + s * xs);
    }
    ws_extra(ws, D / 16);                                             // 16 younger loads sit behind the primed slabs
    WMZ_TS(1);
    if (WPP * (RING - 1) == 6) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");   // o tile + vectors landed (the primed slab
    else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");                          // pieces + 16 loads may fly)
    static_assert(WPP * (RING - 1) == 6 || WPP * (RING - 1) == 4, "literals above");
    __builtin_amdgcn_s_barrier();                                      // everyone's share of the vectors did
    WMZ_TS(44);
    init_vec<D / 32>(xr, v_bout);
    WMZ_TS(2);
    {
      const int t = lane & 31;
      gemm_stage_b<D / 32, I / 16>(xr, [&](int ks) { return stage_get(stg, t, h * (I / 16) + ks); }, ring0, ws, lane);
    }                                                                  // o Wout^T + bout
    wait_untracked<D / 16>(xb);                                        // the residual rows are older than the ring's 6 pieces
    WMZ_TS(45);
    add_bop<D / 32>(xr, xb);                                           // + x  -> x1
    const bool x1_hat = (P.xflags & WMZ_FUSED_X1_NORMALISED) != 0;
    if (P.x1o != nullptr && !x1_hat) {                                 // training, op-by-op backward: x1 itself
      bop_from_acc<D / 32>(xb, xr);
      store_tile256(stg, P.x1o, tok0, P.ntok, xb, lane);
      ws_extra(ws, 16);
    }
    WMZ_TS(3);
    ln_to_bop<D / 32>(xb, xr, P.eps, P.st_ff, tok, P.ntok, lane);   // LN2(x1)
    if (P.st_ff != nullptr) ws_extra(ws, 2);
    if (P.x1o != nullptr && x1_hat) {                                  // training, fused backward: the NORMALISED rows -- all the
      store_tile256(stg, P.x1o, tok0, P.ntok, xb, lane);               // backward needs of x1 (LayerNorm backward and the dW1
      ws_extra(ws, 16);                                                // operand), already in registers as the W1 operand
    }
    WMZ_TS(4);
    // feed-forward, MC hidden units at a time: W1[c] -> GELU -> W2[c], with the stream packed as W1[0], W1[1], W2[0], W1[2],
    // W2[1], .., W1[7], W2[6], W2[7]: GELU(c) is VALU work that rides under the MFMAs of the two stages between W1[c] and
    // W2[c] -- its first half under W2[c-1], its second half under W1[c+1] -- so no stage is VALU-bound.  The upper half
    // of the LN2 operand is parked in the wave's LDS image for the duration (the registers carry two chunk accumulators).
#pragma unroll
    for (int sx = 8; sx < 16; ++sx) frag_park(stg, sx - 8, xb[sx], lane);
    {
      auto lnb = [&](int sx) { return sx < 8 ? xb[sx] : frag_unpark(stg, sx - 8, lane); };
      auto gelu_n = [&](float (&y)[8], const f32x16& zz, int first, int yo, int n) {   // n values zz[first..] -> y[yo..]
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < n) y[yo + j] = wmz_gelu_fast(zz[first + j]);
      };
      f32x16 zc[1], zn[1];
      Frag8<bf16_t> gb[2], gn[2];
      float y[8];
      init_vec<1>(zc, v_b1);
      gemm_stage_b<1, D / 16>(zc, lnb, ring0, ws, lane);                           // z0 = b1[0] + W1[0] LN2(x1)
      const bool zsave = P.zt != nullptr && tile_ok;
      if (zsave) { store_z_tiled(P.zt, tok0, 0, zc[0], lane); ws_extra(ws, 2); }
      WMZ_TS(5);
      init_vec<1>(zn, v_b1 + MC);
      gemm_stage_b<1, D / 16>(zn, lnb, ring0, ws, lane, side_work<9>([&](int g) {  // z1 | GELU(0), all of it
        gelu_n(y, zc[0], 4 * g, (g & 1) * 4, 4);
        if (g & 1) pack8(gb[g >> 1], y);
      }));
      if (zsave) { store_z_tiled(P.zt, tok0, 1, zn[0], lane); ws_extra(ws, 2); }
      zc[0] = zn[0];
      WMZ_TS(6);
#pragma unroll 1
      for (int c = 1; c < M / MC - 1; ++c) {                                       // zc = z_c, gb = GELU(c-1)
        if (c == 4) ws.probe = 48;
        gemm_stage<D / 32, MC / 16>(xr, gb, ring0, ws, lane, side_work<5>([&](int g) {   // x1 += W2[:, c-1] GELU(c-1) | GELU(c) 0..7
          gelu_n(y, zc[0], 2 * g, 2 * g, 2);
          if (g == 3) pack8(gn[0], y);
        }));
        ws.probe = 0;
        WMZ_TS(5 + 3 * c);
        init_vec<1>(zn, v_b1 + (c + 1) * MC);
        if (c == 4) ws.probe = 52;
        gemm_stage_b<1, D / 16>(zn, lnb, ring0, ws, lane, side_work<5>([&](int g) {      // z_{c+1} | GELU(c) 8..15
          gelu_n(y, zc[0], 8 + 2 * g, 2 * g, 2);
          if (g == 3) pack8(gn[1], y);
        }));
        ws.probe = 0;
        if (zsave) { store_z_tiled(P.zt, tok0, c + 1, zn[0], lane); ws_extra(ws, 2); }
        WMZ_TS(7 + 3 * c);
        zc[0] = zn[0];
        gb[0] = gn[0];
        gb[1] = gn[1];
      }
      gemm_stage<D / 32, MC / 16>(xr, gb, ring0, ws, lane, side_work<9>([&](int g) {     // x1 += W2[:, 6] GELU(6) | GELU(7), all of it
        gelu_n(y, zc[0], 4 * g, (g & 1) * 4, 4);
        if (g & 1) pack8(gn[g >> 1], y);
      }));
      gemm_stage<D / 32, MC / 16>(xr, gn, ring0, ws, lane);                        // x1 += W2[:, 7] GELU(7)
      WMZ_TS(29);
    }
    add_vec<D / 32>(xr, v_b2);                                         //                 -> x2
    Frag8<bf16_t> x2b[D / 16];                                         // x2 as the stream carries it
    if constexpr (TAIL) {                                              // + LN1'(x2), while x2 is still fp32
      ln_and_pack<D / 32>(xb, x2b, xr, P.eps, P.st_attn, tok, P.ntok, lane);
      if (P.st_attn != nullptr) ws_extra(ws, 2);
    }
    else bop_from_acc<D / 32>(x2b, xr);
    WMZ_TS(30);
    if (P.xflags & WMZ_FUSED_X_OUT_TILED) { if (tile_ok) store_bop_tiled<D / 16>(P.xo + tok0 * D, x2b, lane); }
    else store_tile256(stg, P.xo, tok0, P.ntok, x2b, lane);
    ws_extra(ws, 16);
    if (P.xo_rm != nullptr) {        // training: the row-major copy for the backward -- x2 itself, or (WMZ_FUSED_XRM_NORMALISED)
      if (TAIL && (P.xflags & WMZ_FUSED_XRM_NORMALISED)) store_tile256(stg, P.xo_rm, tok0, P.ntok, xb, lane);   // LN1'(x2)'s rows
      else store_tile256(stg, P.xo_rm, tok0, P.ntok, x2b, lane);
      ws_extra(ws, 16);
    }
    WMZ_TS(31);
    if constexpr (TAIL) {
      f32x16 qa[I / 32];
      zero_acc(qa);
      gemm_stage<I / 32, D / 16>(qa, x2b, ring0, ws, lane);            // to_q on the raw stream
      WMZ_TS(32);
      bop_from_acc<I / 32>(qkvb, qa);                                  // stored under the to_k MFMAs below
    }
  } else {
    if (P.z != nullptr) {                                            // first layer: x = embedding, also written to x_out
      const bool coop = P.W == 16 && (P.H * 16) % 32 == 0 && (P.rows_out == 0 || (P.rows_out % 32 == 0 && P.rows_in % 32 == 0 && P.row0 % 32 == 0));
      if (coop) embed_coop<D>(xb, stg, P, tok0, lane);
      else embed_bop<D>(xb, P, tokc, true, h);
    }
    else if (P.xflags & WMZ_FUSED_X_IN_TILED) load_bop_tiled<D / 16>(xb, P.x + src_row(P, tok0c) * D, lane);
    else load_bop<D / 16>(xb, P.x + src_row(P, tokc) * D + h * (D / 2));
    f32x4 vecv[8 / FW];
#pragma unroll
    for (int i = 0; i < 8 / FW; ++i) vecv[i] = *reinterpret_cast<const f32x4*>(P.vec + (tid + i * NTHR) * 4);
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) ws_issue(ws);
#pragma unroll
    for (int i = 0; i < 8 / FW; ++i) *reinterpret_cast<f32x4*>(vecs + (tid + i * NTHR) * 4) = vecv[i];
    __syncthreads();
    if (P.z != nullptr) {
      if (P.xflags & WMZ_FUSED_X_OUT_TILED) { if (tile_ok) store_bop_tiled<D / 16>(P.xo + tok0 * D, xb, lane); }
      else store_tile256(stg, P.xo, tok0, P.ntok, xb, lane);
      ws_extra(ws, 16);
      if (P.xo_rm != nullptr && !(P.xflags & WMZ_FUSED_XRM_NORMALISED)) { store_tile256(stg, P.xo_rm, tok0, P.ntok, xb, lane); ws_extra(ws, 16); }
    }
    {
      f32x16 qa[I / 32];
      zero_acc(qa);
      gemm_stage<I / 32, D / 16>(qa, xb, ring0, ws, lane);             // to_q on the raw stream
      bop_from_acc<I / 32>(qkvb, qa);
    }
    acc_from_bop<D / 32>(xr, xb);
    ln_to_bop<D / 32>(xb, xr, P.eps, P.st_attn, tok, P.ntok, lane);   // LN1'(x)
    if (P.st_attn != nullptr) ws_extra(ws, 2);
    if (P.xo_rm != nullptr && (P.xflags & WMZ_FUSED_XRM_NORMALISED)) { store_tile256(stg, P.xo_rm, tok0, P.ntok, xb, lane); ws_extra(ws, 16); }
  }
  if constexpr (TAIL) {
    f32x16 ka[I / 32];
    Frag8<bf16_t> kb[I / 16];
    init_vec<I / 32>(ka, v_bk);
    store_tile128(stg, P.q, I, tok0, P.ntok, 0, qkvb, lane);         // q rows (as side work under the to_k MFMAs: slower)
    ws_extra(ws, 8);
    WMZ_TS(33);
    gemm_stage<I / 32, D / 16>(ka, xb, ring0, ws, lane);               // to_k
    WMZ_TS(34);
    bop_from_acc<I / 32>(kb, ka);
    if (P.kv_combined) store_tile128(stg, P.kv, 2 * I, tok0, P.ntok, 0, kb, lane);   // k | v column halves of [ntok, 2I]
    else store_tile128(stg, P.kv, I, tok0, P.ntok, 0, kb, lane);       // k rows
    ws_extra(ws, 8);
    init_vec<I / 32>(ka, v_bv);
    WMZ_TS(35);
    gemm_stage<I / 32, D / 16>(ka, xb, ring0, ws, lane);               // to_v
    WMZ_TS(36);
    bop_from_acc<I / 32>(kb, ka);
    if (P.kv_combined) store_tile128(stg, P.kv, 2 * I, tok0, P.ntok, I, kb, lane);
    else store_tile128(stg, P.kv + (long)P.ntok * I, I, tok0, P.ntok, 0, kb, lane);    // v rows, behind the k rows
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the padding slabs still in flight target this workgroup's LDS
  WMZ_TS(37);
}

}  // namespace

static int fused_launch(FusedParams& P, int ntok, int D, int I, int M, int has_head, int has_tail, void* stream);
static long long* g_fused_ts = nullptr;
static int g_fused_dbg = 0;       // ablation switches (wmz_debug_fused_knobs): 1 = skip the MFMA loops, 2 = skip the weight DMA + waits
#ifndef WMZ_OP16_F16
extern "C" int wmz_debug_fused_knobs(int dbg) { g_fused_dbg = dbg; return WMZ_OK; }
// Timing probe for kernel development (tools/ts_fused.py): a device buffer of 8 * 64 int64; NULL switches it off.
extern "C" int wmz_debug_fused_timestamps(void* buf) { g_fused_ts = (long long*)buf; return WMZ_OK; }
#endif

extern "C" int WMZ_FN(wmz_layer_fused_fwd)(const void* o, const void* x, void* x_out, void* q_out, void* kv_out,
                                   const void* wpack, const float* vec, int ntok, int D, int I, int M, int has_head,
                                   int has_tail, float eps, void* stream) {
  return WMZ_FN(wmz_layer_fused_fwd_planes)(o, x, x_out, q_out, kv_out, wpack, vec, 1, 1, 1, ntok, D, I, M, has_head, has_tail, 0,
                                    eps, stream);
}


// ---- weight stream packer (host side of the kernel above used to be ~60 small torch launches per layer) ----------
namespace {
// element (f, k) of a block = w[f * rs + k * ks] (* gamma[k]) (* rgamma[f]); N output features, K contraction length.
// gn / gk: ownership groups -- within every group of gn output features (gk contraction indices) lane half 0 owns the
// first half and lane half 1 the second (forward: gn = N, gk = K: a lane owns one contiguous half row; the backward
// kernels use groups of 128, the width of an LDS-staged row tile, and of 32 for the hidden axis walked in chunks).
struct PackBlock { const float* w; long rs, ks; int N, K, gn, gk; const float* gamma; const float* rgamma; long dst; };
struct PackParams {
  PackBlock blk[24]; int nblk; long total;          // total bf16 elements of the stream (without the padding)
  bf16_t* wpack; long padded;
  // vec: bout | b1 + W1 be2 | b2 | Wk be1 | bv + Wv be1   (2048 floats, zero padded)
  const float *bout, *b1, *w1, *be2, *b2, *wk, *wv, *be1, *bv; float* vec; int D, I, M;
};
__global__ __launch_bounds__(256) void fused_pack_kernel(PackParams P) {
  const long e8 = ((long)blockIdx.x * 256 + threadIdx.x) * 8;          // 8 consecutive stream elements = one lane's fragment
  if (e8 < P.total) {
    int bi = 0;
#pragma unroll 1
    for (int i = 1; i < P.nblk; ++i) if (e8 >= P.blk[i].dst) bi = i;
    const PackBlock B = P.blk[bi];
    const long e = e8 - B.dst;
    const int NB = B.N / 32;
    const int piece = (int)(e / 512), lane = (int)((e % 512) / 8);
    const int s = piece / NB, b = piece % NB;
    const int r = lane & 31, h = lane >> 5;
    const int bpg = B.gn / 32, spg = B.gk / 16;
    // output feature of MFMA row r of block b (row r lands in lane half (r >> 2) & 1, accumulator register (r & 3) + 4 (r >> 3))
    const int f = (b / bpg) * B.gn + ((r >> 2) & 1) * (B.gn / 2) + (b % bpg) * 16 + (r & 3) + 4 * (r >> 3);
    const int k0 = (s / spg) * B.gk + h * (B.gk / 2) + (s % spg) * 8;
    s16x8 v;
    const float rg = B.rgamma ? B.rgamma[f] : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = B.w[(long)f * B.rs + (long)(k0 + j) * B.ks] * rg;
      if (B.gamma) x *= B.gamma[k0 + j];
      v[j] = (short)f32_to_bf16_bits(x);
    }
    *reinterpret_cast<s16x8*>(P.wpack + e8) = v;
  } else if (e8 < P.padded) {
    *reinterpret_cast<s16x8*>(P.wpack + e8) = (s16x8)(0);
  }
}
// the vector block: bout | b1 + W1 be2 | b2 | Wk be1 | bv + Wv be1.  One wave per entry: plain entries are a copy, the
// matrix-vector entries a coalesced 256-long dot (4 floats per lane) + wave reduction.
__global__ __launch_bounds__(256) void fused_pack_vec_kernel(PackParams P) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= 2048) return;
  const int D = P.D, I = P.I, M = P.M;
  const float* row = nullptr; const float* vecb = nullptr; float base = 0.f;
  if (t < D) base = P.bout ? P.bout[t] : 0.f;
  else if (t < D + M) { if (P.b1) { base = P.b1[t - D]; row = P.w1 + (long)(t - D) * D; vecb = P.be2; } }
  else if (t < 2 * D + M) base = P.b2 ? P.b2[t - D - M] : 0.f;
  else if (t < 2 * D + M + I) { if (P.wk) { row = P.wk + (long)(t - 2 * D - M) * D; vecb = P.be1; } }
  else if (t < 2 * D + M + 2 * I) { if (P.wv) { base = P.bv[t - 2 * D - M - I]; row = P.wv + (long)(t - 2 * D - M - I) * D; vecb = P.be1; } }
  float acc = 0.f;
  if (row != nullptr) {                                   // wave-uniform
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + lane * 4), b = *reinterpret_cast<const f32x4*>(vecb + lane * 4);
    acc = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    acc = wave_sum(acc);
  }
  if (lane == 0) P.vec[t] = base + acc;
}
}  // namespace

extern "C" int WMZ_FN(wmz_layer_fused_pack)(const float* wout, const float* bout, const float* g2, const float* be2, const float* w1,
                                    const float* b1, const float* w2, const float* b2, const float* g1, const float* be1,
                                    const float* wq, const float* wk, const float* wv, const float* bv, void* wpack,
                                    float* vec, int D, int I, int M, void* stream) {
  const bool head = wout != nullptr, tail = wq != nullptr;
  WMZ_REQUIRE(wpack && vec && (head || tail), "wmz_layer_fused_pack: bad arguments");
  WMZ_REQUIRE(D == 256 && I == 128 && M == 256, "wmz_layer_fused_pack: built for dim 256 / inner 128 / mlp 256");
  WMZ_REQUIRE(!head || (bout && g2 && be2 && w1 && b1 && w2 && b2), "wmz_layer_fused_pack: head parameters missing");
  WMZ_REQUIRE(!tail || (g1 && be1 && wk && wv && bv), "wmz_layer_fused_pack: tail parameters missing");
  PackParams P;
  int n = 0;
  long off = 0;
  auto add = [&](const float* w, long rs, int N, int K, const float* gamma) {
    P.blk[n].w = w; P.blk[n].rs = rs; P.blk[n].ks = 1; P.blk[n].N = N; P.blk[n].K = K; P.blk[n].gn = N; P.blk[n].gk = K;
    P.blk[n].gamma = gamma; P.blk[n].rgamma = nullptr; P.blk[n].dst = off;
    off += (long)N * K; ++n;
  };
  constexpr int MCH = 32;
  if (head) {
    add(wout, I, D, I, nullptr);
    add(w1, D, MCH, D, g2);                                              // W1[0]
    for (int c = 1; c < M / MCH; ++c) {
      add(w1 + (long)c * MCH * D, D, MCH, D, g2);                        // W1[c]
      add(w2 + (c - 1) * MCH, M, D, MCH, nullptr);                       // W2[:, c-1]
    }
    add(w2 + (M / MCH - 1) * MCH, M, D, MCH, nullptr);
  }
  if (tail) {
    add(wq, D, I, D, nullptr);
    add(wk, D, I, D, g1);
    add(wv, D, I, D, g1);
  }
  P.nblk = n; P.total = off; P.padded = off + 65536 / 2;                 // + 64 KB of zeros behind the stream
  P.wpack = (bf16_t*)wpack;
  P.bout = head ? bout : nullptr; P.b1 = head ? b1 : nullptr; P.w1 = w1; P.be2 = be2; P.b2 = head ? b2 : nullptr;
  P.wk = tail ? wk : nullptr; P.wv = tail ? wv : nullptr; P.be1 = be1; P.bv = bv; P.vec = vec; P.D = D; P.I = I; P.M = M;
  hipLaunchKernelGGL(fused_pack_kernel, dim3((unsigned)wmz_cdiv(P.padded / 8, 256)), dim3(256), 0, (hipStream_t)stream, P);
  hipLaunchKernelGGL(fused_pack_vec_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, P);
  WMZ_LAUNCH_CHECK("wmz_layer_fused_pack");
  return WMZ_OK;
}

// ---- every weight stream and vector block of a whole model in TWO launches (training: after each optimizer step) ----------
// The per-boundary entry points above cost a launch pair per stream (13 + 5 launches of ~7 us per config-4 step).  Here the
// block descriptors live in DEVICE memory, built once by the host (fused.PackSet): table rows of eleven 64-bit fields
//   { w, rs, ks, N, K, gn, gk, gamma, rgamma, dst, start8 }      (PackBlock above; dst = the block's first stream element,
//                                                                  start8 = first 8-element group of the block in the
//                                                                  launch's global numbering; one extra row = the end)
// and vector jobs of ten 64-bit fields { bout, b1, w1, be2, b2, wk, wv, be1, bv, vec } (fused_pack_vec_kernel's inputs).
namespace {
struct PackRowG { const float* w; long rs, ks, N, K, gn, gk; const float* gamma; const float* rgamma; bf16_t* dst; long start8; };
struct VecJobG { const float *bout, *b1, *w1, *be2, *b2, *wk, *wv, *be1, *bv; float* vec; };
static_assert(sizeof(PackRowG) == 88 && sizeof(VecJobG) == 80, "table layouts are part of the C ABI");

__global__ __launch_bounds__(256) void fused_pack_table_kernel(const PackRowG* __restrict__ rows, int nblk, long total8) {
  const long g8 = (long)blockIdx.x * 256 + threadIdx.x;
  if (g8 >= total8) return;
  int lo = 0, hi = nblk - 1;                             // last block with start8 <= g8
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rows[mid].start8 <= g8) lo = mid; else hi = mid - 1;
  }
  const PackRowG B = rows[lo];
  const long e = (g8 - B.start8) * 8;
  const int NB = (int)B.N / 32;
  const int piece = (int)(e / 512), lane = (int)((e % 512) / 8);
  const int s = piece / NB, b = piece % NB;
  const int r = lane & 31, h = lane >> 5;
  const int bpg = (int)B.gn / 32, spg = (int)B.gk / 16;
  const int f = (b / bpg) * (int)B.gn + ((r >> 2) & 1) * ((int)B.gn / 2) + (b % bpg) * 16 + (r & 3) + 4 * (r >> 3);
  const int k0 = (s / spg) * (int)B.gk + h * ((int)B.gk / 2) + (s % spg) * 8;
  const float rg = B.rgamma ? B.rgamma[f] : 1.f;
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = B.w[(long)f * B.rs + (long)(k0 + j) * B.ks] * rg;
    if (B.gamma) x *= B.gamma[k0 + j];
    v[j] = (short)f32_to_bf16_bits(x);
  }
  *reinterpret_cast<s16x8*>(B.dst + e) = v;
}

__global__ __launch_bounds__(256) void fused_pack_vec_table_kernel(const VecJobG* __restrict__ jobs, int D, int I, int M) {
  const VecJobG P = jobs[blockIdx.y];
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= 2048) return;
  const float* row = nullptr; const float* vecb = nullptr; float base = 0.f;
  if (t < D) base = P.bout ? P.bout[t] : 0.f;
  else if (t < D + M) { if (P.b1) { base = P.b1[t - D]; row = P.w1 + (long)(t - D) * D; vecb = P.be2; } }
  else if (t < 2 * D + M) base = P.b2 ? P.b2[t - D - M] : 0.f;
  else if (t < 2 * D + M + I) { if (P.wk) { row = P.wk + (long)(t - 2 * D - M) * D; vecb = P.be1; } }
  else if (t < 2 * D + M + 2 * I) { if (P.wv) { base = P.bv[t - 2 * D - M - I]; row = P.wv + (long)(t - 2 * D - M - I) * D; vecb = P.be1; } }
  float acc = 0.f;
  if (row != nullptr) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + lane * 4), b = *reinterpret_cast<const f32x4*>(vecb + lane * 4);
    acc = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    acc = wave_sum(acc);
  }
  if (lane == 0) P.vec[t] = base + acc;
}
}  // namespace

extern "C" int WMZ_FN(wmz_fused_pack_table)(const void* block_rows, int nblk, long total8, const void* vec_jobs, int nvec, int D, int I,
                                    int M, void* stream) {
  WMZ_REQUIRE(nblk >= 0 && nvec >= 0 && (nblk == 0 || (block_rows && total8 > 0)) && (nvec == 0 || vec_jobs),
              "wmz_fused_pack_table: bad arguments");
  WMZ_REQUIRE(D == 256 && I == 128 && M == 256, "wmz_fused_pack_table: built for dim 256 / inner 128 / mlp 256");
  if (nblk > 0)
    hipLaunchKernelGGL(fused_pack_table_kernel, dim3((unsigned)wmz_cdiv(total8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const PackRowG*)block_rows, nblk, total8);
  if (nvec > 0)
    hipLaunchKernelGGL(fused_pack_vec_table_kernel, dim3(512, (unsigned)nvec), dim3(256), 0, (hipStream_t)stream,
                       (const VecJobG*)vec_jobs, D, I, M);
  WMZ_LAUNCH_CHECK("wmz_fused_pack_table");
  return WMZ_OK;
}

#ifndef WMZ_OP16_F16      // (training forward + backward streams: the bf16 unit only)
// Weight streams of the fused BACKWARD kernels (layer_fused_bwd.hip), TRANSPOSED blocks in consumption order:
//   wpack_qkv:  Wk'^T | Wv'^T | Wq^T          ([D x I] each; ' = the attention LayerNorm's gamma folded in: rows scaled)
//   wpack_ff:   W2^T[c] (c = 0 .. M/32-1: [32 x D]) | W1'^T [D x M] | Wout^T [I x D]
// Either may be NULL.  Each stream is followed by 64 KB of zeros (the kernels' prefetch runs past the end).
extern "C" int wmz_layer_fused_bwd_pack(const float* wq, const float* wk, const float* wv, const float* g1, const float* wout,
                                        const float* w1, const float* g2, const float* w2, void* wpack_qkv, void* wpack_ff,
                                        int D, int I, int M, void* stream) {
  WMZ_REQUIRE(D == 256 && I == 128 && M == 256, "wmz_layer_fused_bwd_pack: built for dim 256 / inner 128 / mlp 256");
  WMZ_REQUIRE(wpack_qkv || wpack_ff, "wmz_layer_fused_bwd_pack: nothing to do");
  WMZ_REQUIRE(!wpack_qkv || (wq && wk && wv && g1), "wmz_layer_fused_bwd_pack: attention parameters missing");
  WMZ_REQUIRE(!wpack_ff || (wout && w1 && g2 && w2), "wmz_layer_fused_bwd_pack: feed-forward parameters missing");
  for (int which = 0; which < 2; ++which) {
    void* dst = which == 0 ? wpack_qkv : wpack_ff;
    if (dst == nullptr) continue;
    PackParams P;
    int n = 0;
    long off = 0;
    // element (f, k) = w[f + k * ld]: the transpose of a row-major [K, ld] matrix
    auto addT = [&](const float* w, long ld, int N, int K, int gn, int gk, const float* rgamma) {
      P.blk[n].w = w; P.blk[n].rs = 1; P.blk[n].ks = ld; P.blk[n].N = N; P.blk[n].K = K; P.blk[n].gn = gn; P.blk[n].gk = gk;
      P.blk[n].gamma = nullptr; P.blk[n].rgamma = rgamma; P.blk[n].dst = off;
      off += (long)N * K; ++n;
    };
    if (which == 0) {
      addT(wk, D, D, I, 128, 128, g1);
      addT(wv, D, D, I, 128, 128, g1);
      addT(wq, D, D, I, 128, 128, nullptr);
    } else {
      for (int c = 0; c < M / 32; ++c) addT(w2 + c * 32, M, 32, D, 32, 128, nullptr);     // dg_c = W2[:, c]^T dy
      addT(w1, D, D, M, 128, 32, g2);                                                      // dxhat = W1'^T dz
      addT(wout, I, I, D, 128, 128, nullptr);                                              // do = Wout^T dx1
    }
    P.nblk = n; P.total = off; P.padded = off + 65536 / 2;
    P.wpack = (bf16_t*)dst;
    P.bout = P.b1 = P.w1 = P.be2 = P.b2 = P.wk = P.wv = P.be1 = P.bv = nullptr; P.vec = nullptr; P.D = D; P.I = I; P.M = M;
    hipLaunchKernelGGL(fused_pack_kernel, dim3((unsigned)wmz_cdiv(P.padded / 8, 256)), dim3(256), 0, (hipStream_t)stream, P);
  }
  WMZ_LAUNCH_CHECK("wmz_layer_fused_bwd_pack");
  return WMZ_OK;
}

extern "C" int wmz_layer_fused_fwd_train(const void* o, const void* x, void* x_out, void* x_out_rowmajor, void* x1_out,
                                         void* q_out, void* kv_out, float* ln_ff_stats, float* ln_attn_stats,
                                         void* z_tiled_out, const void* wpack, const float* vec, int ntok, int D, int I,
                                         int M, int has_head, int has_tail, int xflags, float eps, void* stream) {
  WMZ_REQUIRE(x && wpack && vec && ntok > 0, "wmz_layer_fused_fwd_train: bad arguments");
  WMZ_REQUIRE(has_head || has_tail, "wmz_layer_fused_fwd_train: nothing to do");
  WMZ_REQUIRE(!has_head || (o && x_out), "wmz_layer_fused_fwd_train: head needs o and x_out");
  WMZ_REQUIRE(!has_tail || (q_out && kv_out), "wmz_layer_fused_fwd_train: tail needs q_out and kv_out");
  WMZ_REQUIRE((xflags & ~15) == 0 && ((xflags & 3) == 0 || ntok % 32 == 0), "wmz_layer_fused_fwd_train: bad layout flags");
  WMZ_REQUIRE(!(xflags & WMZ_FUSED_XRM_NORMALISED) || (has_tail && x_out_rowmajor && (xflags & WMZ_FUSED_X_OUT_TILED)),
              "wmz_layer_fused_fwd_train: WMZ_FUSED_XRM_NORMALISED needs the tail, x_rm_out and a tiled x_out (the raw stream must survive somewhere)");
  WMZ_REQUIRE(!(xflags & WMZ_FUSED_X1_NORMALISED) || (has_head && x1_out), "wmz_layer_fused_fwd_train: WMZ_FUSED_X1_NORMALISED needs the head and x1_out");
  FusedParams P;
  P.o = (const bf16_t*)o; P.x = (const bf16_t*)x; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.ntok = ntok; P.eps = eps;
  P.z = nullptr; P.emb = P.pos_s = P.pos_h = P.pos_w = nullptr; P.S = P.H = P.W = P.num_classes = 0;
  P.rows_out = P.rows_in = P.row0 = 0;
  P.xflags = xflags;
  P.x1o = (bf16_t*)x1_out; P.xo_rm = (bf16_t*)x_out_rowmajor; P.kv_combined = 1;
  P.st_ff = ln_ff_stats; P.st_attn = has_tail ? ln_attn_stats : nullptr;
  WMZ_REQUIRE(z_tiled_out == nullptr || (has_head && ntok % 32 == 0), "wmz_layer_fused_fwd_train: z_tiled_out needs the head and whole 32-token tiles");
  P.zt = (bf16_t*)z_tiled_out;
  return fused_launch(P, ntok, D, I, M, has_head, has_tail, stream);
}

extern "C" int wmz_embed_qkv_fused_fwd_train(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                             const float* pos_w, void* x_out, void* x_out_rowmajor, void* q_out,
                                             void* kv_out, float* ln_attn_stats, const void* wpack, const float* vec, int B,
                                             int S, int H, int W, int D, int I, int M, int num_classes, int xflags, float eps,
                                             void* stream) {
  WMZ_REQUIRE(z && emb && pos_s && pos_h && pos_w && x_out && q_out && kv_out && wpack && vec, "wmz_embed_qkv_fused_fwd_train: null tensor");
  WMZ_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && num_classes > 0 && (long)B * S * H * W < (1L << 31), "wmz_embed_qkv_fused_fwd_train: bad shape");
  WMZ_REQUIRE((xflags & ~(WMZ_FUSED_X_OUT_TILED | WMZ_FUSED_XRM_NORMALISED)) == 0 && (xflags == 0 || ((long)S * H * W) % 32 == 0), "wmz_embed_qkv_fused_fwd_train: bad layout flags");
  WMZ_REQUIRE(!(xflags & WMZ_FUSED_XRM_NORMALISED) || (x_out_rowmajor && (xflags & WMZ_FUSED_X_OUT_TILED)),
              "wmz_embed_qkv_fused_fwd_train: WMZ_FUSED_XRM_NORMALISED needs x_out_rowmajor and a tiled x_out");
  FusedParams P;
  P.o = nullptr; P.x = nullptr; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.ntok = B * S * H * W; P.eps = eps;
  P.rows_out = P.rows_in = P.row0 = 0;
  P.z = z; P.emb = emb; P.pos_s = pos_s; P.pos_h = pos_h; P.pos_w = pos_w; P.S = S; P.H = H; P.W = W; P.num_classes = num_classes;
  P.xflags = xflags;
  P.x1o = nullptr; P.xo_rm = (bf16_t*)x_out_rowmajor; P.kv_combined = 1;
  P.st_ff = nullptr; P.st_attn = ln_attn_stats; P.zt = nullptr;
  return fused_launch(P, P.ntok, D, I, M, 0, 1, stream);
}

#endif  // WMZ_OP16_F16

extern "C" int WMZ_FN(wmz_layer_fused_fwd_planes)(const void* o, const void* x, void* x_out, void* q_out, void* kv_out,
                                          const void* wpack, const float* vec, int B, int planes_out, int planes_in, int HW,
                                          int D, int I, int M, int has_head, int has_tail, int xflags, float eps,
                                          void* stream) {
  WMZ_REQUIRE((xflags & ~3) == 0, "wmz_layer_fused_fwd: bad layout flags");
  WMZ_REQUIRE(xflags == 0 || ((long)planes_out * HW) % 32 == 0 && ((long)planes_in * HW) % 32 == 0,
              "wmz_layer_fused_fwd: the tiled stream layout needs whole 32-token tiles per clip");
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
  P.xflags = xflags;
  P.x1o = nullptr; P.xo_rm = nullptr; P.kv_combined = 0;
  P.st_ff = P.st_attn = nullptr; P.zt = nullptr;
  return fused_launch(P, ntok, D, I, M, has_head, has_tail, stream);
}

extern "C" int WMZ_FN(wmz_embed_qkv_fused_fwd)(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                       const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                       const float* vec, int B, int S, int H, int W, int D, int I, int M, int num_classes,
                                       float eps, void* stream) {
  return WMZ_FN(wmz_embed_qkv_fused_fwd_planes)(z, emb, pos_s, pos_h, pos_w, x_out, q_out, kv_out, wpack, vec, B, S, H, W, S, D, I, M,
                                        num_classes, 0, eps, stream);
}

extern "C" int WMZ_FN(wmz_embed_qkv_fused_fwd_planes)(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                              const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                              const float* vec, int B, int S, int H, int W, int planes_out, int D, int I,
                                              int M, int num_classes, int xflags, float eps, void* stream) {
  WMZ_REQUIRE((xflags & ~WMZ_FUSED_X_OUT_TILED) == 0, "wmz_embed_qkv_fused_fwd: bad layout flags");
  WMZ_REQUIRE(xflags == 0 || ((long)planes_out * H * W) % 32 == 0, "wmz_embed_qkv_fused_fwd: the tiled stream layout needs whole 32-token tiles per clip");
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
  P.xflags = xflags;
  P.x1o = nullptr; P.xo_rm = nullptr; P.kv_combined = 0;
  P.st_ff = P.st_attn = nullptr; P.zt = nullptr;
  return fused_launch(P, P.ntok, D, I, M, 0, 1, stream);
}

static int fused_launch(FusedParams& P, int ntok, int D, int I, int M, int has_head, int has_tail, void* stream) {
  if (!(D == 256 && I == 128 && M == 256)) {
    wmz_set_error("wmz_layer_fused_fwd: built for dim 256 / inner 128 / mlp 256 (got %d/%d/%d); use the unfused path", D, I, M);
    return WMZ_ERR_UNSUPPORTED;
  }
  P.dbg = g_fused_dbg;
  P.ts = g_fused_ts;
  const size_t smem = VECB + RING * SLAB + FW * 8192;
  dim3 grid((unsigned)wmz_cdiv(ntok, TW * FW)), block(NTHR);
  hipStream_t st = (hipStream_t)stream;
#define WMZ_FUSED(H, T)                                                                                            \
  do {                                                                                                             \
    auto kern = layer_fused_kernel<256, 128, 256, H, T>;                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(kern, grid, block, smem, st, P);                                                            \
  } while (0)
  if (has_head && has_tail) WMZ_FUSED(true, true);
  else if (has_head) WMZ_FUSED(true, false);
  else WMZ_FUSED(false, true);
#undef WMZ_FUSED
  WMZ_LAUNCH_CHECK("wmz_layer_fused_fwd");
  return WMZ_OK;
}
