// Local windowed 3D attention, forward (replaces Local3dAttention.local_attention,
// vq-video-diffusion/local_3d_attention.py:78-99, without materialising the unfolded K/V).
//
// Work split: a workgroup (bf16: 16 waves = four per SIMD, one 16-query tile each; the per-tile dependency chain
// S -> max -> exp -> PV is latency-bound, so occupancy is what hides it) owns NWAVES*QPW consecutive 16-query tiles of one (b, head, s) plane -- a
// whole 16x16 plane at the BASELINE shapes, so every K/V row is staged once per neighbouring plane (7x) instead of once
// per half plane with its halo (12x).  For every in-range key plane s+ds, slabs of KC key tiles of K and V are brought
// into a DOUBLE-BUFFERED pair of swizzled LDS images by LDS-DMA (global_load_lds, swizzle applied on the source address):
// the slab after the one being processed is always in flight.  Each wave walks the key tiles its query tiles can see,
// two at a time:
//   S^T[32 keys x 16 queries] = K . Q^T          (MFMA 16x16x32, A = K rows from LDS, B = Q rows in registers)
//   window mask (packed u16 coordinate test), online softmax (row max via two wave shuffles)
//   O^T[dh x 16 queries]    += V^T . P^T          (A = V^T by ds_read_b64_tr_b16, B = P straight from S^T's
//                                                  accumulator layout: no cross-lane movement)
// The reference's zero-padded, -1e9-masked slots have probability exactly 0, so they are simply not visited.
#include "attn_common.h"
#include "wmz_debug.h"
#include <stdlib.h>

namespace {


template <typename T, int DH, int QPW, int KC, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void attn_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                               const T* __restrict__ V, T* __restrict__ O,
                                                               float* __restrict__ LSE, float* __restrict__ DBG,
                                                               AttnGeom G) {
  constexpr int NTHREADS = NWAVES * 64;
  constexpr int ROWB = DH * (int)sizeof(T);
  constexpr int IMG = KC * 16 * ROWB;
  constexpr int KS = DH / 32;   // k-steps of QK^T
  constexpr int MT = DH / 16;   // 16-row blocks of O^T
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* coords = reinterpret_cast<int*>(smem + 4 * IMG);      // [buf0: K | V][buf1: K | V][tables]
  TileInfo* tinfo = reinterpret_cast<TileInfo*>(coords + G.tiles * 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qg = lid % G.qgroups; lid /= G.qgroups;
  const int sq = lid % G.Sq; lid /= G.Sq;
  const int s = G.qs0 + sq;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  attn_build_tables(coords, tinfo, G.HW, G.W, G.tiles, tid, NTHREADS);
  __syncthreads();

  const int HW = G.HW, dh = G.dh;
  const long plane_q = ((long)b * G.S + s) * HW;
  const long plane_o = ((long)b * G.Sq + sq) * HW;
  const float c2 = G.scale * 1.4426950408889634f;   // logits -> log2 domain
  const unsigned lim = ((unsigned)(2 * G.eH) << 16) | (unsigned)(2 * G.eW);
  const int KHW = (2 * G.eH + 1) * (2 * G.eW + 1), KW = 2 * G.eW + 1;

  // ---- per query tile state
  Frag8<T> qf[QPW][KS];
  f32x4 o[QPW][MT];
  float m_run[QPW], l_run[QPW];
  unsigned cmin[QPW];
  int need_lo[QPW], need_hi[QPW], qwlo[QPW], qwhi[QPW];
  bool active[QPW];
  const int qt0 = (qg * NWAVES + wave) * QPW;
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    const int qt = qt0 + qi;
    active[qi] = qt < G.tiles;
    const int qtc = active[qi] ? qt : G.tiles - 1;
    const int pq = qtc * 16 + li;
    cmin[qi] = win_cmin(coords[pq], G.eH, G.eW);
    const TileInfo ti = tinfo[qtc];
    const int hlo = __builtin_amdgcn_readfirstlane(ti.hlo), hhi = __builtin_amdgcn_readfirstlane(ti.hhi);
    qwlo[qi] = __builtin_amdgcn_readfirstlane(ti.wlo) - G.eW;
    qwhi[qi] = __builtin_amdgcn_readfirstlane(ti.whi) + G.eW;
    need_lo[qi] = (max(hlo - G.eH, 0) * G.W) >> 4;
    need_hi[qi] = (min(hhi + G.eH, G.H - 1) * G.W + G.W - 1) >> 4;
    const bool qok = active[qi] && pq < HW;
    const T* qrow = Q + (plane_q + pq) * G.ldq + (long)head * dh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_zero(qf[qi][ks]);
      if (qok && ks * 32 + g * 8 < dh) frag_load(qf[qi][ks], qrow + ks * 32 + g * 8);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) o[qi][mt] = (f32x4)(0.f);
    m_run[qi] = -1e30f;
    l_run[qi] = 0.f;
  }

  // ---- key tile range of the whole workgroup (wave-uniform)
  const int wq_first = qg * NWAVES * QPW;
  const int wq_last = min(G.tiles - 1, wq_first + NWAVES * QPW - 1);
  const int wg_hlo = __builtin_amdgcn_readfirstlane(tinfo[wq_first].hlo);
  const int wg_hhi = __builtin_amdgcn_readfirstlane(tinfo[wq_last].hhi);
  const int t_lo = (max(wg_hlo - G.eH, 0) * G.W) >> 4;
  const int t_hi = (min(wg_hhi + G.eH, G.H - 1) * G.W + G.W - 1) >> 4;

  // slab schedule: planes sk_lo..sk_hi x nch slabs of KC tiles; slab j+1 is in flight while slab j is processed
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int nch = (t_hi - t_lo + KC) / KC;
  const int nslab = (sk_hi - sk_lo + 1) * nch;
  auto issue = [&](int j) {
    if (G.dbg & 2) return;
    const int pl = j / nch, c0 = t_lo + (j - pl * nch) * KC;
    const long plane_k = ((long)b * G.S + (sk_lo + pl)) * HW;
    const int ntl = min(KC, t_hi - c0 + 1);
    char* buf = smem + (j & 1) * 2 * IMG;
    attn_stage_dma<T, DH, false, KC, NWAVES>(buf, K + plane_k * G.ldk + (long)head * dh, G.ldk, c0, ntl, HW, dh, wave, lane);
    attn_stage_dma<T, DH, true, KC, NWAVES>(buf + IMG, V + plane_k * G.ldv + (long)head * dh, G.ldv, c0, ntl, HW, dh, wave, lane);
  };
  issue(0);
  for (int j = 0; j < nslab; ++j) {
    const int pl = j / nch, c0 = t_lo + (j - pl * nch) * KC;
    const int ds = sk_lo + pl - s;
    const int ntl = min(KC, t_hi - c0 + 1);
    const char* Ks = smem + (j & 1) * 2 * IMG;
    const char* Vs = Ks + IMG;
    // this wave's pieces of slab j have landed; the barrier makes every wave's pieces visible and retires slab j-1,
    // whose buffer is refilled right away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + 1 < nslab) issue(j + 1);
    {
      const int c_hi = c0 + ntl - 1;
#pragma unroll
      for (int qi = 0; qi < QPW; ++qi) {
        if (!active[qi] || (G.dbg & 1)) continue;
        const int lo = max(c0, need_lo[qi]), hi = min(c_hi, need_hi[qi]);
        for (int t0 = lo; t0 <= hi; t0 += 2) {
          const bool has1 = t0 + 1 <= hi;
          const int t1 = has1 ? t0 + 1 : t0;
          // column-range test (only bites when a tile is narrower than a row: W > 16)
          const TileInfo k0 = tinfo[t0], k1 = tinfo[t1];
          const bool n0 = __builtin_amdgcn_readfirstlane(k0.wlo) <= qwhi[qi] &&
                          __builtin_amdgcn_readfirstlane(k0.whi) >= qwlo[qi];
          const bool n1 = has1 && __builtin_amdgcn_readfirstlane(k1.wlo) <= qwhi[qi] &&
                          __builtin_amdgcn_readfirstlane(k1.whi) >= qwlo[qi];
          if (!n0 && !n1) continue;
          const int r0 = (t0 - c0) * 16, r1 = (t1 - c0) * 16;

          // ---- S^T = K Q^T for the two key tiles
          f32x4 s0 = (f32x4)(0.f), s1 = (f32x4)(0.f);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            Frag8<T> ka, kb;
            lds_row_frag<T, DH, false>(ka, Ks, r0 + li, ks * 32 + g * 8);
            lds_row_frag<T, DH, false>(kb, Ks, r1 + li, ks * 32 + g * 8);
            mma16(s0, ka, qf[qi][ks]);
            mma16(s1, kb, qf[qi][ks]);
          }
          // ---- window mask.  lane holds keys 4g..4g+3 of each tile for query li
          const i32x4 kc0 = *reinterpret_cast<const i32x4*>(coords + t0 * 16 + 4 * g);
          const i32x4 kc1 = *reinterpret_cast<const i32x4*>(coords + t1 * 16 + 4 * g);
          float sv[8];
          float mx = -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned d0 = win_delta(kc0[r], cmin[qi]);
            const unsigned d1 = win_delta(kc1[r], cmin[qi]);
            const bool ok0 = win_inside(d0, lim);
            const bool ok1 = has1 && win_inside(d1, lim);
            if (DBG != nullptr && qt0 * 16 + qi * 16 + li < HW) {
              const long qrow = ((plane_q + qt0 * 16 + qi * 16 + li) * G.heads + head) * (long)((2 * G.eS + 1) * KHW);
              if (ok0) DBG[qrow + (ds + G.eS) * KHW + (int)(d0 >> 16) * KW + (int)(d0 & 0xFFFF)] = s0[r] * G.scale;
              if (ok1) DBG[qrow + (ds + G.eS) * KHW + (int)(d1 >> 16) * KW + (int)(d1 & 0xFFFF)] = s1[r] * G.scale;
            }
            sv[r] = ok0 ? s0[r] : -INFINITY;
            sv[4 + r] = ok1 ? s1[r] : -INFINITY;
            mx = fmaxf(mx, fmaxf(sv[r], sv[4 + r]));
          }
          // ---- online softmax; all four lane groups of a query agree on the running max
          mx = wave_xor_max(mx, 16);
          mx = wave_xor_max(mx, 32);
          const float m_new = fmaxf(m_run[qi], mx);
          const float alpha = exp2f((m_run[qi] - m_new) * c2);
          m_run[qi] = m_new;
          const float mb = -m_new * c2;
          float p[8];
          float psum = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            p[r] = exp2f(fmaf(sv[r], c2, mb));
            psum += p[r];
          }
          l_run[qi] = l_run[qi] * alpha + psum;
          Frag8<T> pf;
          frag_from_f32<T>(pf, p);
          // ---- O^T = alpha O^T + V^T P^T
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            Frag8<T> vf;
            lds_col_frag<T, DH, true>(vf, Vs, r0, r1, g, li, mt * 16);
            o[qi][mt] *= alpha;
            mma16(o[qi][mt], vf, pf);
          }
        }
      }
    }
  }

  // ---- normalise and store.  lane holds query li, channels 16*mt + 4g + (0..3)
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    if (!active[qi]) continue;
    const int pq = (qt0 + qi) * 16 + li;
    float l = l_run[qi];
    l = wave_xor_add(l, 16);
    l = wave_xor_add(l, 32);
    if (pq >= HW) continue;
    const float inv = 1.f / l;
    T* orow = O + (plane_o + pq) * G.ldo + (long)head * dh;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ch = mt * 16 + 4 * g;
      if (ch < dh) {
        if constexpr (sizeof(T) == 2) {
          s16x4 pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(o[qi][mt][r] * inv);
          *reinterpret_cast<s16x4*>(orow + ch) = pk;
        } else {
          *reinterpret_cast<f32x4*>(orow + ch) = o[qi][mt] * inv;
        }
      }
    }
    if (LSE != nullptr && g == 0) LSE[(plane_o + pq) * G.heads + head] = m_run[qi] * G.scale + logf(l);
  }
}

template <typename T, int DH, int QPW, int KC, int NWAVES>
int launch(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg, AttnGeom G,
           hipStream_t st) {
  G.qgroups = wmz_cdiv(G.tiles, NWAVES * QPW);
  const long nwg = (long)G.B * G.heads * G.Sq * G.qgroups;
  const size_t smem = 4 * (size_t)KC * 16 * DH * sizeof(T) + (size_t)G.tiles * 16 * 4 + (size_t)G.tiles * sizeof(TileInfo);
  if (smem > 160 * 1024) { wmz_set_error("wmz_local3d_attn_fwd: plane too large for the LDS tables (H*W=%d)", G.HW); return WMZ_ERR_UNSUPPORTED; }
  constexpr int NTHREADS = NWAVES * 64;
  auto kern = attn_fwd_kernel<T, DH, QPW, KC, NWAVES>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NTHREADS), smem, st, (const T*)q, (const T*)k, (const T*)v,
                     (T*)out, lse, dbg, G);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_fwd");
  return WMZ_OK;
}

}  // namespace

// fast path for 16-wide planes: one query row per wave (attn_fwd_row16.hip)
int wmz_attn_fwd_row16_dispatch(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg,
                                const AttnGeom& G, hipStream_t st);
// ... and its IEEE-half instantiation (attn_fwd_row16_f16.hip: the precise fused mode, dtype WMZ_F16)
int wmz_attn_fwd_row16_dispatch_f16(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg,
                                    const AttnGeom& G, hipStream_t st);

// development knobs (wmz_debug_attn_knobs): ablation switches and kernel-variant selector, 0 / 0 in production
static int g_attn_dbg = 0, g_attn_variant = 0;
extern "C" int wmz_debug_attn_knobs(int dbg, int variant) { g_attn_dbg = dbg; g_attn_variant = variant; return WMZ_OK; }

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse, float* logits_dbg, int B, int S,
                         int H, int W, int heads, int dh, int eS, int eH, int eW, long ldq, long ldk, long ldv, long ldo,
                         int q_plane0, int q_planes, int dtype, void* stream, bool general);

extern "C" int wmz_local3d_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                                    float* logits_dbg, int B, int S, int H, int W, int heads, int dh, int eS, int eH,
                                    int eW, long ldq, long ldk, long ldv, long ldo, int dtype, void* stream) {
  return attn_fwd_impl(q, k, v, out, lse, logits_dbg, B, S, H, W, heads, dh, eS, eH, eW, ldq, ldk, ldv, ldo, 0, S, dtype, stream, false);
}

// Same contract, always on the general kernel (any W, fp32 or bf16): the parity tests compare the fast path with it.
extern "C" int wmz_local3d_attn_fwd_general(const void* q, const void* k, const void* v, void* out, float* lse,
                                            float* logits_dbg, int B, int S, int H, int W, int heads, int dh, int eS, int eH,
                                            int eW, long ldq, long ldk, long ldv, long ldo, int dtype, void* stream) {
  return attn_fwd_impl(q, k, v, out, lse, logits_dbg, B, S, H, W, heads, dh, eS, eH, eW, ldq, ldk, ldv, ldo, 0, S, dtype, stream, true);
}

extern "C" int wmz_local3d_attn_fwd_planes(const void* q, const void* k, const void* v, void* out, float* lse, int B, int S,
                                           int H, int W, int heads, int dh, int eS, int eH, int eW, long ldq, long ldk,
                                           long ldv, long ldo, int q_plane0, int q_planes, int dtype, void* stream) {
  return attn_fwd_impl(q, k, v, out, lse, nullptr, B, S, H, W, heads, dh, eS, eH, eW, ldq, ldk, ldv, ldo, q_plane0, q_planes,
                       dtype, stream, false);
}

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse, float* logits_dbg, int B, int S,
                         int H, int W, int heads, int dh, int eS, int eH, int eW, long ldq, long ldk, long ldv, long ldo,
                         int q_plane0, int q_planes, int dtype, void* stream, bool general) {
  WMZ_REQUIRE(q && k && v && out, "wmz_local3d_attn_fwd: null tensor");
  WMZ_REQUIRE(q_plane0 >= 0 && q_planes > 0 && q_plane0 + q_planes <= S, "wmz_local3d_attn_fwd: query planes [%d, %d) outside [0, %d)", q_plane0, q_plane0 + q_planes, S);
  WMZ_REQUIRE(logits_dbg == nullptr || q_planes == S, "wmz_local3d_attn_fwd: the logits probe needs the full grid");
  WMZ_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && heads > 0 && dh > 0, "wmz_local3d_attn_fwd: bad shape");
  WMZ_REQUIRE(eS >= 0 && eH >= 0 && eW >= 0, "wmz_local3d_attn_fwd: negative extent");
  WMZ_REQUIRE(H <= 16384 && W <= 16384, "wmz_local3d_attn_fwd: H, W must be <= 16384");
  WMZ_REQUIRE(dh % 8 == 0, "wmz_local3d_attn_fwd: dim_head must be a multiple of 8 (got %d)", dh);
  WMZ_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "wmz_local3d_attn_fwd: row strides must be multiples of 8");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16 || dtype == WMZ_F16, "wmz_local3d_attn_fwd: bad dtype %d", dtype);
  if (dh > 128) { wmz_set_error("wmz_local3d_attn_fwd: dim_head %d > 128 not built", dh); return WMZ_ERR_UNSUPPORTED; }
  AttnGeom G;
  G.B = B; G.S = S; G.H = H; G.W = W; G.heads = heads; G.dh = dh; G.eS = eS; G.eH = eH; G.eW = eW;
  G.ldq = ldq; G.ldk = ldk; G.ldv = ldv; G.ldo = ldo;
  G.HW = H * W; G.tiles = (G.HW + 15) / 16; G.qgroups = 0;
  G.qs0 = q_plane0; G.Sq = q_planes;
  G.scale = 1.0f / sqrtf((float)dh);
  G.dbg = g_attn_dbg;
  G.variant = g_attn_variant;
  G.w8 = 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype != WMZ_F32 && (dh == 32 || dh == 64 || dh == 128) && !general) {
    const auto row16 = dtype == WMZ_F16 ? wmz_attn_fwd_row16_dispatch_f16 : wmz_attn_fwd_row16_dispatch;
    if (W == 16) return row16(q, k, v, out, lse, logits_dbg, G, st);
    if (W == 8 && (H & 1) == 0) {
      // an 8-wide plane with an even number of rows is a 16-wide plane of H / 2 tile rows in memory (the reference's own 8x8
      // latents, main.py:394): the row kernel with its window test in tile coordinates
      AttnGeom G8 = G;
      G8.w8 = 1; G8.H = H / 2; G8.W = 16;
      return row16(q, k, v, out, lse, logits_dbg, G8, st);
    }
  }
  if (dtype == WMZ_F16) {
    wmz_set_error("wmz_local3d_attn_fwd: half operands are built for the row kernel's shapes (dim_head 32 / 64 / 128, planes 16 wide or 8 wide with an even number of rows)");
    return WMZ_ERR_UNSUPPORTED;
  }
  const int DHp = dh <= 32 ? 32 : (dh <= 64 ? 64 : 128);
  if (dtype == WMZ_BF16) {
    if (DHp == 32) return launch<bf16_t, 32, 1, 8, 16>(q, k, v, out, lse, logits_dbg, G, st);
    if (DHp == 64) return launch<bf16_t, 64, 1, 8, 16>(q, k, v, out, lse, logits_dbg, G, st);
    return launch<bf16_t, 128, 1, 8, 16>(q, k, v, out, lse, logits_dbg, G, st);
  }
  if (DHp == 32) return launch<float, 32, 1, 8, 8>(q, k, v, out, lse, logits_dbg, G, st);
  if (DHp == 64) return launch<float, 64, 1, 8, 8>(q, k, v, out, lse, logits_dbg, G, st);
  return launch<float, 128, 1, 4, 8>(q, k, v, out, lse, logits_dbg, G, st);
}
