// Local windowed 3D attention, backward.  Replaces autograd through (and the checkpoint re-run of)
// Local3dAttention.local_attention (vq-video-diffusion/local_3d_attention.py:78-99, :110-111).
//
// The window relation is symmetric (|dh|<=eH etc. both ways), so both gradients are computed in GATHER form, no atomics,
// bitwise reproducible, by ONE kernel template in two roles:
//   MODE 0 (owner = queries)  dQ_i = scale * sum_j dS_ij K_j      and delta_i = rowsum(dO_i * O_i) (written for MODE 1)
//   MODE 1 (owner = keys)     dK_j = scale * sum_i dS_ij Q_i ,  dV_j = sum_i P_ij dO_i
// with P_ij = exp(scale*s_ij - lse_i), dS_ij = P_ij (dO_i.V_j - delta_i).  "Owner" tiles (16 positions) live in
// registers as MFMA B operands (X1 = Q|K, X2 = dO|V); "visitor" tiles of the neighbouring planes are staged in LDS
// (Y1 = K|Q, Y2 = V|dO) exactly like the forward's K/V slabs:
//   S^T [32 visitors x 16 owners] = Y1 . X1^T          dP^T = Y2 . X2^T          (row fragments, MFMA 16x16x32)
//   acc1^T[dh x 16 owners] += Y1^T . dS^T               (MODE 1 also: acc2^T += Y2^T . P^T)   (transposed LDS reads)
#include "attn_common.h"
#include <stdlib.h>

namespace {

constexpr int NWAVES = 4;
constexpr int NTHREADS = NWAVES * 64;

struct BwdPtrs {
  const void *x1, *x2;   // owner rows:   MODE0: Q, dO      MODE1: K, V
  const void *y1, *y2;   // visitor rows: MODE0: K, V       MODE1: Q, dO
  const void* o;         // MODE0 only: forward output (for delta)
  const float* lse;      // [N, heads]
  float* delta;          // [N, heads]   MODE0 writes, MODE1 reads
  void *g1, *g2;         // MODE0: dQ, -      MODE1: dK, dV
  long ldx1, ldx2, ldy1, ldy2, ldo, ldg1, ldg2;
};

template <typename T, int DH, int OPW, int KC, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void attn_bwd_kernel(BwdPtrs P, AttnGeom G) {
  constexpr int ROWB = DH * (int)sizeof(T);
  constexpr int IMG = KC * 16 * ROWB;
  constexpr int KS = DH / 32;
  constexpr int MT = DH / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Y1s = smem;
  char* Y2s = smem + IMG;
  int* coords = reinterpret_cast<int*>(smem + 2 * IMG);
  TileInfo* tinfo = reinterpret_cast<TileInfo*>(coords + G.tiles * 16);
  float* vlse = reinterpret_cast<float*>(tinfo + G.tiles);    // MODE 1: per-visitor lse*log2e and delta of the slab
  float* vdel = vlse + KC * 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int s = lid % G.S; lid /= G.S;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  attn_build_tables(coords, tinfo, G.HW, G.W, G.tiles, tid, NTHREADS);
  __syncthreads();

  const int HW = G.HW, dh = G.dh;
  const long plane_o = ((long)b * G.S + s) * HW;
  const float L2E = 1.4426950408889634f;
  const float c2 = G.scale * L2E;
  const unsigned lim = ((unsigned)(2 * G.eH) << 16) | (unsigned)(2 * G.eW);
  const T* X1 = reinterpret_cast<const T*>(P.x1);
  const T* X2 = reinterpret_cast<const T*>(P.x2);
  const T* Y1 = reinterpret_cast<const T*>(P.y1);
  const T* Y2 = reinterpret_cast<const T*>(P.y2);

  Frag8<T> x1f[OPW][KS], x2f[OPW][KS];
  f32x4 acc1[OPW][MT];
  f32x4 acc2[MODE == 1 ? OPW : 1][MODE == 1 ? MT : 1];
  float own_lse[OPW], own_del[OPW];      // MODE 0: lse*log2e and delta of the lane's query
  unsigned cmin[OPW];
  int need_lo[OPW], need_hi[OPW], owlo[OPW], owhi[OPW];
  bool active[OPW];
  const int ot0 = (og * NWAVES + wave) * OPW;
#pragma unroll
  for (int oi = 0; oi < OPW; ++oi) {
    const int ot = ot0 + oi;
    active[oi] = ot < G.tiles;
    const int otc = active[oi] ? ot : G.tiles - 1;
    const int po = otc * 16 + li;
    cmin[oi] = win_cmin(coords[po], G.eH, G.eW);
    const TileInfo ti = tinfo[otc];
    const int hlo = __builtin_amdgcn_readfirstlane(ti.hlo), hhi = __builtin_amdgcn_readfirstlane(ti.hhi);
    owlo[oi] = __builtin_amdgcn_readfirstlane(ti.wlo) - G.eW;
    owhi[oi] = __builtin_amdgcn_readfirstlane(ti.whi) + G.eW;
    need_lo[oi] = (max(hlo - G.eH, 0) * G.W) >> 4;
    need_hi[oi] = (min(hhi + G.eH, G.H - 1) * G.W + G.W - 1) >> 4;
    const bool ok = active[oi] && po < HW;
    const T* r1 = X1 + (plane_o + po) * P.ldx1 + (long)head * dh;
    const T* r2 = X2 + (plane_o + po) * P.ldx2 + (long)head * dh;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_zero(x1f[oi][ks]);
      frag_zero(x2f[oi][ks]);
      if (ok && ks * 32 + g * 8 < dh) {
        frag_load(x1f[oi][ks], r1 + ks * 32 + g * 8);
        frag_load(x2f[oi][ks], r2 + ks * 32 + g * 8);
        if constexpr (MODE == 0) {
          // delta = rowsum(dO * O): this lane's 8-channel slice, reduced over the 4 lane groups below
          Frag8<T> of;
          frag_load(of, reinterpret_cast<const T*>(P.o) + (plane_o + po) * P.ldo + (long)head * dh + ks * 32 + g * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float a, c;
            if constexpr (sizeof(T) == 2) {
              a = bf16_bits_to_f32((unsigned short)x2f[oi][ks].v[j]);
              c = bf16_bits_to_f32((unsigned short)of.v[j]);
            } else {
              a = x2f[oi][ks].v[j];
              c = of.v[j];
            }
            dsum = fmaf(a, c, dsum);
          }
        }
      }
    }
    if constexpr (MODE == 0) {
      dsum = wave_xor_add(dsum, 16);
      dsum = wave_xor_add(dsum, 32);
      own_del[oi] = dsum;
      own_lse[oi] = ok ? P.lse[(plane_o + po) * G.heads + head] * L2E : 0.f;
      if (ok && g == 0) P.delta[(plane_o + po) * G.heads + head] = dsum;
    } else {
      own_del[oi] = 0.f;
      own_lse[oi] = 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc1[oi][mt] = (f32x4)(0.f);
    if constexpr (MODE == 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc2[oi][mt] = (f32x4)(0.f);
    }
  }

  const int wo_first = og * NWAVES * OPW;
  const int wo_last = min(G.tiles - 1, wo_first + NWAVES * OPW - 1);
  const int wg_hlo = __builtin_amdgcn_readfirstlane(tinfo[wo_first].hlo);
  const int wg_hhi = __builtin_amdgcn_readfirstlane(tinfo[wo_last].hhi);
  const int t_lo = (max(wg_hlo - G.eH, 0) * G.W) >> 4;
  const int t_hi = (min(wg_hhi + G.eH, G.H - 1) * G.W + G.W - 1) >> 4;

  for (int ds = -G.eS; ds <= G.eS; ++ds) {
    const int sv = s + ds;
    if (sv < 0 || sv >= G.S) continue;
    const long plane_v = ((long)b * G.S + sv) * HW;
    const T* y1p = Y1 + plane_v * P.ldy1 + (long)head * dh;
    const T* y2p = Y2 + plane_v * P.ldy2 + (long)head * dh;
    for (int c0 = t_lo; c0 <= t_hi; c0 += KC) {
      const int ntl = min(KC, t_hi - c0 + 1);
      {
        constexpr int NREG = KC * 16 * (ROWB / 16) / NTHREADS;
        i32x4 r1[NREG], r2[NREG];
        attn_stage_load<T, DH, false, KC, NTHREADS>(r1, y1p, P.ldy1, c0, ntl, HW, dh, tid);
        attn_stage_load<T, DH, false, KC, NTHREADS>(r2, y2p, P.ldy2, c0, ntl, HW, dh, tid);
        float vl = 0.f, vd = 0.f;
        if constexpr (MODE == 1) {
          if (tid < KC * 16) {
            const int p = c0 * 16 + tid;
            if (tid < ntl * 16 && p < HW) {
              vl = P.lse[(plane_v + p) * G.heads + head] * L2E;
              vd = P.delta[(plane_v + p) * G.heads + head];
            }
          }
        }
        __syncthreads();
        attn_stage_store<T, DH, false, KC, NTHREADS>(Y1s, r1, ntl, tid);
        attn_stage_store<T, DH, false, KC, NTHREADS>(Y2s, r2, ntl, tid);
        if constexpr (MODE == 1) {
          if (tid < KC * 16) { vlse[tid] = vl; vdel[tid] = vd; }
        }
        __syncthreads();
      }
      const int c_hi = c0 + ntl - 1;
#pragma unroll
      for (int oi = 0; oi < OPW; ++oi) {
        if (!active[oi]) continue;
        const int lo = max(c0, need_lo[oi]), hi = min(c_hi, need_hi[oi]);
        for (int t0 = lo; t0 <= hi; t0 += 2) {
          const bool has1 = t0 + 1 <= hi;
          const int t1 = has1 ? t0 + 1 : t0;
          const TileInfo k0 = tinfo[t0], k1 = tinfo[t1];
          const bool n0 = __builtin_amdgcn_readfirstlane(k0.wlo) <= owhi[oi] &&
                          __builtin_amdgcn_readfirstlane(k0.whi) >= owlo[oi];
          const bool n1 = has1 && __builtin_amdgcn_readfirstlane(k1.wlo) <= owhi[oi] &&
                          __builtin_amdgcn_readfirstlane(k1.whi) >= owlo[oi];
          if (!n0 && !n1) continue;
          const int r0 = (t0 - c0) * 16, r1 = (t1 - c0) * 16;

          f32x4 s0 = (f32x4)(0.f), s1 = (f32x4)(0.f), d0 = (f32x4)(0.f), d1 = (f32x4)(0.f);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            Frag8<T> a0, a1, b0, b1;
            lds_row_frag<T, DH, false>(a0, Y1s, r0 + li, ks * 32 + g * 8);
            lds_row_frag<T, DH, false>(a1, Y1s, r1 + li, ks * 32 + g * 8);
            lds_row_frag<T, DH, false>(b0, Y2s, r0 + li, ks * 32 + g * 8);
            lds_row_frag<T, DH, false>(b1, Y2s, r1 + li, ks * 32 + g * 8);
            mma16(s0, a0, x1f[oi][ks]);
            mma16(s1, a1, x1f[oi][ks]);
            mma16(d0, b0, x2f[oi][ks]);
            mma16(d1, b1, x2f[oi][ks]);
          }
          const i32x4 kc0 = *reinterpret_cast<const i32x4*>(coords + t0 * 16 + 4 * g);
          const i32x4 kc1 = *reinterpret_cast<const i32x4*>(coords + t1 * 16 + 4 * g);
          f32x4 l0, l1, e0, e1;
          if constexpr (MODE == 1) {
            l0 = *reinterpret_cast<const f32x4*>(vlse + r0 + 4 * g);
            l1 = *reinterpret_cast<const f32x4*>(vlse + r1 + 4 * g);
            e0 = *reinterpret_cast<const f32x4*>(vdel + r0 + 4 * g);
            e1 = *reinterpret_cast<const f32x4*>(vdel + r1 + 4 * g);
          }
          float pv[8], dsv[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok0 = win_inside(win_delta(kc0[r], cmin[oi]), lim);
            const bool ok1 = has1 && win_inside(win_delta(kc1[r], cmin[oi]), lim);
            const float ls0 = MODE == 1 ? l0[r] : own_lse[oi], ls1 = MODE == 1 ? l1[r] : own_lse[oi];
            const float de0 = MODE == 1 ? e0[r] : own_del[oi], de1 = MODE == 1 ? e1[r] : own_del[oi];
            const float p0 = ok0 ? exp2f(fmaf(s0[r], c2, -ls0)) : 0.f;
            const float p1 = ok1 ? exp2f(fmaf(s1[r], c2, -ls1)) : 0.f;
            pv[r] = p0;
            pv[4 + r] = p1;
            dsv[r] = p0 * (d0[r] - de0) * G.scale;
            dsv[4 + r] = p1 * (d1[r] - de1) * G.scale;
          }
          Frag8<T> dsf;
          frag_from_f32<T>(dsf, dsv);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            Frag8<T> yf;
            lds_col_frag<T, DH, false>(yf, Y1s, r0, r1, g, li, mt * 16);
            mma16(acc1[oi][mt], yf, dsf);
          }
          if constexpr (MODE == 1) {
            Frag8<T> pf;
            frag_from_f32<T>(pf, pv);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              Frag8<T> yf;
              lds_col_frag<T, DH, false>(yf, Y2s, r0, r1, g, li, mt * 16);
              mma16(acc2[oi][mt], yf, pf);
            }
          }
        }
      }
    }
  }

  auto store = [&](void* dst, long ld, const f32x4 (&acc)[MT], int po) {
    T* row = reinterpret_cast<T*>(dst) + (plane_o + po) * ld + (long)head * dh;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ch = mt * 16 + 4 * g;
      if (ch < dh) {
        if constexpr (sizeof(T) == 2) {
          s16x4 pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(acc[mt][r]);
          *reinterpret_cast<s16x4*>(row + ch) = pk;
        } else {
          *reinterpret_cast<f32x4*>(row + ch) = acc[mt];
        }
      }
    }
  };
#pragma unroll
  for (int oi = 0; oi < OPW; ++oi) {
    if (!active[oi]) continue;
    const int po = (ot0 + oi) * 16 + li;
    if (po >= HW) continue;
    store(P.g1, P.ldg1, acc1[oi], po);
    if constexpr (MODE == 1) store(P.g2, P.ldg2, acc2[oi], po);
  }
}

template <typename T, int DH, int OPW, int KC, int MODE>
int launch(const BwdPtrs& P, AttnGeom G, hipStream_t st) {
  G.qgroups = wmz_cdiv(G.tiles, NWAVES * OPW);
  const long nwg = (long)G.B * G.heads * G.S * G.qgroups;
  const size_t smem = 2 * (size_t)KC * 16 * DH * sizeof(T) + (size_t)G.tiles * 16 * 4 + (size_t)G.tiles * sizeof(TileInfo) +
                      2 * KC * 16 * sizeof(float);
  if (smem > 160 * 1024) { wmz_set_error("wmz_local3d_attn_bwd: plane too large for the LDS tables (H*W=%d)", G.HW); return WMZ_ERR_UNSUPPORTED; }
  auto kern = attn_bwd_kernel<T, DH, OPW, KC, MODE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NTHREADS), smem, st, P, G);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_bwd");
  return WMZ_OK;
}

template <typename T, int DH, int KC>
int launch_both(const BwdPtrs& PQ, const BwdPtrs& PK, const AttnGeom& G, hipStream_t st) {
  constexpr int OPW0 = sizeof(T) == 2 ? 2 : 1;
  int rc = launch<T, DH, OPW0, KC, 0>(PQ, G, st);
  if (rc != WMZ_OK) return rc;
  return launch<T, DH, 1, KC, 1>(PK, G, st);
}

}  // namespace

// fast path for 16-wide planes (attn_bwd_row16.hip)
int wmz_attn_bwd_row16_dispatch(const void* q, const void* k, const void* v, const void* out, const float* lse,
                                const void* dout, void* dq, void* dk, void* dv, float* delta, const AttnGeom& G, long lddo,
                                long lddq, long lddk, long lddv, hipStream_t st);

extern "C" int wmz_local3d_attn_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                                    const void* dout, void* dq, void* dk, void* dv, float* delta_ws, int B, int S, int H,
                                    int W, int heads, int dh, int eS, int eH, int eW, long ldq, long ldk, long ldv,
                                    long ldo, long lddo, long lddq, long lddk, long lddv, int dtype, void* stream) {
  WMZ_REQUIRE(q && k && v && out && lse && dout && dq && dk && dv && delta_ws, "wmz_local3d_attn_bwd: null tensor");
  WMZ_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && heads > 0 && dh > 0, "wmz_local3d_attn_bwd: bad shape");
  WMZ_REQUIRE(eS >= 0 && eH >= 0 && eW >= 0, "wmz_local3d_attn_bwd: negative extent");
  WMZ_REQUIRE(H <= 16384 && W <= 16384, "wmz_local3d_attn_bwd: H, W must be <= 16384");
  WMZ_REQUIRE(dh % 8 == 0, "wmz_local3d_attn_bwd: dim_head must be a multiple of 8 (got %d)", dh);
  WMZ_REQUIRE((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) % 8 == 0, "wmz_local3d_attn_bwd: row strides must be multiples of 8");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_local3d_attn_bwd: bad dtype %d", dtype);
  if (dh > 128) { wmz_set_error("wmz_local3d_attn_bwd: dim_head %d > 128 not built", dh); return WMZ_ERR_UNSUPPORTED; }
  AttnGeom G;
  G.B = B; G.S = S; G.H = H; G.W = W; G.heads = heads; G.dh = dh; G.eS = eS; G.eH = eH; G.eW = eW;
  G.ldq = ldq; G.ldk = ldk; G.ldv = ldv; G.ldo = ldo;
  G.HW = H * W; G.tiles = (G.HW + 15) / 16; G.qgroups = 0;
  G.scale = 1.0f / sqrtf((float)dh);
  G.dbg = 0; G.qs0 = 0; G.Sq = S; G.variant = 0; G.w8 = 0;
  BwdPtrs PQ, PK;
  PQ.x1 = q; PQ.x2 = dout; PQ.y1 = k; PQ.y2 = v; PQ.o = out; PQ.lse = lse; PQ.delta = delta_ws; PQ.g1 = dq; PQ.g2 = nullptr;
  PQ.ldx1 = ldq; PQ.ldx2 = lddo; PQ.ldy1 = ldk; PQ.ldy2 = ldv; PQ.ldo = ldo; PQ.ldg1 = lddq; PQ.ldg2 = 0;
  PK.x1 = k; PK.x2 = v; PK.y1 = q; PK.y2 = dout; PK.o = nullptr; PK.lse = lse; PK.delta = delta_ws; PK.g1 = dk; PK.g2 = dv;
  PK.ldx1 = ldk; PK.ldx2 = ldv; PK.ldy1 = ldq; PK.ldy2 = lddo; PK.ldo = 0; PK.ldg1 = lddk; PK.ldg2 = lddv;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16 && W == 16 && (dh == 32 || dh == 64 || dh == 128) )
    return wmz_attn_bwd_row16_dispatch(q, k, v, out, lse, dout, dq, dk, dv, delta_ws, G, lddo, lddq, lddk, lddv, st);
  if (dtype == WMZ_BF16 && W == 8 && (H & 1) == 0 && (dh == 32 || dh == 64 || dh == 128)) {
    AttnGeom G8 = G;                       // an 8-wide plane as H / 2 tile rows of 16 (attn_fwd_row16.hip)
    G8.w8 = 1; G8.H = H / 2; G8.W = 16;
    return wmz_attn_bwd_row16_dispatch(q, k, v, out, lse, dout, dq, dk, dv, delta_ws, G8, lddo, lddq, lddk, lddv, st);
  }
  const int DHp = dh <= 32 ? 32 : (dh <= 64 ? 64 : 128);
  if (dtype == WMZ_BF16) {
    if (DHp == 32) return launch_both<bf16_t, 32, 8>(PQ, PK, G, st);
    if (DHp == 64) return launch_both<bf16_t, 64, 8>(PQ, PK, G, st);
    return launch_both<bf16_t, 128, 8>(PQ, PK, G, st);
  }
  if (DHp == 32) return launch_both<float, 32, 8>(PQ, PK, G, st);
  if (DHp == 64) return launch_both<float, 64, 8>(PQ, PK, G, st);
  return launch_both<float, 128, 4>(PQ, PK, G, st);
}
