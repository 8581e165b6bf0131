// Local windowed 3D attention backward, fast path for 16-wide planes (bf16, dim_head in {32,64,128}).  Same math and the
// same gather-form / two-role structure as attn_bwd.hip (which stays the general / fp32 path), with the forward fast
// path's machinery (attn_fwd_row16.hip): key-row loop bounds + four column-window biases instead of per-element tests,
// padded LDS rows (address = lane base + row offset + immediate), LDS-DMA double-buffered slabs of 8 plane rows.
//   MODE 0 (owner = one query row per wave, 16 waves):  dQ = scale * sum_j dS_ij K_j ;  delta_i = rowsum(dO_i * O_i)
//   MODE 1 (owner = one key row per wave,    8 waves):  dK = scale * sum_i dS_ij Q_i ;  dV = sum_i P_ij dO_i
// P_ij = exp2(c2 * s_ij + bias - lse2_i), dS_ij = P_ij (dO_i . V_j - delta_i).  Owner rows are MFMA B operands in registers,
// the visiting rows' two tensors (K,V | Q,dO) are the slab images: row fragments for S^T and dP^T, transposed reads
// (ds_read_b64_tr_b16) for acc^T += Y^T . dS^T.
#include "attn_common.h"

namespace {

constexpr int KC = 8;

template <int DH> struct BImg {
  static constexpr int ROWP = DH * 2 + 32;             // both read kinds hit this image: +32 B keeps tr reads conflict-free
  static constexpr int IMG = KC * 16 * ROWP;
  static constexpr int BUF = 2 * IMG + 2 * KC * 16 * 4; // Y1 | Y2 | visitor lse2 | visitor delta
  static_assert(IMG % 1024 == 0, "image must be whole 1 KB DMA pieces");
};

template <int DH, int NW>
__device__ __forceinline__ void stage_img(char* dst, const bf16_t* plane, long ld, int row0, int last_row, int wave, int lane) {
  constexpr int ROWP = BImg<DH>::ROWP;
  constexpr int PIECES = BImg<DH>::IMG / 1024;
#pragma unroll
  for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
    const int piece = wave + NW * i;
    if (piece >= PIECES) break;
    const int off = piece * 1024 + lane * 16;
    const int r = off / ROWP;
    int c = (off - r * ROWP) >> 4;
    c = c < DH / 8 ? c : 0;
    const int rr = min(r, last_row);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(plane + (long)(row0 + rr) * ld + c * 8),
                                     (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
  }
}

struct RBwdPtrs {
  const bf16_t *x1, *x2, *y1, *y2, *o;
  const float* lse;
  float* delta;
  bf16_t *g1, *g2;
  long ldx1, ldx2, ldy1, ldy2, ldo, ldg1, ldg2;
};

template <int DH, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_bwd_row16_kernel(RBwdPtrs P, AttnGeom G) {
  using I = BImg<DH>;
  constexpr int KS = DH / 32, MT = DH / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int s = lid % G.S; lid /= G.S;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  const int HW = G.HW, H = G.H;
  const int h = og * NW + wave;
  const bool active = h < H;
  const long plane_o = ((long)b * G.S + s) * HW;
  const float L2E = 1.4426950408889634f;
  const float c2 = G.scale * L2E;

  // window bias of this lane's 4 visitor columns (w = 4g + r) against its owner column (w = li): symmetric in the roles
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int d = 4 * g + r - li; bias[r] = (d <= G.eW && -d <= G.eW) ? 0.f : -INFINITY; }

  Frag8<bf16_t> x1f[KS], x2f[KS];
  float own_lse = 0.f, own_del = 0.f;
  {
    const long row = plane_o + (active ? h : 0) * 16 + li;
    const bf16_t* r1 = P.x1 + row * P.ldx1 + (long)head * DH;
    const bf16_t* r2 = P.x2 + row * P.ldx2 + (long)head * DH;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_zero(x1f[ks]);
      frag_zero(x2f[ks]);
      if (active) {
        frag_load(x1f[ks], r1 + ks * 32 + g * 8);
        frag_load(x2f[ks], r2 + ks * 32 + g * 8);
        if constexpr (MODE == 0) {
          Frag8<bf16_t> of;
          frag_load(of, P.o + row * P.ldo + (long)head * DH + ks * 32 + g * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            dsum = fmaf(bf16_bits_to_f32((unsigned short)x2f[ks].v[j]), bf16_bits_to_f32((unsigned short)of.v[j]), dsum);
        }
      }
    }
    if constexpr (MODE == 0) {
      dsum = wave_groups_sum(dsum);
      own_del = dsum;
      own_lse = active ? P.lse[row * G.heads + head] * L2E : 0.f;
      if (active && g == 0) P.delta[row * G.heads + head] = dsum;
    }
  }
  f32x4 acc1[MT];
  f32x4 acc2[MODE == 1 ? MT : 1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc1[mt] = (f32x4)(0.f);
  if constexpr (MODE == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc2[mt] = (f32x4)(0.f);
  }

  const int rbase = li * I::ROWP + g * 16;                                  // row fragment: visitor row li, chunk g
  const int tbase = (4 * g + (li >> 2)) * I::ROWP + (li & 3) * 8;           // transposed read: rows 4g..4g+3, 4 columns
  const int my_lo = max(h - G.eH, 0), my_hi = min(h + G.eH, H - 1);
  const int h0 = og * NW;
  const int t_lo = max(h0 - G.eH, 0), t_hi = min(min(h0 + NW - 1, H - 1) + G.eH, H - 1);
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int nch = (t_hi - t_lo + KC) / KC;
  const int nslab = (sk_hi - sk_lo + 1) * nch;

  auto issue = [&](int j) {
    const int pl = j / nch, c0 = t_lo + (j - pl * nch) * KC;
    const long plane_v = ((long)b * G.S + (sk_lo + pl)) * HW;
    const int nrow = min(KC, t_hi - c0 + 1) * 16;
    char* buf = smem + (j & 1) * I::BUF;
    stage_img<DH, NW>(buf, P.y1 + plane_v * P.ldy1 + (long)head * DH, P.ldy1, c0 * 16, nrow - 1, wave, lane);
    stage_img<DH, NW>(buf + I::IMG, P.y2 + plane_v * P.ldy2 + (long)head * DH, P.ldy2, c0 * 16, nrow - 1, wave, lane);
  };
  // MODE 1: per-visitor (query) lse2 / delta of a slab: plain loads issued BEFORE that slab's DMA (so waiting for them
  // never waits for the DMA), parked in registers during the previous slab's compute, written to LDS at its end
  float vl = 0.f, vd = 0.f;
  auto fetch_rows = [&](int j) {
    vl = 0.f; vd = 0.f;
    if (tid < KC * 16) {
      const int pl = j / nch, c0 = t_lo + (j - pl * nch) * KC;
      const long plane_v = ((long)b * G.S + (sk_lo + pl)) * HW;
      if (tid < min(KC, t_hi - c0 + 1) * 16) {
        vl = P.lse[(plane_v + c0 * 16 + tid) * G.heads + head] * L2E;
        vd = P.delta[(plane_v + c0 * 16 + tid) * G.heads + head];
      }
    }
  };
  auto put_rows = [&](int j) {
    if (tid < KC * 16) {
      float* vt = reinterpret_cast<float*>(smem + (j & 1) * I::BUF + 2 * I::IMG);
      vt[tid] = vl;
      vt[KC * 16 + tid] = vd;
    }
  };
  if constexpr (MODE == 1) { fetch_rows(0); put_rows(0); }
  issue(0);
  for (int j = 0; j < nslab; ++j) {
    const int pl = j / nch, c0 = t_lo + (j - pl * nch) * KC;
    const int c_hi = min(c0 + KC - 1, t_hi);
    const char* Y1s = smem + (j & 1) * I::BUF;
    const char* Y2s = Y1s + I::IMG;
    const float* vlse = reinterpret_cast<const float*>(Y1s + 2 * I::IMG);
    const float* vdel = vlse + KC * 16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + 1 < nslab) {
      if constexpr (MODE == 1) fetch_rows(j + 1);
      issue(j + 1);
    }
    const int lo = active ? max(c0, my_lo) : 1, hi = active ? min(c_hi, my_hi) : 0;
    for (int t0 = lo; t0 <= hi; t0 += 2) {
      const bool has1 = t0 + 1 <= hi;
      const int r0 = (t0 - c0) * 16, r1 = has1 ? r0 + 16 : r0;
      const int ro0 = rbase + r0 * I::ROWP, ro1 = rbase + r1 * I::ROWP;
      const int to0 = tbase + r0 * I::ROWP, to1 = tbase + r1 * I::ROWP;
      f32x4 s0 = (f32x4)(0.f), s1 = (f32x4)(0.f), d0 = (f32x4)(0.f), d1 = (f32x4)(0.f);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag8<bf16_t> a0, a1, b0, b1;
        a0.v = *reinterpret_cast<const s16x8*>(Y1s + ro0 + ks * 64);
        a1.v = *reinterpret_cast<const s16x8*>(Y1s + ro1 + ks * 64);
        b0.v = *reinterpret_cast<const s16x8*>(Y2s + ro0 + ks * 64);
        b1.v = *reinterpret_cast<const s16x8*>(Y2s + ro1 + ks * 64);
        mma16(s0, a0, x1f[ks]);
        mma16(s1, a1, x1f[ks]);
        mma16(d0, b0, x2f[ks]);
        mma16(d1, b1, x2f[ks]);
      }
      f32x4 l0, l1, e0, e1;
      if constexpr (MODE == 1) {
        l0 = *reinterpret_cast<const f32x4*>(vlse + r0 + 4 * g);
        l1 = *reinterpret_cast<const f32x4*>(vlse + r1 + 4 * g);
        e0 = *reinterpret_cast<const f32x4*>(vdel + r0 + 4 * g);
        e1 = *reinterpret_cast<const f32x4*>(vdel + r1 + 4 * g);
      }
      const float b1m = has1 ? 0.f : -INFINITY;
      float pv[8], dsv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ls0 = MODE == 1 ? l0[r] : own_lse, ls1 = MODE == 1 ? l1[r] : own_lse;
        const float de0 = MODE == 1 ? e0[r] : own_del, de1 = MODE == 1 ? e1[r] : own_del;
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s0[r], c2, bias[r]) - ls0);
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s1[r], c2, bias[r] + b1m) - ls1);
        pv[r] = p0;
        pv[4 + r] = p1;
        dsv[r] = p0 * (d0[r] - de0) * G.scale;
        dsv[4 + r] = p1 * (d1[r] - de1) * G.scale;
      }
      Frag8<bf16_t> dsf;
      frag_from_f32<bf16_t>(dsf, dsv);
      // transposed fragments by inline-asm reads: the builtin makes hipcc drain vmcnt (the next slab's LDS-DMA) in front
      // of every one of them (wmz_common.h: ds_read_tr16_asm)
      {
        const unsigned ya0 = lds_addr(Y1s + to0), ya1 = lds_addr(Y1s + to1);
        s16x4 x0[MT], x1[MT];
        static_for<MT>([&](auto mt) {
          x0[mt] = ds_read_tr16_asm<mt * 32>(ya0);
          x1[mt] = ds_read_tr16_asm<mt * 32>(ya1);
        });
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]), "+v"(x1[mt]));
          Frag8<bf16_t> yf;
          yf.v = __builtin_shufflevector(x0[mt], x1[mt], 0, 1, 2, 3, 4, 5, 6, 7);
          mma16(acc1[mt], yf, dsf);
        }
      }
      if constexpr (MODE == 1) {
        Frag8<bf16_t> pf;
        frag_from_f32<bf16_t>(pf, pv);
        const unsigned ya0 = lds_addr(Y2s + to0), ya1 = lds_addr(Y2s + to1);
        s16x4 x0[MT], x1[MT];
        static_for<MT>([&](auto mt) {
          x0[mt] = ds_read_tr16_asm<mt * 32>(ya0);
          x1[mt] = ds_read_tr16_asm<mt * 32>(ya1);
        });
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]), "+v"(x1[mt]));
          Frag8<bf16_t> yf;
          yf.v = __builtin_shufflevector(x0[mt], x1[mt], 0, 1, 2, 3, 4, 5, 6, 7);
          mma16(acc2[mt], yf, pf);
        }
      }
    }
    if constexpr (MODE == 1) {
      if (j + 1 < nslab) put_rows(j + 1);
    }
  }
  if (!active) return;
  const long orow = plane_o + h * 16 + li;
  bf16_t* g1 = P.g1 + orow * P.ldg1 + (long)head * DH;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    s16x4 pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(acc1[mt][r]);
    *reinterpret_cast<s16x4*>(g1 + mt * 16 + 4 * g) = pk;
  }
  if constexpr (MODE == 1) {
    bf16_t* g2 = P.g2 + orow * P.ldg2 + (long)head * DH;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      s16x4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(acc2[mt][r]);
      *reinterpret_cast<s16x4*>(g2 + mt * 16 + 4 * g) = pk;
    }
  }
}

template <int DH, int MODE, int NW>
int launch_one(const RBwdPtrs& P, AttnGeom G, hipStream_t st) {
  G.qgroups = wmz_cdiv(G.H, NW);
  const long nwg = (long)G.B * G.heads * G.S * G.qgroups;
  const size_t smem = 2 * (size_t)BImg<DH>::BUF;
  auto kern = attn_bwd_row16_kernel<DH, MODE, NW>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), smem, st, P, G);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_bwd(row16)");
  return WMZ_OK;
}

template <int DH>
int launch_both(const RBwdPtrs& PQ, const RBwdPtrs& PK, const AttnGeom& G, hipStream_t st) {
  int rc = launch_one<DH, 0, 16>(PQ, G, st);
  if (rc != WMZ_OK) return rc;
  return launch_one<DH, 1, 8>(PK, G, st);
}

}  // namespace

int wmz_attn_bwd_row16_dispatch(const void* q, const void* k, const void* v, const void* out, const float* lse,
                                const void* dout, void* dq, void* dk, void* dv, float* delta, const AttnGeom& G, long lddo,
                                long lddq, long lddk, long lddv, hipStream_t st) {
  RBwdPtrs PQ, PK;
  PQ.x1 = (const bf16_t*)q; PQ.x2 = (const bf16_t*)dout; PQ.y1 = (const bf16_t*)k; PQ.y2 = (const bf16_t*)v;
  PQ.o = (const bf16_t*)out; PQ.lse = lse; PQ.delta = delta; PQ.g1 = (bf16_t*)dq; PQ.g2 = nullptr;
  PQ.ldx1 = G.ldq; PQ.ldx2 = lddo; PQ.ldy1 = G.ldk; PQ.ldy2 = G.ldv; PQ.ldo = G.ldo; PQ.ldg1 = lddq; PQ.ldg2 = 0;
  PK.x1 = (const bf16_t*)k; PK.x2 = (const bf16_t*)v; PK.y1 = (const bf16_t*)q; PK.y2 = (const bf16_t*)dout;
  PK.o = nullptr; PK.lse = lse; PK.delta = delta; PK.g1 = (bf16_t*)dk; PK.g2 = (bf16_t*)dv;
  PK.ldx1 = G.ldk; PK.ldx2 = G.ldv; PK.ldy1 = G.ldq; PK.ldy2 = lddo; PK.ldo = 0; PK.ldg1 = lddk; PK.ldg2 = lddv;
  if (G.dh == 128) return launch_both<128>(PQ, PK, G, st);
  if (G.dh == 64) return launch_both<64>(PQ, PK, G, st);
  return launch_both<32>(PQ, PK, G, st);
}
