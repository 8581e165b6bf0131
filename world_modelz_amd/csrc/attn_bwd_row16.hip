// Local windowed 3D attention backward, fast path for 16-wide planes (bf16, dim_head in {32,64,128}).  Same math and the
// same gather-form / two-role structure as attn_bwd.hip (which stays the general / fp32 path), on the forward fast path's
// slab machinery (attn_fwd_row16.hip):
//   MODE 0 (owner = one query row per wave, 16 waves):  dQ = scale * sum_j dS_ij K_j ;  delta_i = rowsum(dO_i * O_i)
//   MODE 1 (owner = one key row per wave,    8 waves):  dK = scale * sum_i dS_ij Q_i ;  dV = sum_i P_ij dO_i
// P_ij = exp2(c2 * s_ij + bias - lse2_i), dS_ij = P_ij (dO_i . V_j - delta_i).  Owner rows are MFMA B operands in registers,
// the visiting rows' two tensors (K,V | Q,dO) are the slab images: row fragments for S^T and dP^T, transposed reads
// (ds_read_b64_tr_b16) for acc^T += Y^T . dS^T.
//
// What the second version took over from the forward kernel:
//   * a slab holds every OTHER row of a 16-row chunk (plane row = base + 2 * slab row): whatever its own row, a wave finds
//     about half of its +-eH visiting rows in each slab.  With slabs of 8 CONSECUTIVE rows a wave did all of a plane's
//     work in one slab and idled at the other's barrier -- up to twice the step-times per plane;
//   * an odd last row of a slab is a 16-row step (4 + 4 MFMAs and MFMA 16x16x16 for the accumulation), not a half-empty
//     32-row one;
//   * the slab descriptors (64-bit bases, row limits) advance incrementally and AHEAD of the barrier, the per-lane LDS-DMA
//     source offsets are computed once: right behind a barrier the CU's scalar unit is shared by all of its waves;
//   * the visiting planes are walked in the order rotated by the plane index, so that the workgroups sharing a plane
//     stage it at the same time (one L2 miss, the others hit);
//   * the per-element arithmetic folds the constants: exponent = fma(s, c2, bias - lse2), dS = P * fma(dP, scale, -delta *
//     scale) (MODE 0: both addends live in registers for the whole kernel).
#include "attn_common.h"
#ifndef WMZ_KV_RING
#define WMZ_KV_RING 2       // dk | dv plane kernel: row fragments in flight per wave (2 or 3) ...
#endif
#ifndef WMZ_KV_TRING
#define WMZ_KV_TRING 3      // ... and transposed fragment pairs in flight
#endif
#ifndef WMZ_ABWD_ABL
#define WMZ_ABWD_ABL 0      // timing ablations (tools/build_variant.py; results are garbage): 1 no compute, 2 no slab DMA, 4 no LDS fragment reads, 8 no MFMAs, 16 no exp / dS arithmetic
#endif

namespace {

// A slab holds KC_ visiting rows (every other row of a chunk of 2 KC_ rows): 8 for whole 16-row planes, 2 for the small planes of
// the 4-wave workgroup shape (attn_fwd_row16.hip: planes of up to 8 tile rows, i.e. 8-wide planes of up to 16 rows).
template <int DH, int KC_ = 8> struct BImg {
  static constexpr int ROWP = DH * 2 + 32;             // both read kinds hit this image: +32 B keeps tr reads conflict-free
  static constexpr int IMG = KC_ * 16 * ROWP;
  static constexpr int BUF = 2 * IMG + 2 * 128 * 4;    // Y1 | Y2 | visitor lse2 | visitor delta (room for 128 visitors each)
  static constexpr int PIECES = IMG / 1024;
  static_assert(IMG % 1024 == 0, "image must be whole 1 KB DMA pieces");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// byte offset (relative to plane row `base`) of the 16 bytes this lane fetches for DMA piece `piece`: slab row r = 16 t + w
// is plane row base + 2 t, column w; pad chunks fetch chunk 0; rows past the plane are redirected to row_lim (never read)
template <int DH>
__device__ __forceinline__ unsigned bpiece_voff(int piece, int lane, unsigned ld_bytes, int row_lim) {
  constexpr int ROWP = DH * 2 + 32;
  const int off = piece * 1024 + lane * 16;
  const int r = off / ROWP;
  int c = (off - r * ROWP) >> 4;
  c = c < DH / 8 ? c : 0;
  const int prow = min(2 * (r >> 4), row_lim);
  return (unsigned)((prow << 4) + (r & 15)) * ld_bytes + (unsigned)c * 16u;
}

// Epilogue store of an owner row's accumulators acc[mt][r] = G^T[feature 16 mt + 4 g + r][owner li] (times mul) as bf16: lanes
// g and g ^ 1 exchange halves (v_permlane16_swap) so that every lane stores 16 contiguous bytes -- per wave instruction 64
// contiguous bytes per owner row instead of 32, half as many stores (the forward kernel's epilogue, attn_fwd_row16.hip).
template <int MT>
__device__ __forceinline__ void store_owner_rows(bf16_t* row, const f32x4 (&acc)[MT], float mul, int g) {
  if constexpr (MT % 2 == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; mt += 2) {
      const s16x4 a = cvt_pk4_bf16(acc[mt][0] * mul, acc[mt][1] * mul, acc[mt][2] * mul, acc[mt][3] * mul);
      const s16x4 b = cvt_pk4_bf16(acc[mt + 1][0] * mul, acc[mt + 1][1] * mul, acc[mt + 1][2] * mul, acc[mt + 1][3] * mul);
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      const u32x2 au = __builtin_bit_cast(u32x2, a), bu = __builtin_bit_cast(u32x2, b);
      const auto s0 = __builtin_amdgcn_permlane16_swap(au[0], bu[0], false, false);     // odd rows of a <-> even rows of b
      const auto s1 = __builtin_amdgcn_permlane16_swap(au[1], bu[1], false, false);
      i32x4 pk;
      pk[0] = (int)s0[0]; pk[1] = (int)s1[0]; pk[2] = (int)s0[1]; pk[3] = (int)s1[1];
      const int col = (g & 1) ? (mt + 1) * 16 + 4 * (g - 1) : mt * 16 + 4 * g;
      *reinterpret_cast<i32x4*>(row + col) = pk;
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      *reinterpret_cast<s16x4*>(row + mt * 16 + 4 * g) = cvt_pk4_bf16(acc[mt][0] * mul, acc[mt][1] * mul, acc[mt][2] * mul, acc[mt][3] * mul);
  }
}

struct RBwdPtrs {
  const bf16_t *x1, *x2, *y1, *y2, *o;
  const float* lse;
  float* delta;
  bf16_t *g1, *g2;
  long ldx1, ldx2, ldy1, ldy2, ldo, ldg1, ldg2;
};

// W8 (8-wide planes handed over as H / 2 tile rows of 16, see attn_fwd_row16.hip): the window is |D| <= eHv = ceil(eH / 2) tile
// rows, the column mask compares c & 7, and in the two rim rows |D| = eHv a pair is inside only if |2 D + dp| <= eH with
// dp = sub-row of the lane's visitor columns - sub-row of its owner column: one select per bias there.
template <int DH, int MODE, int NW, int KC, bool ALIGNED, bool W8>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? (MODE == 0 && !W8 ? 4 : 2) : 1)) void attn_bwd_row16_kernel(RBwdPtrs P, AttnGeom G) {
  using I = BImg<DH, KC>;
  constexpr int CH = 2 * KC, LOG_CH = CH == 16 ? 4 : 2;
  static_assert((1 << LOG_CH) == CH, "chunks of 16 or 4 rows");
  constexpr int KS = DH / 32, MT = DH / 16;
  constexpr int NP = (I::PIECES + NW - 1) / NW;          // DMA pieces per wave and image
  __shared__ __attribute__((aligned(1024))) char smem[2 * I::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int s = lid % G.S; lid /= G.S;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  const int HW = G.HW, H = G.H;
  const int h0 = og * NW;
  const int h = h0 + wave;
  const bool active = h < H;
  const long plane_o = ((long)b * G.S + s) * HW;
  const float L2E = 1.4426950408889634f;
  const float c2 = G.scale * L2E;

  // window bias of this lane's 4 visitor columns (w = 4g + r) against its owner column (w = li): symmetric in the roles
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int d = W8 ? ((4 * g + r) & 7) - (li & 7) : 4 * g + r - li;
    bias[r] = (d <= G.eW && -d <= G.eW) ? 0.f : -INFINITY;
  }
  const int eHv = W8 ? (G.eH + 1) >> 1 : G.eH;
  const int dp = (g >> 1) - (li >> 3);
  const bool ok_lo = (2 * eHv - dp <= G.eH) && (dp - 2 * eHv <= G.eH);    // W8: pair inside the window at D = -eHv / D = +eHv
  const bool ok_hi = (2 * eHv + dp <= G.eH) && (-2 * eHv - dp <= G.eH);
  auto rim = [&](int D, auto& o) {                        // a visiting row's four biases / accumulator starts, masked in place
    if constexpr (W8) {
      if (D == -eHv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = ok_lo ? o[r] : -INFINITY;
      } else if (D == eHv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = ok_hi ? o[r] : -INFINITY;
      }
    }
  };

  // ---- slab geometry (as in attn_fwd_row16.hip)
  const int my_lo = max(h - eHv, 0), my_hi = min(h + eHv, H - 1);
  const int t_lo = max(h0 - eHv, 0), t_hi = min(min(h0 + NW - 1, H - 1) + eHv, H - 1);
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int c_first = t_lo >> LOG_CH, c_last = t_hi >> LOG_CH;
  const int nch = (c_last - c_first + 1) * 2;               // slabs per visiting plane: (chunk) x (row parity)
  const int nslab = (sk_hi - sk_lo + 1) * nch;
  const unsigned ld1_b = (unsigned)P.ldy1 * 2u, ld2_b = (unsigned)P.ldy2 * 2u;
  unsigned vo1[NP], vo2[NP];
  if constexpr (ALIGNED) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      vo1[i] = bpiece_voff<DH>(wave + NW * i, lane, ld1_b, CH - 2);
      vo2[i] = bpiece_voff<DH>(wave + NW * i, lane, ld2_b, CH - 2);
    }
  }
  const unsigned rs1 = 16u * ld1_b, rs2 = 16u * ld2_b;
  const long ps1 = (long)HW * (long)ld1_b, ps2 = (long)HW * (long)ld2_b;
  const int nwin = 2 * G.eS + 1;
  int p_first = s - G.eS;
  { const int a = (((G.S - 1 - p_first) % nwin) + nwin) % nwin; p_first += a; }
  if (p_first > sk_hi || p_first < sk_lo) p_first = sk_lo;
  const char* y1pl = (const char*)(P.y1 + ((long)b * G.S + p_first) * HW * P.ldy1 + (long)head * DH);
  const char* y2pl = (const char*)(P.y2 + ((long)b * G.S + p_first) * HW * P.ldy2 + (long)head * DH);
  long vrow_pl = ((long)b * G.S + p_first) * HW;         // first (b, s, h, w) row index of the next slab's visiting plane
  int pl_n = p_first - sk_lo, rem_n = 0, base_n = 0, jn = 0;
  const char* y1p = nullptr;
  const char* y2p = nullptr;
  char* dbuf = nullptr;
  int dlim = CH - 2;
  long vrow_n = 0;
  auto next_state = [&]() {
    base_n = ((c_first + (rem_n >> 1)) << LOG_CH) + (rem_n & 1);
    const int bsafe = min(base_n, H - 1);                // (a one-row plane has no odd row: fetch that unread slab from inside the plane)
    y1p = y1pl + (unsigned)bsafe * rs1;
    y2p = y2pl + (unsigned)bsafe * rs2;
    dbuf = smem + (jn & 1) * I::BUF;
    dlim = max(H - 1 - base_n, 0);
    vrow_n = vrow_pl + (long)bsafe * 16;
  };
  auto advance = [&]() {
    ++jn;
    if (++rem_n == nch) {
      rem_n = 0;
      if (sk_lo + pl_n == sk_hi) { y1pl -= (long)pl_n * ps1; y2pl -= (long)pl_n * ps2; vrow_pl -= (long)pl_n * HW; pl_n = 0; }
      else { ++pl_n; y1pl += ps1; y2pl += ps2; vrow_pl += HW; }
    }
  };
  auto issue = [&]() {                                   // all pieces of the slab described by the current state
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = wave + NW * i;
      if (i * NW + NW <= I::PIECES || piece < I::PIECES) {
        unsigned a, c;
        if constexpr (ALIGNED) { a = vo1[i]; c = vo2[i]; }
        else { a = bpiece_voff<DH>(piece, lane, ld1_b, dlim); c = bpiece_voff<DH>(piece, lane, ld2_b, dlim); }
        __builtin_amdgcn_global_load_lds((gptr_t)(y1p + a), (lptr_t)(dbuf + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(y2p + c), (lptr_t)(dbuf + I::IMG + piece * 1024), 16, 0, 0);
      }
    }
    if constexpr (MODE == 1) {
      // per-visitor (query) lse / delta of the slab (slab row r = 16 t + w <-> plane row base + 2 t, column w) ride the same
      // DMA queue, one dword per lane: waves 0, 1 bring the 128 lse values, waves 2, 3 the 128 deltas.  (As plain loads parked
      // in registers they made hipcc drain vmcnt(0) -- i.e. the slab just requested -- at the first LDS read of every step.)
      if (wave < 4) {
        const int idx = (wave & 1) * 64 + lane;             // slab key 0 .. 127
        const int prow = min(2 * (idx >> 4), dlim);
        const long row = vrow_n + (long)prow * 16 + (idx & 15);
        const float* src = (wave < 2 ? P.lse : P.delta) + row * G.heads + head;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dbuf + 2 * I::IMG + (wave >> 1) * 512 + (wave & 1) * 256), 4, 0, 0);
      }
    }
  };
  next_state();
  if (nslab > 0) issue();

  // ---- owner rows (after the first slab's requests: both are in flight together)
  Frag8<bf16_t> x1f[KS], x2f[KS];
  float bl[4], nde = 0.f;                                // MODE 0: the accumulators' initial values -- (bias - own lse2) / c2 per
                                                         // column (S' = Q.K + that, exponent = c2 S') and -own delta (D' = dP - delta)
  {
    const long row = plane_o + (active ? h : 0) * 16 + li;
    const bf16_t* r1 = P.x1 + row * P.ldx1 + (long)head * DH;
    const bf16_t* r2 = P.x2 + row * P.ldx2 + (long)head * DH;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
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
#pragma unroll
    for (int r = 0; r < 4; ++r) bl[r] = bias[r];
    if constexpr (MODE == 0) {
      dsum = wave_groups_sum(dsum);
      const float own_lse = P.lse[row * G.heads + head] * L2E;
#pragma unroll
      for (int r = 0; r < 4; ++r) bl[r] = (bias[r] - own_lse) / c2;
      nde = -dsum;
      if (active && g == 0) {
        // workspace for the dk | dv pass: delta, and behind it (one row of [tokens, heads] further) -lse / scale -- the initial
        // values of its dP and S accumulators
        P.delta[row * G.heads + head] = dsum;
        P.delta[(long)G.B * G.S * HW * G.heads + row * G.heads + head] = -P.lse[row * G.heads + head] / G.scale;
      }
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

  int base = base_n;
  advance();
  next_state();
  for (int j = 0; j < nslab; ++j) {
    const char* Y1s = smem + (j & 1) * I::BUF;
    const char* Y2s = Y1s + I::IMG;
    const float* vlse = reinterpret_cast<const float*>(Y1s + 2 * I::IMG);
    const float* vdel = vlse + 128;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = j + 1 < nslab;
    if (more) {
      if (!(WMZ_ABWD_ABL & 2)) issue();
    }
    const int lo = max(0, (my_lo - base + 1) >> 1), hi = min(KC - 1, (my_hi - base) >> 1);   // slab rows this wave needs
    if (active && !(WMZ_ABWD_ABL & 1)) {
      int t0 = lo;
      for (; t0 + 1 <= hi; t0 += 2) {
        // ---- two visiting rows: S^T and dP^T, then P, dS, then the accumulations
        const int ro0 = rbase + t0 * 16 * I::ROWP, ro1 = ro0 + 16 * I::ROWP;
        const int to0 = tbase + t0 * 16 * I::ROWP;
        f32x4 s0 = (f32x4)(0.f), s1 = (f32x4)(0.f), d0 = (f32x4)(0.f), d1 = (f32x4)(0.f);
        float bA[4] = {bias[0], bias[1], bias[2], bias[3]}, bB[4] = {bias[0], bias[1], bias[2], bias[3]};     // MODE 1: the rows' biases
        if constexpr (MODE == 0) {
          s0 = s1 = (f32x4){bl[0], bl[1], bl[2], bl[3]};           // accumulator starts; rim rows of 8-wide planes masked in place
          d0 = d1 = (f32x4)(nde);
          rim(base + 2 * t0 - h, s0);
          rim(base + 2 * t0 + 2 - h, s1);
        } else {
          rim(base + 2 * t0 - h, bA);
          rim(base + 2 * t0 + 2 - h, bB);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Frag8<bf16_t> a0, a1, b0, b1;
          if constexpr (WMZ_ABWD_ABL & 4) { a0.v = x1f[ks].v; a1.v = x2f[ks].v; b0.v = x1f[ks].v; b1.v = x2f[ks].v; }
          else {
            a0.v = *reinterpret_cast<const s16x8*>(Y1s + ro0 + ks * 64);
            a1.v = *reinterpret_cast<const s16x8*>(Y1s + ro1 + ks * 64);
            b0.v = *reinterpret_cast<const s16x8*>(Y2s + ro0 + ks * 64);
            b1.v = *reinterpret_cast<const s16x8*>(Y2s + ro1 + ks * 64);
          }
          if constexpr (WMZ_ABWD_ABL & 8) { asm volatile("" :: "v"(a0.v), "v"(a1.v), "v"(b0.v), "v"(b1.v)); }
          else {
            mma16(s0, a0, x1f[ks]);
            mma16(s1, a1, x1f[ks]);
            mma16(d0, b0, x2f[ks]);
            mma16(d1, b1, x2f[ks]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // transposed fragments of Y1 requested now (inline asm: the builtin makes hipcc drain the LDS-DMA in front of each)
        const unsigned ya0 = lds_addr(Y1s + to0), ya1 = ya0 + 16 * I::ROWP;
        s16x4 x0[MT], x1[MT];
        if constexpr (WMZ_ABWD_ABL & 4) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { x0[mt] = (s16x4)((short)ya0); x1[mt] = (s16x4)((short)ya1); }
        } else {
          static_for<MT>([&](auto mt) {
            x0[mt] = ds_read_tr16_asm<mt * 32>(ya0);
            x1[mt] = ds_read_tr16_asm<mt * 32>(ya1);
          });
        }
        // MODE 1 (two waves per SIMD, 256 registers): Y2's transposed fragments are requested here as well, so that BOTH
        // accumulations run back to back behind one wait instead of read -> wait -> MFMA twice
        s16x4 z0[MODE == 1 ? MT : 1], z1[MODE == 1 ? MT : 1];
        if constexpr (MODE == 1) {
          const unsigned yb0 = lds_addr(Y2s + to0), yb1 = yb0 + 16 * I::ROWP;
          if constexpr (WMZ_ABWD_ABL & 4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { z0[mt] = (s16x4)((short)yb0); z1[mt] = (s16x4)((short)yb1); }
          } else {
            static_for<MT>([&](auto mt) {
              z0[mt] = ds_read_tr16_asm<mt * 32>(yb0);
              z1[mt] = ds_read_tr16_asm<mt * 32>(yb1);
            });
          }
        }
        float pv[8], dsv[8];
        if constexpr (MODE == 1) {
          const int r0 = t0 * 16 + 4 * g, r1 = r0 + 16;
          const f32x4 l0 = *reinterpret_cast<const f32x4*>(vlse + r0), l1 = *reinterpret_cast<const f32x4*>(vlse + r1);
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(vdel + r0), e1 = *reinterpret_cast<const f32x4*>(vdel + r1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(s0[r], c2, fmaf(l0[r], -L2E, bA[r])));     // (lse arrives in natural-log units)
            const float p1 = __builtin_amdgcn_exp2f(fmaf(s1[r], c2, fmaf(l1[r], -L2E, bB[r])));
            pv[r] = p0;
            pv[4 + r] = p1;
            dsv[r] = p0 * ((d0[r] - e0[r]) * G.scale);
            dsv[4 + r] = p1 * ((d1[r] - e1[r]) * G.scale);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dsv[r] = __builtin_amdgcn_exp2f(s0[r] * c2) * d0[r];          // dS / scale: the scale waits for the epilogue
            dsv[4 + r] = __builtin_amdgcn_exp2f(s1[r] * c2) * d1[r];
          }
        }
        Frag8<bf16_t> dsf, pf;
        frag_from_f32<bf16_t>(dsf, dsv);
        if constexpr (MODE == 1) frag_from_f32<bf16_t>(pf, pv);
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]), "+v"(x1[mt]));
          Frag8<bf16_t> yf;
          yf.v = __builtin_shufflevector(x0[mt], x1[mt], 0, 1, 2, 3, 4, 5, 6, 7);
          if constexpr (WMZ_ABWD_ABL & 8) { asm volatile("" :: "v"(yf.v), "v"(dsf.v)); } else mma16(acc1[mt], yf, dsf);
        }
        if constexpr (MODE == 1) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            asm volatile("" : "+v"(z0[mt]), "+v"(z1[mt]));
            Frag8<bf16_t> yf;
            yf.v = __builtin_shufflevector(z0[mt], z1[mt], 0, 1, 2, 3, 4, 5, 6, 7);
            if constexpr (WMZ_ABWD_ABL & 8) { asm volatile("" :: "v"(yf.v), "v"(pf.v)); } else mma16(acc2[mt], yf, pf);
          }
        }
      }
      if (t0 <= hi) {
        // ---- odd last visiting row of the slab: a 16-row step
        const int ro0 = rbase + t0 * 16 * I::ROWP;
        const int to0 = tbase + t0 * 16 * I::ROWP;
        f32x4 s0 = (f32x4)(0.f), d0 = (f32x4)(0.f);
        float bA[4] = {bias[0], bias[1], bias[2], bias[3]};
        if constexpr (MODE == 0) { s0 = (f32x4){bl[0], bl[1], bl[2], bl[3]}; d0 = (f32x4)(nde); rim(base + 2 * t0 - h, s0); }
        else rim(base + 2 * t0 - h, bA);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Frag8<bf16_t> a0, b0;
          a0.v = *reinterpret_cast<const s16x8*>(Y1s + ro0 + ks * 64);
          b0.v = *reinterpret_cast<const s16x8*>(Y2s + ro0 + ks * 64);
          mma16(s0, a0, x1f[ks]);
          mma16(d0, b0, x2f[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned ya0 = lds_addr(Y1s + to0);
        s16x4 x0[MT];
        static_for<MT>([&](auto mt) { x0[mt] = ds_read_tr16_asm<mt * 32>(ya0); });
        float pv[4], dsv[4];
        if constexpr (MODE == 1) {
          const int r0 = t0 * 16 + 4 * g;
          const f32x4 l0 = *reinterpret_cast<const f32x4*>(vlse + r0), e0 = *reinterpret_cast<const f32x4*>(vdel + r0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pv[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c2, fmaf(l0[r], -L2E, bA[r])));
            dsv[r] = pv[r] * ((d0[r] - e0[r]) * G.scale);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) dsv[r] = __builtin_amdgcn_exp2f(s0[r] * c2) * d0[r];
        }
        s16x4 dsf, pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) dsf[r] = (short)f32_to_bf16_bits(dsv[r]);
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]));
          acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x0[mt], dsf, acc1[mt], 0, 0, 0);
        }
        if constexpr (MODE == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pf[r] = (short)f32_to_bf16_bits(pv[r]);
          const unsigned yb0 = lds_addr(Y2s + to0);
          s16x4 z0[MT];
          static_for<MT>([&](auto mt) { z0[mt] = ds_read_tr16_asm<mt * 32>(yb0); });
          ds_tr_wait();
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            asm volatile("" : "+v"(z0[mt]));
            acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(z0[mt], pf, acc2[mt], 0, 0, 0);
          }
        }
      }
    }
    base = base_n;
    advance();
    next_state();
  }
  if (!active) return;
  const long orow = plane_o + h * 16 + li;
  bf16_t* g1 = P.g1 + orow * P.ldg1 + (long)head * DH;
  store_owner_rows<MT>(g1, acc1, MODE == 0 ? G.scale : 1.f, g);
  if constexpr (MODE == 1) store_owner_rows<MT>(P.g2 + orow * P.ldg2 + (long)head * DH, acc2, 1.f, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// dk | dv pass, whole-plane form (round 4): 16 waves = the 16 key rows of one (b, head, s) plane IN ONE WORKGROUP, written
// for the 128 registers a wave has at four waves per SIMD (the 8-wave form above holds 229: one workgroup per CU, a plane's
// two half-plane workgroups each stage every visiting plane, the launch runs as two rounds of workgroups).
// Register plan at dim_head 128: dK^T / dV^T accumulators 64, the owner's K and (negated) V fragments 32, three running LDS
// addresses 3 -- 99 resident, which leaves 29 for a step.  What makes a step fit:
//   * the per-visitor row constants are the INITIAL VALUE of the S and dP accumulators, read from LDS straight into them: the dq
//     pass leaves delta and -lse / scale per token (ws), so S' = Q.K - lse / scale needs only exp2(c2 S') and, with the owner's V
//     negated, D' = delta - dO.V = -(dP - delta): dS = -scale P D' and the -scale waits for the epilogue.  No lse / delta
//     registers, no subtract, no second multiply per element;
//   * the window's column mask is four lane masks in scalar registers (v_cndmask on P), not four bias registers;
//   * the LDS-DMA source offsets of a wave's pieces live in an LDS table (12 KB of the 14 KB the two slab buffers leave), read back
//     per slab, not in registers;
//   * one visitor operand's fragments at a time: row fragments four in flight (16 registers), transposed fragments as a
//     double-buffered pair of 2-feature-tile chunks (16 registers); every LDS read is inline asm retired by a counted lgkmcnt
//     wait that names what it releases, every MFMA result is pinned at its program point (the scheduler may not widen a live range).
// Requires whole 16-row chunks (H % 16 == 0); everything else takes the 8-wave form.  SAME_LD: both visiting tensors have one row
// stride and the table holds the byte offsets themselves (otherwise row and column, two multiply-adds per piece and tensor).
template <int DH, bool SAME_LD>
__global__ __launch_bounds__(1024) void attn_bwd_kvplane_kernel(RBwdPtrs P, AttnGeom G, const float* ws, float c2) {   // c2 = scale * log2(e): a kernel argument stays scalar
  using I = BImg<DH>;
  constexpr int NW = 16, KC = 8, KS = DH / 32, MT = DH / 16;
  constexpr int NP = (I::PIECES + NW - 1) / NW;
  constexpr int HDR = 2 * KC * 16 * 4;                   // delta | -lse / scale of the slab's 128 visitors
  constexpr int BUFB = HDR + 2 * I::IMG;
  constexpr int R16 = 16 * I::ROWP;                      // one slab row = 16 visitors
  constexpr int Y2O = I::IMG;
  constexpr int TAB = 2 * BUFB;
  static_assert(HDR + Y2O + R16 + DH * 2 < 65536, "fragment offsets must fit the ds_read immediate");
  __shared__ __attribute__((aligned(1024))) char smem[2 * BUFB + NW * NP * 256];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int s = lid % G.S; lid /= G.S;
  const int head = lid % G.heads;
  const int b = lid / G.heads;
  const int HW = G.HW, H = G.H;
  const int h0 = og * NW;
  const int h = h0 + wave;                               // H % 16 == 0: every wave owns a key row
  const long NH = (long)G.B * G.S * HW * G.heads;

  // ---- slab geometry (as above)
  const int my_lo = max(h - G.eH, 0), my_hi = min(h + G.eH, H - 1);
  const int t_lo = max(h0 - G.eH, 0), t_hi = min(h0 + NW - 1 + G.eH, H - 1);
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int c_first = t_lo >> 4, c_last = t_hi >> 4;
  const int nch = (c_last - c_first + 1) * 2;
  const int nslab = (sk_hi - sk_lo + 1) * nch;
  const unsigned ld1_b = (unsigned)P.ldy1 * 2u, ld2_b = (unsigned)P.ldy2 * 2u;      // SAME_LD: equal
  const unsigned rs1 = 16u * ld1_b, rs2 = 16u * ld2_b;
  const long ps1 = (long)HW * (long)ld1_b, ps2 = (long)HW * (long)ld2_b;
  const int nwin = 2 * G.eS + 1;
  int p_first = s - G.eS;
  { const int a = (((G.S - 1 - p_first) % nwin) + nwin) % nwin; p_first += a; }
  if (p_first > sk_hi || p_first < sk_lo) p_first = sk_lo;
  const char* y1pl = (const char*)(P.y1 + ((long)b * G.S + p_first) * HW * P.ldy1 + (long)head * DH);
  const char* y2pl = (const char*)(P.y2 + ((long)b * G.S + p_first) * HW * P.ldy2 + (long)head * DH);
  long vrow_pl = ((long)b * G.S + p_first) * HW;
  int pl_n = p_first - sk_lo, rem_n = 0, base_n = 0, jn = 0;
  const char* y1p = nullptr;
  const char* y2p = nullptr;
  char* dbuf = nullptr;
  long vrow_n = 0;
  auto next_state = [&]() {
    base_n = ((c_first + (rem_n >> 1)) << 4) + (rem_n & 1);
    y1p = y1pl + (unsigned)base_n * rs1;
    y2p = y2pl + (unsigned)base_n * rs2;
    dbuf = smem + (jn & 1) * BUFB;
    vrow_n = vrow_pl + (long)base_n * 16;
  };
  auto advance = [&]() {
    ++jn;
    if (++rem_n == nch) {
      rem_n = 0;
      if (sk_lo + pl_n == sk_hi) { y1pl -= (long)pl_n * ps1; y2pl -= (long)pl_n * ps2; vrow_pl -= (long)pl_n * HW; pl_n = 0; }
      else { ++pl_n; y1pl += ps1; y2pl += ps2; vrow_pl += HW; }
    }
  };
  auto issue = [&]() {
    const unsigned la = lds_addr(smem + TAB + wave * NP * 256) + lane_id_volatile() * 4u;
    unsigned vo[NP];
    static_for<NP>([&](auto i) { vo[i] = ds_read_u32_asm<i * 256>(la); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = wave + NW * i;
      if (i * NW + NW <= I::PIECES || piece < I::PIECES) {
        asm volatile("" : "+v"(vo[i]));
        unsigned a1 = vo[i], a2 = vo[i];
        if constexpr (!SAME_LD) {                          // table entry = slab-image row << 8 | byte offset inside the row
          a1 = (vo[i] >> 8) * ld1_b + (vo[i] & 0xffu);
          a2 = (vo[i] >> 8) * ld2_b + (vo[i] & 0xffu);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(y1p + a1), (lptr_t)(dbuf + HDR + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(y2p + a2), (lptr_t)(dbuf + HDR + Y2O + piece * 1024), 16, 0, 0);
      }
    }
    if (wave < 4) {                                        // the slab's 128 deltas (waves 0, 1) and -lse / scale (waves 2, 3)
      const int idx = (wave & 1) * 64 + (int)lane_id_volatile();
      const long row = vrow_n + (long)(2 * (idx >> 4)) * 16 + (idx & 15);
      const float* src = ws + (wave >> 1) * NH + row * G.heads + head;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dbuf + (wave >> 1) * (KC * 16 * 4) + (wave & 1) * 256), 4, 0, 0);
    }
  };

  // ---- the table of DMA source offsets, then the first slab's requests
  {
    const int lane = tid & 63;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      *reinterpret_cast<unsigned*>(smem + TAB + (wave * NP + i) * 256 + lane * 4) =
          SAME_LD ? bpiece_voff<DH>(wave + NW * i, lane, ld1_b, 14) : bpiece_voff<DH>(wave + NW * i, lane, 256u, 14);
  }
  next_state();
  if (nslab > 0) issue();

  // ---- owner rows: K as it is, V negated (sign flip of a bf16 is exact)
  s16x8 kf[KS], vf[KS];
  bool in0, in1, in2, in3;
  unsigned arow, atr, alse;                               // running LDS addresses of the step's three read patterns
  {
    const int lane = tid & 63, g = lane >> 4, li = lane & 15;
    const long row = ((long)b * G.S + s) * HW + h * 16 + li;
    const bf16_t* r1 = P.x1 + row * P.ldx1 + (long)head * DH;
    const bf16_t* r2 = P.x2 + row * P.ldx2 + (long)head * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const s16x8*>(r1 + ks * 32 + g * 8);
      vf[ks] = *reinterpret_cast<const s16x8*>(r2 + ks * 32 + g * 8) ^ (s16x8)((short)0x8000);
    }
    const int d0 = 4 * g - li;
    in0 = (d0 <= G.eW) && (-d0 <= G.eW);
    in1 = (d0 + 1 <= G.eW) && (-d0 - 1 <= G.eW);
    in2 = (d0 + 2 <= G.eW) && (-d0 - 2 <= G.eW);
    in3 = (d0 + 3 <= G.eW) && (-d0 - 3 <= G.eW);
    const unsigned base = lds_addr(smem);
    arow = base + li * I::ROWP + g * 16;
    atr = base + (4 * g + (li >> 2)) * I::ROWP + (li & 3) * 8;
    alse = base + g * 16;
  }
  f32x4 acc1[MT], acc2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { acc1[mt] = (f32x4)(0.f); acc2[mt] = (f32x4)(0.f); }

  // S' and D' of one visiting row (XO = its byte offset inside the slab image pair: 0 or R16; LO = the same in the header).
  // Row fragments travel in pairs: two in flight (8 registers) beside the two accumulators.
  auto sdp_row = [&](auto XO, auto LO, f32x4& sv, f32x4& dv) {
    constexpr int X = decltype(XO)::value, L = decltype(LO)::value;
    sv = ds_read_f32x4_asm<KC * 16 * 4 + L>(alse);
    if constexpr (KS >= 2) {
      // the two chains alternate (consecutive MFMAs never depend on each other); fragments Q0 dO0 Q1 dO1 ... in a ring of
      // WMZ_KV_RING (2 or 3) in flight
      dv = ds_read_f32x4_asm<L>(alse);
      constexpr int NF = 2 * KS, RING = WMZ_KV_RING;
      s16x8 fr[RING];
      auto rd = [&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        fr[i % RING] = ds_read_b128_asm<HDR + X + (i & 1) * Y2O + (i >> 1) * 64>(arow);
      };
      static_for<RING>([&](auto Ic) { rd(Ic); });
      static_for<NF>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        constexpr int behind = (NF - 1 - i) < (RING - 1) ? (NF - 1 - i) : (RING - 1);
        if constexpr ((i & 1) == 0) {
          lgkm_wait_for2<behind>(sv, fr[i % RING]);
          sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i % RING], kf[i >> 1], sv, 0, 0, 0);
          asm volatile("" : "+v"(sv));
        } else {
          lgkm_wait_for2<behind>(dv, fr[i % RING]);
          dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i % RING], vf[i >> 1], dv, 0, 0, 0);
          asm volatile("" : "+v"(dv));
        }
        if constexpr (i + RING < NF) rd(std::integral_constant<int, i + RING>{});
      });
    } else {
      dv = ds_read_f32x4_asm<L>(alse);
      s16x8 f0 = ds_read_b128_asm<HDR + X>(arow), g0 = ds_read_b128_asm<HDR + Y2O + X>(arow);
      lgkm_wait_for2<1>(sv, f0);
      sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, kf[0], sv, 0, 0, 0);
      lgkm_wait_for2<0>(dv, g0);
      dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g0, vf[0], dv, 0, 0, 0);
      asm volatile("" : "+v"(sv), "+v"(dv));
    }
  };
  // P (masked) and P * D' of one row, packed to bf16: two registers each
  auto pds_row = [&](const f32x4& sv, const f32x4& dv, s16x4& pk, s16x4& dk) {
    float p0 = __builtin_amdgcn_exp2f(sv[0] * c2), p1 = __builtin_amdgcn_exp2f(sv[1] * c2);
    float p2 = __builtin_amdgcn_exp2f(sv[2] * c2), p3 = __builtin_amdgcn_exp2f(sv[3] * c2);
    p0 = in0 ? p0 : 0.f; p1 = in1 ? p1 : 0.f; p2 = in2 ? p2 : 0.f; p3 = in3 ? p3 : 0.f;
    pk = cvt_pk4_bf16(p0, p1, p2, p3);
    dk = cvt_pk4_bf16(p0 * dv[0], p1 * dv[1], p2 * dv[2], p3 * dv[3]);
  };

  int base = base_n;
  advance();
  next_state();
  for (int j = 0; j < nslab; ++j) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + 1 < nslab && !(WMZ_ABWD_ABL & 2)) issue();
    const int lo = max(0, (my_lo - base + 1) >> 1), hi = min(KC - 1, (my_hi - base) >> 1);   // slab rows this wave needs
    if (lo <= hi && !(WMZ_ABWD_ABL & 1)) {
      const unsigned off0 = (unsigned)((j & 1) * BUFB + lo * R16), hoff0 = (unsigned)((j & 1) * BUFB + lo * 64);
      arow += off0; atr += off0; alse += hoff0;
      int t0 = lo;
      for (; t0 + 1 <= hi; t0 += 2) {
        // ---- two visiting rows (32 visitors): S', D' -> P, P D' -> dK^T += Q^T (P D'), dV^T += dO^T P on MFMA 16x16x32
        s16x8 pf, df;
        {
          f32x4 sv, dv;
          s16x4 pa, da, pb, db;
          sdp_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, sv, dv);
          pds_row(sv, dv, pa, da);
          asm volatile("" : "+v"(pa), "+v"(da));
          sdp_row(std::integral_constant<int, R16>{}, std::integral_constant<int, 64>{}, sv, dv);
          pds_row(sv, dv, pb, db);
          pf = __builtin_shufflevector(pa, pb, 0, 1, 2, 3, 4, 5, 6, 7);
          df = __builtin_shufflevector(da, db, 0, 1, 2, 3, 4, 5, 6, 7);
          asm volatile("" : "+v"(pf), "+v"(df));
        }
        // 2 MT feature tiles, one transposed fragment pair (rows A | B: 4 registers) each, a ring of three in flight: tiles
        // 0 .. MT-1 read Q (-> acc1 with P D'), the rest dO (-> acc2 with P)
        constexpr int TR = WMZ_KV_TRING < 2 * MT ? WMZ_KV_TRING : 2 * MT;
        s16x4 ca[TR], cbb[TR];
        auto rd_tile = [&](auto C) {
          constexpr int c = decltype(C)::value;
          constexpr int YO = HDR + (c >= MT ? Y2O : 0) + (c % MT) * 32;
          ca[c % TR] = ds_read_tr16_asm<YO>(atr);
          cbb[c % TR] = ds_read_tr16_asm<YO + R16>(atr);
        };
        static_for<TR>([&](auto C) { rd_tile(C); });
        static_for<2 * MT>([&](auto C) {
          constexpr int c = decltype(C)::value;
          constexpr int behind = (2 * MT - 1 - c) < (TR - 1) ? (2 * MT - 1 - c) : (TR - 1);      // younger tiles still in flight
          lgkm_wait_for2<2 * behind>(ca[c % TR], cbb[c % TR]);
          const s16x8 ya = __builtin_shufflevector(ca[c % TR], cbb[c % TR], 0, 1, 2, 3, 4, 5, 6, 7);
          if constexpr (c < MT) {
            acc1[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya, df, acc1[c], 0, 0, 0);
            asm volatile("" : "+v"(acc1[c]));
          } else {
            acc2[c - MT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya, pf, acc2[c - MT], 0, 0, 0);
            asm volatile("" : "+v"(acc2[c - MT]));
          }
          if constexpr (c + TR < 2 * MT) rd_tile(std::integral_constant<int, c + TR>{});
        });
        arow += 2 * R16; atr += 2 * R16; alse += 128;
      }
      if (t0 <= hi) {
        // ---- odd last visiting row of the slab: 16 visitors, the accumulations on MFMA 16x16x16
        s16x4 pa, da;
        {
          f32x4 sv, dv;
          sdp_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, sv, dv);
          pds_row(sv, dv, pa, da);
          asm volatile("" : "+v"(pa), "+v"(da));
        }
        s16x4 ca[4];
        auto rd_tile = [&](auto C) {
          constexpr int c = decltype(C)::value;
          ca[c % 4] = ds_read_tr16_asm<HDR + (c >= MT ? Y2O : 0) + (c % MT) * 32>(atr);
        };
        static_for<(2 * MT < 4 ? 2 * MT : 4)>([&](auto C) { rd_tile(C); });
        static_for<2 * MT>([&](auto C) {
          constexpr int c = decltype(C)::value;
          constexpr int behind = (2 * MT - 1 - c) < 3 ? (2 * MT - 1 - c) : 3;
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ca[c % 4]) : "n"(behind) : "memory");
          f32x4& acc = c < MT ? acc1[c % MT] : acc2[c % MT];
          acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ca[c % 4], c < MT ? da : pa, acc, 0, 0, 0);
          asm volatile("" : "+v"(acc));
          if constexpr (c + 4 < 2 * MT) rd_tile(std::integral_constant<int, c + 4>{});
        });
      }
      const unsigned back = (unsigned)((j & 1) * BUFB + t0 * R16), hback = (unsigned)((j & 1) * BUFB + t0 * 64);
      arow -= back; atr -= back; alse -= hback;
    }
    base = base_n;
    advance();
    next_state();
  }
  // ---- dK = -scale * acc1 (the sign of the negated V, the scale of dS), dV = acc2
  const int lane = (int)lane_id_volatile(), g = lane >> 4, li = lane & 15;
  const long orow = ((long)b * G.S + s) * HW + h * 16 + li;
  bf16_t* g1 = P.g1 + orow * P.ldg1 + (long)head * DH;
  bf16_t* g2 = P.g2 + orow * P.ldg2 + (long)head * DH;
  store_owner_rows<MT>(g1, acc1, -G.scale, g);
  store_owner_rows<MT>(g2, acc2, 1.f, g);
}

template <int DH, int MODE, int NW, int KC, bool W8>
int launch_one(const RBwdPtrs& P, AttnGeom G, hipStream_t st) {
  G.qgroups = wmz_cdiv(G.H, NW);
  const long nwg = (long)G.B * G.heads * G.S * G.qgroups;
  if ((G.H % (2 * KC)) == 0) hipLaunchKernelGGL((attn_bwd_row16_kernel<DH, MODE, NW, KC, true, W8>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, P, G);
  else hipLaunchKernelGGL((attn_bwd_row16_kernel<DH, MODE, NW, KC, false, W8>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, P, G);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_bwd(row16)");
  return WMZ_OK;
}

template <int DH>
int launch_both(const RBwdPtrs& PQ, const RBwdPtrs& PK, const AttnGeom& G, hipStream_t st) {
  if (G.w8) {
    // 8-wide planes (G.H = H / 2 tile rows): up to 8 tile rows on 4-wave workgroups with 2-row slabs, larger planes on the big shapes
    if (G.H <= 8) {
      const int rc = launch_one<DH, 0, 4, 2, true>(PQ, G, st);
      return rc != WMZ_OK ? rc : launch_one<DH, 1, 4, 2, true>(PK, G, st);
    }
    // (the dq pass with its rim masks does not fit 128 registers: 8-wave workgroups, two per plane chunk, for it as well)
    const int rc = launch_one<DH, 0, 8, 8, true>(PQ, G, st);
    return rc != WMZ_OK ? rc : launch_one<DH, 1, 8, 8, true>(PK, G, st);
  }
  int rc = launch_one<DH, 0, 16, 8, false>(PQ, G, st);
  if (rc != WMZ_OK) return rc;
#ifndef WMZ_ABWD_PLANE
#define WMZ_ABWD_PLANE 1    // 1: the whole-plane 16-wave dk | dv kernel where its preconditions hold; 0: always the 8-wave form
#endif
  if (WMZ_ABWD_PLANE && (G.H & 15) == 0) {
    AttnGeom Gp = G;
    Gp.qgroups = G.H / 16;
    const long nwg = (long)G.B * G.heads * G.S * Gp.qgroups;
    const float c2 = G.scale * 1.4426950408889634f;
    if (PK.ldy1 == PK.ldy2)
      hipLaunchKernelGGL((attn_bwd_kvplane_kernel<DH, true>), dim3((unsigned)nwg), dim3(1024), 0, st, PK, Gp, (const float*)PK.delta, c2);
    else
      hipLaunchKernelGGL((attn_bwd_kvplane_kernel<DH, false>), dim3((unsigned)nwg), dim3(1024), 0, st, PK, Gp, (const float*)PK.delta, c2);
    WMZ_LAUNCH_CHECK("wmz_local3d_attn_bwd(row16, dk|dv plane)");
    return WMZ_OK;
  }
  return launch_one<DH, 1, 8, 8, false>(PK, G, st);
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
