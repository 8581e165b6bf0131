// Local windowed 3D attention forward, fast path for the reference's canonical latent planes: W == 16 (a 16-query tile is
// exactly one row of the plane), bf16, dim_head in {32, 64, 128}.  Same algorithm and outputs as attn_fwd.hip (which stays
// the general / fp32 / parity path).
//
// Work split: a workgroup = 16 waves = 16 consecutive query rows of one (b, head, s) plane (the whole plane at the
// BASELINE shapes), one query row per wave; the key planes s-eS..s+eS arrive as slabs of 8 key rows (K and V, 64 KB +
// padding) in a double-buffered pair of LDS images by LDS-DMA.  A slab holds every OTHER row of a 16-row chunk (plane
// row = base + 2 * slab row), so whatever its own row a wave finds about half of its +-eH key rows in each slab and all
// waves are busy between two barriers.  Per wave and slab: key rows two at a time (32 keys per step):
//   S^T[32 keys x 16 queries] = K Q^T   MFMA 16x16x32 bf16, A = K rows from LDS (ds_read_b128), B = Q held in registers;
//   softmax in the log2 domain, the query on the lane (col = lane & 15), 8 keys in 8 accumulator registers;
//   O^T[dh x 16 queries] += V^T P^T     B = P straight from S^T's accumulator layout, A = V^T by ds_read_b64_tr_b16.
// An odd last key row runs as a 16-key step (4 MFMAs + 8 MFMA 16x16x16) instead of a half-empty 32-key one.
//
// What keeps the per-step instruction count down (the kernel is issue-bound, not MFMA-bound: DESIGN.md 4.1):
//   * the row test (|hk-hq| <= eH, plane bounds) is decided per key ROW by the loop bounds, the column test (|wk-wq| <= eW)
//     is four additive 0 / -inf biases per lane; the biases carry MINUS THE RUNNING MAX as well, so ONE fma takes a raw dot
//     product to the exponent of 2 (scale * log2 e folded in) -- no separate subtraction;
//   * the running max is a deferred reference: it only moves when some logit exceeds it by more than 2^DEFER, which the
//     wave decides from lane-local max3 trees and ONE wave-wide vote (no cross-lane traffic in the common step); the
//     cross-lane max, the O^T rescale and the bias refresh live in the rare branch;
//   * K / V slabs sit in PADDED LDS rows (K +16 B, V +32 B): every fragment address is `lane base + row offset +
//     immediate`, conflict-free for ds_read_b128 and ds_read_b64_tr_b16; padding is applied on the DMA's source side;
//   * the V^T fragments of a step are requested BEFORE its softmax, so their LDS latency runs under the VALU work;
//   * the LDS-DMA descriptors (per-lane source offsets of the wave's pieces) are computed once; per slab a piece costs
//     one scalar base and one instruction, and the pieces are spread over the step instead of queueing at the texture
//     addresser right behind the barrier (SPLIT).
#include "attn_common.h"
#include "wmz_debug.h"

// Compile-time ablations for timing experiments (tools/variants builds; results are garbage): bit 0 no LDS fragment reads,
// bit 1 no softmax arithmetic, bit 2 no MFMAs, bit 3 no V staging, bit 4 no K staging, bit 5 no slab loop at all, bit 6 one key plane only,
// bit 7 odd waves stage but do not compute.  0 in the product.
#ifndef WMZ_ATTN_ABL
#define WMZ_ATTN_ABL 0
#endif

namespace {

template <typename A, typename B>
__device__ __forceinline__ void mma16_abl(f32x4& acc, const A& a, const B& b) {
  if constexpr (WMZ_ATTN_ABL & 4) { asm volatile("" :: "v"(a.v), "v"(b.v)); } else { mma16(acc, a, b); }
}

// development probes (the only process-wide state of the library, see include/wmz.h): stamp buffer and A/B knobs
long long* g_attn_ts = nullptr;     // 16 waves x 64 int64, wmz_debug_attn_timestamps

// Workgroup shapes.  A plane is cut into chunks of CH rows, a slab holds every RS-th row of a chunk (KC = CH / RS rows: plane
// row = base + RS * slab row), two slabs (K and V images each) in LDS, one in flight.
//   Big   (NW 16, CH 16, KC 8): a workgroup = 16 query rows = a whole 16x16 plane, two 74 KB slabs, one workgroup per CU;
//   Small (NW  4, CH  4, KC 2): a workgroup = 4 query rows, two 18 KB slabs, four workgroups per CU -- planes of up to 8 tile rows
//         (8-wide planes, below), where the big shape would run a quarter full and stage 8-row slabs holding 2 rows.
// (Round 2 / 3 measured and dropped, now removed: 4-row slabs in a ring of four with three in flight, two 8-wave half-plane
//  workgroups per CU, a two-query-rows-per-wave kernel on MFMA 32x32x16: all equal or slower.)
template <int NW_, int CH_, int KC_> struct Shape {
  static constexpr int NW = NW_, CH = CH_, KC = KC_, RS = CH_ / KC_, LOG_RS = RS == 2 ? 1 : (RS == 1 ? 0 : 2);
  static_assert(KC_ * RS == CH_ && (1 << LOG_RS) == RS, "a slab is one phase of a chunk");
};
using Big = Shape<16, 16, 8>;
using Small = Shape<4, 4, 2>;
constexpr int NBUF = 2;               // LDS slab pair
constexpr float DEFER = 8.f;          // log2 units

template <int DH, typename SH> struct Img {
#ifndef WMZ_ATTN_KPAD
#define WMZ_ATTN_KPAD 32
#endif
  static constexpr int NW = SH::NW, KC = SH::KC;
  static constexpr int KROW = DH * 2 + WMZ_ATTN_KPAD, VROW = DH * 2 + 32;
  static constexpr int KIMG = KC * 16 * KROW, VIMG = KC * 16 * VROW;
  static constexpr int BUF = KIMG + VIMG;
  static constexpr int PK = KIMG / 1024, PV = VIMG / 1024;            // 1 KB DMA pieces per image
  static constexpr int NPK = (PK + NW - 1) / NW, NPV = (PV + NW - 1) / NW;   // ... per wave
  static_assert(KIMG % 1024 == 0 && VIMG % 1024 == 0, "images must be whole 1 KB DMA pieces");
};

// Issue points of the next slab's LDS-DMA pieces inside a slab iteration: 0 = right behind the barrier, 1 = behind the
// first step's QK^T MFMAs, 2 = behind its softmax, 3 = behind its PV MFMAs, 4 = after the wave's last step.  A wave has up
// to three K pieces and three V pieces; whatever a wave without a (first) step has left goes out at point 4.
constexpr int kSched[8][6] = {
    {0, 0, 0, 0, 0, 0},   // 0: everything behind the barrier
    {0, 0, 0, 1, 2, 4},   // 1: K behind the barrier, V spread over the first step
    {0, 1, 3, 0, 2, 4},   // 2: one piece per point
    {0, 0, 0, 0, 1, 2},   // 3
    {1, 1, 1, 3, 3, 3},   // 4: K under the QK^T MFMAs, V under the PV MFMAs
    {0, 1, 1, 1, 3, 3},   // 5
    {0, 0, 0, 1, 1, 1},   // 6: K behind the barrier, V under the QK^T MFMAs
    {0, 0, 1, 1, 3, 3},   // 7
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Byte offset (inside a key plane, relative to plane row `base`) of the 16 bytes this lane fetches for DMA piece `piece` of
// a padded image: the lane landing on (slab row, 16-byte chunk) fetches that chunk of plane row base + 2 * (slab row / 16),
// key column slab row % 16; pad chunks fetch chunk 0 (never read).  `row_lim`: rows past the plane are redirected to the
// last valid one (never read either).
template <int DH, int ROWP, int RS>
__device__ __forceinline__ unsigned piece_voff(int piece, int lane, unsigned ld_bytes, int row_lim) {
  const int off = piece * 1024 + lane * 16;
  const int r = off / ROWP;
  int c = (off - r * ROWP) >> 4;
  c = c < DH / 8 ? c : 0;
  const int prow = min(RS * (r >> 4), row_lim);
  return (unsigned)((prow << 4) + (r & 15)) * ld_bytes + (unsigned)c * 16u;
}

// W8: planes 8 wide.  A plane [H, 8] (H even) IS a plane [H / 2, 16] in memory: tile row R = plane rows 2R, 2R + 1, tile column
// c = 8 (row & 1) + w.  Staging, fragments, Q / O rows are those of a 16-wide plane with H / 2 rows (G.H holds H / 2); what changes
// is the window: key tile row R + D is visited for |D| <= eHv = ceil(eH / 2), the column mask compares c & 7, and in the two
// OUTERMOST tile rows (|D| = eHv) a (query, key) pair is inside the window only if |2 D + (c_k >> 3) - (c_q >> 3)| <= eH -- a
// per-lane condition (a lane's four key columns share one sub-row), applied as one select per logit in those steps only.  There a
// query can have nothing visible in a whole step, so the deferred running maximum is started per lane (firstl).
template <int DH, typename SH, int MODE, bool ALIGNED, bool PROBE, bool TS, bool W8>
__global__ __launch_bounds__(SH::NW * 64, 4) void attn_fwd_row16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                      const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                                      float* __restrict__ LSE, float* __restrict__ DBG,
                                                                      AttnGeom G, long long* ts) {
  using I = Img<DH, SH>;
  constexpr int NW = SH::NW, CH = SH::CH, KC = SH::KC, RS = SH::RS, LOG_RS = SH::LOG_RS;
  constexpr int LOG_CH = CH == 16 ? 4 : (CH == 8 ? 3 : 2);
  static_assert((1 << LOG_CH) == CH, "chunk of 4, 8 or 16 rows");
  constexpr int KS = DH / 32, MT = DH / 16;
  constexpr int SPLIT = MODE & 7;                        // where the LDS-DMA pieces of the next slab are issued (kSched)
  constexpr bool EPI16 = (MODE & 8) != 0;                // 16-byte output stores (lane pairs swap halves)
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * I::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
#define WMZ_ATS(slot) do { if constexpr (TS) { if (blockIdx.x == 0 && lane == 0) ts[wave * 64 + (slot)] = __builtin_readcyclecounter(); } } while (0)
  WMZ_ATS(0);
  if constexpr (TS) { if (blockIdx.x == 0 && lane == 0) ts[wave * 64 + 60] = __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int sq = lid % G.Sq; lid /= G.Sq;
  const int s = G.qs0 + sq;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  const int HW = G.HW, H = G.H;
  const int h0 = og * NW;                               // first query row of the workgroup
  const int hq = h0 + wave;                             // this wave's query row
  const bool act = hq < H && !((WMZ_ATTN_ABL & 128) && (wave & 1));   // (ablation bit 7: every other wave only stages)
  const long plane_q = ((long)b * G.S + s) * HW;
  const long plane_o = ((long)b * G.Sq + sq) * HW;
  const float c2 = G.scale * 1.4426950408889634f;

  // bm[r] = column-window bias of this lane's key column (w = 4g + r) against its query (w = li), minus the running max
  float bm[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int d = W8 ? ((4 * g + r) & 7) - (li & 7) : 4 * g + r - li;
    bm[r] = (d <= G.eW && -d <= G.eW) ? 0.f : -INFINITY;
  }
  const int eHv = W8 ? (G.eH + 1) >> 1 : G.eH;          // window extent in tile rows
  const int dp = (g >> 1) - (li >> 3);                  // W8: sub-row of the lane's key columns minus sub-row of its query
  const bool ok_lo = (2 * eHv - dp <= G.eH) && (dp - 2 * eHv <= G.eH);    // W8: pair inside the window at D = -eHv / D = +eHv
  const bool ok_hi = (2 * eHv + dp <= G.eH) && (-2 * eHv - dp <= G.eH);
  bool firstl = true;                                   // W8: this lane's query has not seen a logit yet

  Frag8<bf16_t> qf[KS];
  f32x4 o[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) o[mt] = (f32x4)(0.f);
  float m_run = 0.f, l_run = 0.f;                        // m_run: log2 units of the scaled logits; set by the first step
  bool first = true;

  const int kbase = li * I::KROW + g * 16;
  const int vbase = I::KIMG + (4 * g + (li >> 2)) * I::VROW + (li & 3) * 8;
  const int my_lo = max(hq - eHv, 0), my_hi = min(hq + eHv, H - 1);
  // key rows the workgroup stages
  const int t_lo = max(h0 - eHv, 0), t_hi = min(min(h0 + NW - 1, H - 1) + eHv, H - 1);
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int c_first = t_lo >> LOG_CH, c_last = t_hi >> LOG_CH;
  const int nch = (c_last - c_first + 1) * RS;              // slabs per key plane: (chunk) x (row phase)
  const int nslab = (WMZ_ATTN_ABL & 32) ? 0 : (WMZ_ATTN_ABL & 64) ? nch : (sk_hi - sk_lo + 1) * nch;   // ablations: no slab loop / one key plane

  // ---- LDS-DMA descriptors: this wave's pieces are wave + 16 i; per-lane source offsets for planes of whole 16-row chunks
  const unsigned ldk_b = (unsigned)G.ldk * 2u, ldv_b = (unsigned)G.ldv * 2u;
  // ALIGNED (H % 16 == 0): every staged chunk has all 16 rows, the offsets never change; otherwise rows past the plane
  // are clamped per slab and the offsets are recomputed at each use
  unsigned kvo[I::NPK], vvo[I::NPV];
  if constexpr (ALIGNED) {
#pragma unroll
    for (int i = 0; i < I::NPK; ++i) kvo[i] = piece_voff<DH, I::KROW, RS>(wave + NW * i, lane, ldk_b, CH - RS);
#pragma unroll
    for (int i = 0; i < I::NPV; ++i) vvo[i] = piece_voff<DH, I::VROW, RS>(wave + NW * i, lane, ldv_b, CH - RS);
  }

  // Scalar state of the slab being prefetched.  It is advanced INCREMENTALLY (adds, one 32-bit multiply) and at the END of
  // a slab iteration, in front of the barrier: right behind a barrier the scalar unit of the CU is shared by 16 waves, and
  // ~55 scalar instructions of index arithmetic per wave there cost every wave ~1k cycles per slab.
  const unsigned rsk = 16u * ldk_b, rsv = 16u * ldv_b;                   // bytes per plane row of 16 keys
  const long psk = (long)HW * (long)ldk_b, psv = (long)HW * (long)ldv_b; // bytes per key plane
  // Key planes are walked in a ROTATED order: at its t-th plane every workgroup reads the plane p = t (mod 2 eS + 1) of its
  // window, so the 2 eS + 1 workgroups that need plane p (query planes p - eS .. p + eS, one per CU of the same XCD, started
  // together and in step) stage it at the same time: one of them misses in L2, the others hit.  Walking s - eS .. s + eS in
  // order instead, a plane is read at 2 eS + 1 different times and has left the 4 MiB L2 (the clip's K / V alone fill it) in
  // between.  The softmax is order-independent (online), the logits probe records the plane it actually visits.
  // (The phase counts planes from the END of the clip: the trailing-planes entry point hands over only the last planes of a
  // clip, and its visiting order -- hence every rounding -- must be the full grid's.)
  const int nwin = 2 * G.eS + 1;
  int p_first = s - G.eS;
  { const int a = (((G.S - 1 - p_first) % nwin) + nwin) % nwin; p_first += a; }          // first plane of the window with S-1-p = 0 (mod nwin)
  if (p_first > sk_hi || p_first < sk_lo) p_first = sk_lo;
  const char* kpl = (const char*)(K + ((long)b * G.S + p_first) * HW * G.ldk + (long)head * DH);   // next slab's key plane
  const char* vpl = (const char*)(V + ((long)b * G.S + p_first) * HW * G.ldv + (long)head * DH);
  int pl_n = p_first - sk_lo, rem_n = 0, base_n = 0, jn = 0;             // next slab: plane (from sk_lo), slab in plane, first row, index
  const char* kp = nullptr;
  const char* vp = nullptr;
  char* dbuf = nullptr;
  int dlim = CH - RS;
  auto next_state = [&]() {                                              // descriptors of slab (pl_n, rem_n)
    base_n = ((c_first + (rem_n >> LOG_RS)) << LOG_CH) + (rem_n & (RS - 1));
    // (a one-row plane has no row of phase 1: that slab -- which no wave reads -- is fetched from the last row instead of from
    //  behind the plane; found by tools/guard_overread.py: a read past the end of the K / V tensor for H = 1)
    const int bsafe = min(base_n, H - 1);
    kp = kpl + (unsigned)bsafe * rsk;
    vp = vpl + (unsigned)bsafe * rsv;
    dbuf = smem + (jn & (NBUF - 1)) * I::BUF;
    dlim = max(H - 1 - base_n, 0);
  };
  auto advance = [&]() {                                                 // step (pl_n, rem_n) to the following slab
    ++jn;
    if (++rem_n == nch) {
      rem_n = 0;
      if (sk_lo + pl_n == sk_hi) { kpl -= (long)pl_n * psk; vpl -= (long)pl_n * psv; pl_n = 0; }     // wrap to the window's first plane
      else { ++pl_n; kpl += psk; vpl += psv; }
    }
  };
  auto issue_k = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < I::NPK && !(WMZ_ATTN_ABL & 16)) {
      const int piece = wave + NW * i;
      if (i * NW + NW <= I::PK || piece < I::PK) {       // wave-uniform
        unsigned vo;
        if constexpr (ALIGNED) vo = kvo[i]; else vo = piece_voff<DH, I::KROW, RS>(piece, lane, ldk_b, dlim);
        __builtin_amdgcn_global_load_lds((gptr_t)(kp + vo), (lptr_t)(dbuf + piece * 1024), 16, 0, 0);
      }
    }
  };
  auto issue_v = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < I::NPV && !(WMZ_ATTN_ABL & 8)) {
      const int piece = wave + NW * i;
      if (i * NW + NW <= I::PV || piece < I::PV) {
        unsigned vo;
        if constexpr (ALIGNED) vo = vvo[i]; else vo = piece_voff<DH, I::VROW, RS>(piece, lane, ldv_b, dlim);
        __builtin_amdgcn_global_load_lds((gptr_t)(vp + vo), (lptr_t)(dbuf + I::KIMG + piece * 1024), 16, 0, 0);
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
  static_assert(I::NPK <= 3 && I::NPV <= 3, "at most three pieces of each image per wave");
  const bool staging = !(G.dbg & 2);

  // ---- rare branch of the online softmax: move the running max by the cross-lane maximum `gm` of the step
  auto rescale = [&](float mx, auto& t) {
    constexpr int nt = (int)(sizeof(t) / sizeof(float));
    float gm = wave_groups_max(mx);
    if constexpr (W8) {
      // per lane: a query with nothing visible in this step (gm = -inf) neither moves nor starts its reference
      const bool dead = gm == -INFINITY;
      gm = dead ? 0.f : (firstl ? gm : fmaxf(gm, 0.f));
      const float alpha = firstl ? 1.f : __builtin_amdgcn_exp2f(-gm);     // (o and l of a lane that starts here are still zero)
      l_run *= alpha;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) o[mt] *= alpha;
      firstl = firstl && dead;
      first = __any(firstl);
    } else {
      if (!first) {
        gm = fmaxf(gm, 0.f);                               // the reference only moves up
        const float alpha = __builtin_amdgcn_exp2f(-gm);
        l_run *= alpha;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o[mt] *= alpha;
      }
      first = false;
    }
    m_run += gm;
#pragma unroll
    for (int r = 0; r < 4; ++r) bm[r] = (bm[r] == -INFINITY) ? -INFINITY : -m_run;
#pragma unroll
    for (int r = 0; r < nt; ++r) t[r] -= gm;
  };
  // W8: the logits of a key tile row at the window's rim (D = -eHv or +eHv) that this lane's query does not see
  auto rim_mask = [&](int D, float* t4) {
    if constexpr (W8) {
      if (D == -eHv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] = ok_lo ? t4[r] : -INFINITY;
      } else if (D == eHv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] = ok_hi ? t4[r] : -INFINITY;
      }
    }
  };

  WMZ_ATS(1);
  // the first slab is requested up front; bq / pq: first plane row and key plane of the slab in flight
  int bq = 0, pq = 0;
  if (0 < nslab) {
    next_state();
    if (staging) {
      issue_k(C0{}); issue_k(C1{}); issue_k(C2{});
      issue_v(C0{}); issue_v(C1{}); issue_v(C2{});
    }
    bq = base_n; pq = pl_n;
    advance();
  }
  {
    // Q right behind the first slab's requests (rows past the plane load a valid row, never stored)
    const bf16_t* qrow = Q + (plane_q + (act ? hq : 0) * 16 + li) * G.ldq + (long)head * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) frag_load(qf[ks], qrow + ks * 32 + g * 8);
  }
  next_state();                                          // descriptors of slab 1
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks].v));   // Q is waited for HERE, once (the compiler's wait is vmcnt(0): inside
                                                                       // the loop it would drain the slabs in flight at every iteration)
  WMZ_ATS(2);
  for (int j = 0; j < nslab; ++j) {
    const char* Sb = smem + (j & (NBUF - 1)) * I::BUF;     // this slab's K image, V image behind it
    const int pl = pq, base = bq;                          // current slab: key plane, plane row of slab row 0 (slab row r <-> base + RS r)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of slab j landed
    if (j < 14) WMZ_ATS(3 + 4 * j);
    __builtin_amdgcn_s_barrier();                        // ... everyone's did, and slab j-1 is retired: refill its slot
    if (j < 14) WMZ_ATS(4 + 4 * j);
    const bool more = j + 1 < nslab && staging;
    bool pend[6] = {more, more, more, more, more, more};                 // K0 K1 K2 V0 V1 V2 of slab j + 1 not requested yet
    auto issue_at = [&](auto pc, bool flush) {
      constexpr int pt = decltype(pc)::value;
      static_for<6>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if ((flush || kSched[SPLIT][i] == pt) && pend[i]) {
          if constexpr (i < 3) issue_k(std::integral_constant<int, i>{}); else issue_v(std::integral_constant<int, i - 3>{});
          pend[i] = false;
        }
      });
    };
    issue_at(C0{}, false);
    if (j < 14) WMZ_ATS(5 + 4 * j);
    const int lo = max(0, (my_lo - base + RS - 1) >> LOG_RS), hi = min(KC - 1, (my_hi - base) >> LOG_RS);   // slab rows this wave needs
    if (act && !(G.dbg & 1)) {
      int t0 = lo;
      for (; t0 + 1 <= hi; t0 += 2) {
        // ---- S^T = K Q^T for key rows t0, t0 + 1
        const int ko = kbase + t0 * 16 * I::KROW;
        const unsigned va0 = lds_addr(Sb + vbase + t0 * 16 * I::VROW), va1 = va0 + 16 * I::VROW;
        f32x4 sc0 = (f32x4)(0.f), sc1 = (f32x4)(0.f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Frag8<bf16_t> ka, kb;
          if constexpr (WMZ_ATTN_ABL & 1) { ka.v = qf[ks].v; kb.v = qf[ks].v; }
          else {
            ka.v = *reinterpret_cast<const s16x8*>(Sb + ko + ks * 64);
            kb.v = *reinterpret_cast<const s16x8*>(Sb + ko + 16 * I::KROW + ks * 64);
          }
          mma16_abl(sc0, ka, qf[ks]);
          mma16_abl(sc1, kb, qf[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- V^T fragments requested now: their LDS latency runs under the softmax
        s16x4 x0[MT], x1[MT];
        if constexpr (WMZ_ATTN_ABL & 1) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { x0[mt] = (s16x4)((short)va0); x1[mt] = (s16x4)((short)va1); }
        } else {
          static_for<MT>([&](auto mt) {
            x0[mt] = ds_read_tr16_asm<mt * 32>(va0);
            x1[mt] = ds_read_tr16_asm<mt * 32>(va1);
          });
        }
        issue_at(C1{}, false);
        if constexpr (PROBE) {
          const int kw = (2 * G.eW + 1), kh = (2 * G.eH + 1);
          const long qn = plane_o + hq * 16 + li;
          const int nk = (2 * G.eS + 1) * kh * kw;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int dw = W8 ? ((4 * g + r) & 7) - (li & 7) : 4 * g + r - li;
            if (dw <= G.eW && -dw <= G.eW) {
              const int ds = (sk_lo + pl) - s;
              const int D0 = base + RS * t0 - hq;
              const int dh0 = W8 ? 2 * D0 + dp : D0, dh1 = W8 ? dh0 + 2 * RS : dh0 + RS;      // in plane rows
              float* row = DBG + (qn * G.heads + head) * nk;
              if (dh0 <= G.eH && -dh0 <= G.eH) row[((ds + G.eS) * kh + (dh0 + G.eH)) * kw + dw + G.eW] = sc0[r] * G.scale;
              if (dh1 <= G.eH && -dh1 <= G.eH) row[((ds + G.eS) * kh + (dh1 + G.eH)) * kw + dw + G.eW] = sc1[r] * G.scale;
            }
          }
        }
        // ---- softmax: exponent of 2 by one fma per logit; deferred running max
        float t[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r] = fmaf(sc0[r], c2, bm[r]);
          t[4 + r] = fmaf(sc1[r], c2, bm[r]);
        }
        if constexpr (W8) {
          const int D0 = base + RS * t0 - hq;
          rim_mask(D0, t);
          rim_mask(D0 + RS, t + 4);
        }
        const float mx = fmaxf(__builtin_fmaxf(__builtin_fmaxf(t[0], t[1]), __builtin_fmaxf(t[2], t[3])),
                               __builtin_fmaxf(__builtin_fmaxf(t[4], t[5]), __builtin_fmaxf(t[6], t[7])));
        if (first || __any(mx > DEFER)) rescale(mx, t);
        float p[8];
        Frag8<bf16_t> pf;
        if constexpr (WMZ_ATTN_ABL & 2) {
#pragma unroll
          for (int r = 0; r < 8; ++r) pf.v[r] = (short)__builtin_bit_cast(int, sc0[r & 3]);
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) p[r] = __builtin_amdgcn_exp2f(t[r]);
          l_run += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
          frag_from_f32<bf16_t>(pf, p);
        }
        issue_at(C2{}, false);
        // ---- O^T += V^T P^T
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]), "+v"(x1[mt]));       // uses stay behind the wait
          Frag8<bf16_t> vf;
          vf.v = __builtin_shufflevector(x0[mt], x1[mt], 0, 1, 2, 3, 4, 5, 6, 7);
          mma16_abl(o[mt], vf, pf);
        }
        issue_at(C3{}, false);
      }
      if (t0 <= hi) {
        // ---- odd last key row of the slab: a 16-key step
        const int ko = kbase + t0 * 16 * I::KROW;
        const unsigned va0 = lds_addr(Sb + vbase + t0 * 16 * I::VROW);
        f32x4 sc0 = (f32x4)(0.f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Frag8<bf16_t> ka;
          ka.v = *reinterpret_cast<const s16x8*>(Sb + ko + ks * 64);
          mma16(sc0, ka, qf[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        s16x4 x0[MT];
        static_for<MT>([&](auto mt) { x0[mt] = ds_read_tr16_asm<mt * 32>(va0); });
        if constexpr (PROBE) {
          const int kw = (2 * G.eW + 1), kh = (2 * G.eH + 1);
          const long qn = plane_o + hq * 16 + li;
          const int nk = (2 * G.eS + 1) * kh * kw;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int dw = W8 ? ((4 * g + r) & 7) - (li & 7) : 4 * g + r - li;
            if (dw <= G.eW && -dw <= G.eW) {
              const int ds = (sk_lo + pl) - s;
              const int D0 = base + RS * t0 - hq;
              const int dh0 = W8 ? 2 * D0 + dp : D0;
              if (dh0 <= G.eH && -dh0 <= G.eH)
                DBG[(qn * G.heads + head) * nk + ((ds + G.eS) * kh + (dh0 + G.eH)) * kw + dw + G.eW] = sc0[r] * G.scale;
            }
          }
        }
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = fmaf(sc0[r], c2, bm[r]);
        if constexpr (W8) rim_mask(base + RS * t0 - hq, t);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(t[0], t[1]), __builtin_fmaxf(t[2], t[3]));
        if (first || __any(mx > DEFER)) rescale(mx, t);
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(t[r]);
        l_run += (p[0] + p[1]) + (p[2] + p[3]);
        s16x4 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[r] = (short)f32_to_bf16_bits(p[r]);
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]));
          o[mt] = op16_mfma_16x16x16(x0[mt], pf, o[mt]);
        }
      }
    }
    // whatever of slab j + 1 has not been requested yet
    issue_at(C3{}, true);
    // the slab just requested becomes the current one; its successor's descriptors are worked out here, ahead of the barrier
    bq = base_n; pq = pl_n;
    advance();
    next_state();
    if (j < 14) WMZ_ATS(6 + 4 * j);
  }
  WMZ_ATS(63);

  if (act) {
    float l = wave_groups_sum(l_run);
    const float inv = 1.f / l;
    bf16_t* orow = O + (plane_o + hq * 16 + li) * G.ldo + (long)head * DH;
    if constexpr (EPI16) {
      // lane (g, li) holds dh 16 mt + 4g .. +3 of query li; lanes g and g ^ 1 swap halves of an (mt, mt + 1) pair, so that the
      // even lane owns dh 16 mt + 4g .. +7 and the odd lane dh 16 (mt + 1) + 4 (g - 1) .. +7: one 16-byte store each
#pragma unroll
      for (int mt = 0; mt < MT; mt += 2) {
        unsigned a0 = (unsigned)f32_to_bf16_bits(o[mt][0] * inv) | ((unsigned)f32_to_bf16_bits(o[mt][1] * inv) << 16);
        unsigned a1 = (unsigned)f32_to_bf16_bits(o[mt][2] * inv) | ((unsigned)f32_to_bf16_bits(o[mt][3] * inv) << 16);
        unsigned b0 = (unsigned)f32_to_bf16_bits(o[mt + 1][0] * inv) | ((unsigned)f32_to_bf16_bits(o[mt + 1][1] * inv) << 16);
        unsigned b1 = (unsigned)f32_to_bf16_bits(o[mt + 1][2] * inv) | ((unsigned)f32_to_bf16_bits(o[mt + 1][3] * inv) << 16);
        const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);     // odd rows of a <-> even rows of b
        const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        i32x4 pk;
        pk[0] = (int)s0[0]; pk[1] = (int)s1[0]; pk[2] = (int)s0[1]; pk[3] = (int)s1[1];
        const int col = (g & 1) ? (mt + 1) * 16 + 4 * (g - 1) : mt * 16 + 4 * g;
        *reinterpret_cast<i32x4*>(orow + col) = pk;
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        s16x4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(o[mt][r] * inv);
        *reinterpret_cast<s16x4*>(orow + mt * 16 + 4 * g) = pk;
      }
    }
    if (LSE != nullptr && g == 0) LSE[(plane_o + hq * 16 + li) * G.heads + head] = m_run * 0.6931471805599453f + logf(l);
  }
  WMZ_ATS(62);
  if constexpr (TS) { if (blockIdx.x == 0 && lane == 0) ts[wave * 64 + 61] = __builtin_amdgcn_s_memrealtime(); }
#undef WMZ_ATS
}

template <int DH, typename SH, int MODE, bool ALIGNED, bool PROBE, bool TS, bool W8>
int launch_row16(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg, AttnGeom G, hipStream_t st) {
  G.qgroups = wmz_cdiv(G.H, SH::NW);
  const long nwg = (long)G.B * G.heads * G.Sq * G.qgroups;
  hipLaunchKernelGGL((attn_fwd_row16_kernel<DH, SH, MODE, ALIGNED, PROBE, TS, W8>), dim3((unsigned)nwg), dim3(SH::NW * 64), 0, st,
                     (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, lse, dbg, G, g_attn_ts);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_fwd(row16)");
  return WMZ_OK;
}

template <typename SH, int SPLIT, bool ALIGNED, bool PROBE, bool TS, bool W8>
int by_dh2(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg, const AttnGeom& G, hipStream_t st) {
  if (G.dh == 128) return launch_row16<128, SH, SPLIT, ALIGNED, PROBE, TS, W8>(q, k, v, out, lse, dbg, G, st);
  if (G.dh == 64) return launch_row16<64, SH, SPLIT, ALIGNED, PROBE, TS, W8>(q, k, v, out, lse, dbg, G, st);
  return launch_row16<32, SH, SPLIT, ALIGNED, PROBE, TS, W8>(q, k, v, out, lse, dbg, G, st);
}
template <typename SH, int SPLIT, bool PROBE, bool TS, bool W8>
int by_dh(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg, const AttnGeom& G, hipStream_t st) {
  if ((G.H % SH::CH) == 0) return by_dh2<SH, SPLIT, true, PROBE, TS, W8>(q, k, v, out, lse, dbg, G, st);
  return by_dh2<SH, SPLIT, false, PROBE, TS, W8>(q, k, v, out, lse, dbg, G, st);
}

}  // namespace

// Development probe: per-wave s_memtime stamps of workgroup 0 (tools/ts_attn.py); nullptr switches back to the product
// instantiation, which carries no stamp code at all.
#ifndef WMZ_OP16_F16
extern "C" int wmz_debug_attn_timestamps(void* buf) { g_attn_ts = (long long*)buf; return WMZ_OK; }
#endif

#ifndef WMZ_ATTN_MODE
#define WMZ_ATTN_MODE 9          // kSched[1]: K pieces behind the barrier, V pieces inside the first step; 16-byte output stores
#endif

// Called by wmz_local3d_attn_fwd when the shape qualifies (bf16, dim_head in {32,64,128}; W == 16, or W == 8 with an even number
// of rows: G.w8 set and G.H = H / 2 tile rows).  dbg: optional logits probe [N, heads, window] (natural-log-domain scaled logits of
// the in-window slots, pre-filled with -1e9 by the caller).
int WMZ_FN(wmz_attn_fwd_row16_dispatch)(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg,
                                const AttnGeom& G, hipStream_t st) {
  if (G.w8) {
    // planes of up to 8 tile rows (the reference's 8x8 latents: 4) on 4-wave workgroups with 2-row slabs, larger ones on the big shape
    if (G.H <= 8) {
      if (dbg != nullptr) return by_dh<Small, WMZ_ATTN_MODE, true, false, true>(q, k, v, out, lse, dbg, G, st);
      return by_dh<Small, WMZ_ATTN_MODE, false, false, true>(q, k, v, out, lse, nullptr, G, st);
    }
    if (dbg != nullptr) return by_dh<Big, WMZ_ATTN_MODE, true, false, true>(q, k, v, out, lse, dbg, G, st);
    return by_dh<Big, WMZ_ATTN_MODE, false, false, true>(q, k, v, out, lse, nullptr, G, st);
  }
  if (dbg != nullptr) return by_dh<Big, WMZ_ATTN_MODE, true, false, false>(q, k, v, out, lse, dbg, G, st);
  if (g_attn_ts != nullptr) return by_dh<Big, WMZ_ATTN_MODE, false, true, false>(q, k, v, out, lse, nullptr, G, st);
  // (the other LDS-DMA schedules / store widths of kSched are timing variants: build them with
  //  tools/build_variant.py <tag> attn_fwd_row16.hip -DWMZ_ATTN_MODE=<0..15> and load the library through WMZ_LIB_PATH)
  return by_dh<Big, WMZ_ATTN_MODE, false, false, false>(q, k, v, out, lse, nullptr, G, st);
}
