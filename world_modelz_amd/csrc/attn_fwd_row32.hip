// Local windowed 3D attention forward, second-generation fast path for 16-wide planes with an even number of rows: bf16,
// W == 16, H % 2 == 0, dim_head in {32, 64, 128}.  Same algorithm, plane order and outputs (to rounding) as
// attn_fwd_row16.hip, which stays for odd H, and as attn_fwd.hip (general / fp32 / parity path).
//
// What changed against row16, and why: row16 gives every wave ONE query row (a 16 x 16 score tile per key
// row, MFMA 16x16x32) and runs 16 waves = 4 per SIMD in lockstep between two barriers; its launch is one workgroup's
// critical path with the matrix pipe ~40 % busy.  Here a wave owns TWO ADJACENT query rows (32 queries) and works on
// 32 x 32 score tiles (two key rows x two query rows, MFMA 32x32x16):
//   * every K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) read from LDS feeds both query rows: half the
//     LDS reads per score, and half the MFMA issue slots per flop (an MFMA holds the SIMD's issue port for 8 cycles whether it
//     is a 16x16x32 or a 32x32x16);
//   * the two query rows' key-row windows overlap in 2 eH of their 2 eH + 2 rows, so with key rows paired per parity slab
//     (below) an interior wave has exactly two full tiles in EVERY slab: uniform work between two barriers;
//   * 8 waves per workgroup = 2 per SIMD with up to 256 VGPRs each: room to keep the next tile's S^T accumulator in flight
//     while the current tile's softmax runs on the vector ALU (software pipelining inside a wave instead of lockstep waves).
//
// Work split: a workgroup = 8 waves = 16 consecutive query rows of one (b, head, s) plane, wave w owns rows h0 + 2w, + 1.
// Key planes s-eS..s+eS arrive (in the rotated order of row16: the workgroups that share a key plane stage it together) as
// slabs of 8 key rows, K image then V image, in a double-buffered pair of LDS images by LDS-DMA; a slab holds every OTHER
// row of a 16-row chunk.  Per wave and slab, key rows two at a time (a TILE = slab rows ta, ta + 1 = 32 keys):
//   S^T[32 keys x 32 queries] = K Q^T + C  MFMA 32x32x16 bf16, A = K rows from LDS, B = Q held in registers, C = the
//                                          column-window bias (0 / -inf per lane and accumulator register) as the INITIAL
//                                          accumulator: the band mask costs no instruction;
//   softmax in the log2 domain, query on the lane (column l & 31), 16 keys in 16 accumulator registers, deferred running
//   max (one wave vote per tile; the cross-lane maximum, the O^T rescale live in the rare branch);
//   O^T[dh x 32 queries] += V^T P^T        B = P straight from S^T's accumulator layout (the order in which a PV k-step
//                                          walks its 16 keys is DEFINED by that layout), A = V^T by ds_read_b64_tr_b16.
// Rows of a tile that are outside one of the two query rows' windows (the far end of the union window; a dummy row when a
// wave's row count in the slab is odd) are masked per 16 x 16 sub-block by 8 selects, only in tiles that need it.
#include "attn_common.h"

// Timing-variant builds only (tools/build_variant.py ... -DWMZ_ATTN32_TS): s_memtime stamps of every wave of workgroup 0
#ifdef WMZ_ATTN32_TS
__device__ long long* g_attn32_ts = nullptr;              // [8 waves][256] stamps
extern "C" int wmz_debug_attn32_ts(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn32_ts), &buf, sizeof(buf)); }
#define WMZ_TS32(slot) do { if (blockIdx.x == 0 && lane == 0 && (slot) < 256) g_attn32_ts[wave * 256 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define WMZ_TS32(slot) do { } while (0)
#endif

namespace {

constexpr int NW = 8;                 // waves per workgroup; 2 query rows per wave
constexpr int QROWS = 2 * NW;         // query rows per workgroup
constexpr int KC = 8;                 // key rows per slab
constexpr int LOG_RS = 1, RS = 2;     // a slab holds plane rows base, base + RS, ..: RS slabs ("phases") per 16-row chunk
constexpr int NBUF = 2;
constexpr float DEFER = 8.f;          // log2 units

template <int DH> struct Img {
  // row pitches: K rows are read by ds_read_b128 with 32 keys on the lanes (pitch = 4 dwords mod 64 banks), V rows by
  // ds_read_b64_tr_b16 as 4 keys x 64 bytes per half wave (pitch = 16 dwords mod 64 banks)
  static constexpr int KROW = DH * 2 + 16, VROW = DH * 2 + (DH == 32 ? 0 : 64);
  static constexpr int KIMG = KC * 16 * KROW, VIMG = KC * 16 * VROW;
  static constexpr int BUF = KIMG + VIMG;
  static constexpr int PK = KIMG / 1024, PV = VIMG / 1024;                    // 1 KB DMA pieces per image
  static constexpr int NPK = (PK + NW - 1) / NW, NPV = (PV + NW - 1) / NW;    // ... per wave
  static_assert(KIMG % 1024 == 0 && VIMG % 1024 == 0, "images must be whole 1 KB DMA pieces");
  static_assert(NBUF * BUF <= 160 * 1024, "slab ring exceeds the LDS");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Byte offset (inside a key plane, relative to plane row `base`) of the 16 bytes this lane fetches for DMA piece `piece` of a
// padded image: the lane landing on (slab key r, 16-byte chunk c) fetches that chunk of plane row base + RS (r / 16), key
// column r % 16; pad chunks fetch chunk 0 (never read).  row_lim: rows past the plane are redirected to the last valid one.
template <int DH, int ROWP>
__device__ __forceinline__ unsigned piece_voff(int piece, int lane, unsigned ld_bytes, int row_lim) {
  const int off = piece * 1024 + lane * 16;
  const int r = off / ROWP;
  int c = (off - r * ROWP) >> 4;
  c = c < DH / 8 ? c : 0;
  const int prow = min(RS * (r >> 4), row_lim);
  return (unsigned)((prow << 4) + (r & 15)) * ld_bytes + (unsigned)c * 16u;
}

__device__ __forceinline__ float pair_max(float v) {               // lanes l, l ^ 32 (the two lanes of a query)
  const unsigned u = __float_as_uint(v);
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int DH, bool ALIGNED, bool PROBE>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_row32_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                  const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                                  float* __restrict__ LSE, float* __restrict__ DBG, AttnGeom G) {
  using I = Img<DH>;
  constexpr int KS = DH / 16;           // QK^T k-steps (MFMA 32x32x16)
  constexpr int MB = DH / 32;           // 32-wide dh blocks of O^T
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * I::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q32 = lane & 31, hi = lane >> 5;          // query of the wave's 32 (column of S^T / O^T), k-half
  const int jq = q32 >> 4, wq = q32 & 15;             // query row of the pair, query column

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int sq = lid % G.Sq; lid /= G.Sq;
  const int s = G.qs0 + sq;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  const int HW = G.HW, H = G.H;
  const int h0 = og * QROWS;                          // first query row of the workgroup
  const int hA = h0 + 2 * wave, hB = hA + 1;          // this wave's query rows (H is even: hB < H whenever hA < H)
  const bool act = hA < H;
  const long plane_q = ((long)b * G.S + s) * HW;
  const long plane_o = ((long)b * G.Sq + sq) * HW;
  const float c2 = G.scale * 1.4426950408889634f;

  // bm[r & 7] = column-window bias (0 / -inf) of (key column of accumulator register r, this lane's query column) MINUS the
  // running reference max: one fma takes a raw dot product to the exponent of 2.  The same eight values serve the tile's
  // first (r < 8) and second (r >= 8) key row.
  float bm[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int wk = (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int d = wk - wq;
    bm[r] = (d <= G.eW && -d <= G.eW) ? 0.f : -INFINITY;
  }

  Frag8<bf16_t> qf[KS];
  f32x16 o[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mb][r] = 0.f;
  // running state of this lane's query: reference max m_run (log2 units; moves only in the rare branch), partial row sum
  // l_run (this lane's 16 key slots; the query's two lanes are added in the epilogue), ub = +inf until the query has seen
  // its first in-window key (m_run is meaningless until then), -inf afterwards -- it rides in the max tree, so an
  // uninitialised query forces the rare branch
  float m_run = 0.f, l_run = 0.f, ub = INFINITY;

  // K fragment: lane (m = l & 31, hi) reads key m of the tile (m < 16: first row, else second = next slab row: 32 consecutive
  // image rows), 16 bytes at k-offset 8 hi of each 16-wide k-step
  const int kbase = q32 * I::KROW + hi * 16;
  // V^T fragment: 16-lane group gq = (l >> 4) & 1 covers dh 16 gq .. + 15 of a 32-wide block, lane 4 q' + p of the group
  // addresses key 4 hi + q' (+ 8 for the second read), dh 4 p .. 4 p + 3
  const int i16 = lane & 15;
  const int vbase = I::KIMG + (4 * hi + (i16 >> 2)) * I::VROW + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;

  const int my_lo = max(hA - G.eH, 0), my_hi = min(hB + G.eH, H - 1);
  // key rows the workgroup stages
  const int t_lo = max(h0 - G.eH, 0), t_hi = min(min(h0 + QROWS - 1, H - 1) + G.eH, H - 1);
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int c_first = t_lo >> 4, c_last = t_hi >> 4;
  const int nch = (c_last - c_first + 1) * RS;              // slabs per key plane: (16-row chunk) x (row phase)
  const int nslab = (sk_hi - sk_lo + 1) * nch;

  // ---- LDS-DMA descriptors: this wave's pieces are wave + 8 i
  const unsigned ldk_b = (unsigned)G.ldk * 2u, ldv_b = (unsigned)G.ldv * 2u;
  unsigned kvo[I::NPK], vvo[I::NPV];
  if constexpr (ALIGNED) {
#pragma unroll
    for (int i = 0; i < I::NPK; ++i) kvo[i] = piece_voff<DH, I::KROW>(wave + NW * i, lane, ldk_b, 16 - RS);
#pragma unroll
    for (int i = 0; i < I::NPV; ++i) vvo[i] = piece_voff<DH, I::VROW>(wave + NW * i, lane, ldv_b, 16 - RS);
  }

  // Scalar state of the slab being prefetched, advanced incrementally at the END of a slab iteration (row16: right behind a
  // barrier the CU's one scalar unit is shared by every wave).  Key planes are walked in the order rotated by the distance
  // from the END of the clip (row16 / DESIGN.md 4.1: the workgroups sharing a key plane stage it at the same time; the
  // trailing-planes entry point must visit in the full grid's order).
  const unsigned rsk = 16u * ldk_b, rsv = 16u * ldv_b;
  const long psk = (long)HW * (long)ldk_b, psv = (long)HW * (long)ldv_b;
  const int nwin = 2 * G.eS + 1;
  int p_first = s - G.eS;
  { const int a = (((G.S - 1 - p_first) % nwin) + nwin) % nwin; p_first += a; }
  if (p_first > sk_hi || p_first < sk_lo) p_first = sk_lo;
  const char* kpl = (const char*)(K + ((long)b * G.S + p_first) * HW * G.ldk + (long)head * DH);
  const char* vpl = (const char*)(V + ((long)b * G.S + p_first) * HW * G.ldv + (long)head * DH);
  int pl_n = p_first - sk_lo, rem_n = 0, base_n = 0, jn = 0;
  const char* kp = nullptr;
  const char* vp = nullptr;
  char* dbuf = nullptr;
  int dlim = 14;
  auto next_state = [&]() {
    base_n = ((c_first + (rem_n >> LOG_RS)) << 4) + (rem_n & (RS - 1));
    kp = kpl + (unsigned)base_n * rsk;
    vp = vpl + (unsigned)base_n * rsv;
    dbuf = smem + (jn & (NBUF - 1)) * I::BUF;
    dlim = max(H - 1 - base_n, 0);
  };
  auto advance = [&]() {
    ++jn;
    if (++rem_n == nch) {
      rem_n = 0;
      if (sk_lo + pl_n == sk_hi) { kpl -= (long)pl_n * psk; vpl -= (long)pl_n * psv; pl_n = 0; }
      else { ++pl_n; kpl += psk; vpl += psv; }
    }
  };
  auto issue_slab = [&]() {
    static_for<I::NPK>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int piece = wave + NW * i;
      if (i * NW + NW <= I::PK || piece < I::PK) {           // wave-uniform
        unsigned vo;
        if constexpr (ALIGNED) vo = kvo[i]; else vo = piece_voff<DH, I::KROW>(piece, lane, ldk_b, dlim);
        __builtin_amdgcn_global_load_lds((gptr_t)(kp + vo), (lptr_t)(dbuf + piece * 1024), 16, 0, 0);
      }
    });
    static_for<I::NPV>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int piece = wave + NW * i;
      if (i * NW + NW <= I::PV || piece < I::PV) {
        unsigned vo;
        if constexpr (ALIGNED) vo = vvo[i]; else vo = piece_voff<DH, I::VROW>(piece, lane, ldv_b, dlim);
        __builtin_amdgcn_global_load_lds((gptr_t)(vp + vo), (lptr_t)(dbuf + I::KIMG + piece * 1024), 16, 0, 0);
      }
    });
  };
  const bool staging = !(G.dbg & 2);

  // first slab requested up front, Q right behind it (rows past the plane load a valid row, never stored)
  int base_c = 0, pl_c = 0;                                // current slab: first plane row, key plane
  if (nslab > 0) {
    next_state();
    if (staging) issue_slab();
    base_c = base_n; pl_c = pl_n;
    advance();
  }
  {
    const bf16_t* qrow = Q + (plane_q + (act ? hA : 0) * 16 + q32) * G.ldq + (long)head * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { if (G.dbg & 128) qf[ks].v = (s16x8)(0); else frag_load(qf[ks], qrow + ks * 16 + hi * 8); }
  }
  next_state();                                            // descriptors of slab 1
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks].v));   // Q is waited for HERE, once

  // ---- The wave's tiles, each in two halves:
  //   H1(T)  K fragments, S^T = K Q^T (8 MFMAs), request of the V^T fragments          -- matrix pipe + LDS
  //   H2(T)  softmax of S^T (vector ALU), O^T += V^T P^T (8 MFMAs)
  // Waves w and w + 4 share a SIMD.  Run in lockstep they would ask for the matrix pipe at the same time and for the
  // vector ALU at the same time; so the second half of the workgroup runs ONE HALF-TILE BEHIND: for waves 4..7 the slab's
  // barrier falls between H1 and H2 of the slab's last tile (S^T and the V^T fragments stay in registers across it: every
  // LDS read of the slab has been issued, and is waited for, in front of the barrier) -- right behind a barrier waves 0..3
  // start with H1 (matrix) while waves 4..7 run the H2 they carried over (vector first), and the halves stay complementary.
  struct TileInfo32 { int rowA, rowB, pl; bool a0, a1, b0, b1; };
  TileInfo32 pend = {0, 0, 0, false, false, false, false};   // the empty tile: everything masked, contributes nothing
  const bool lag = wave >= NW / 2 && !(G.variant & 32);    // (variant bit 5: no stagger, for A/B timing)
  f32x16 sc;                                               // S^T of the tile between its H1 and H2
#pragma unroll
  for (int r = 0; r < 16; ++r) sc[r] = 0.f;
  s16x4 xv[MB][4];                                         // its V^T fragments: [dh block][row A: keys 4hi.., 8+4hi..; row B: same]
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[mb][i] = (s16x4)(0);

  auto h1 = [&](unsigned ka_addr, unsigned va) {
    // all K fragments of the tile are requested up front and retired one by one with counted waits (left to itself hipcc
    // requests two, drains the queue, issues two MFMAs, ...: four exposed LDS round trips per tile).  Inline asm keeps the
    // order; each wait names the fragment it releases, so the MFMA that consumes it cannot move above the wait.
    s16x8 ka[KS];
    static_for<KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      ka[ks] = ds_read_b128_asm<ks * 32>(ka_addr);
    });
    static_for<KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      lgkm_wait_for<KS - 1 - ks>(ka[ks]);
      __builtin_amdgcn_sched_barrier(0);                   // (fences the MFMA below its wait: cdna guide 5.7 / rule 18)
      if (!(G.dbg & 32)) {
        if constexpr (ks == 0) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks].v, (f32x16)(0.f), 0, 0, 0);
        else sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks].v, sc, 0, 0, 0);
      }
    });
    static_for<MB>([&](auto mbc) {
      constexpr int mb = decltype(mbc)::value;
      xv[mb][0] = ds_read_tr16_asm<mb * 64>(va);
      xv[mb][1] = ds_read_tr16_asm<mb * 64 + 8 * I::VROW>(va);
      xv[mb][2] = ds_read_tr16_asm<mb * 64 + 16 * I::VROW>(va);
      xv[mb][3] = ds_read_tr16_asm<mb * 64 + 24 * I::VROW>(va);
    });
  };

  auto h2 = [&](const TileInfo32& ti) {
    if (G.dbg & 4) { ds_tr_wait(); return; }               // (timing ablation: no softmax, no PV)
    if constexpr (PROBE) {
      const int kw = (2 * G.eW + 1), kh = (2 * G.eH + 1);
      const int hq = hA + jq;
      const long qn = plane_o + hq * 16 + wq;
      const int nk = (2 * G.eS + 1) * kh * kw;
      float* row = DBG + (qn * G.heads + head) * nk;
      const int ds = (sk_lo + ti.pl) - s;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int wk = (r & 3) + 8 * ((r >> 2) & 1) + 4 * hi;
        const int dw = wk - wq;
        const int krow = r < 8 ? ti.rowA : ti.rowB;
        const bool ok = r < 8 ? (jq ? ti.a1 : ti.a0) : (jq ? ti.b1 : ti.b0);
        if (ok && dw <= G.eW && -dw <= G.eW) row[((ds + G.eS) * kh + (krow - hq + G.eH)) * kw + dw + G.eW] = sc[r] * G.scale;
      }
    }
    // exponent of 2 by one fma per logit (column bias and running max in the addend); deferred running max
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(sc[r], c2, bm[r & 7]);
    if (!(ti.a0 && ti.a1 && ti.b0 && ti.b1)) {             // wave-uniform: some 16 x 16 sub-block of the tile is out of window
      const bool va_ok = jq ? ti.a1 : ti.a0, vb_ok = jq ? ti.b1 : ti.b0;
#pragma unroll
      for (int r = 0; r < 8; ++r) t[r] = va_ok ? t[r] : -INFINITY;
#pragma unroll
      for (int r = 8; r < 16; ++r) t[r] = vb_ok ? t[r] : -INFINITY;
    }
    float mx = __builtin_fmaxf(__builtin_fmaxf(t[0], t[1]), t[2]);
#pragma unroll
    for (int r = 3; r + 1 < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, t[r]), t[r + 1]);
    mx = __builtin_fmaxf(mx, t[15]);
    if (__any(__builtin_fmaxf(mx, ub) > DEFER)) {
      // rare branch: move the reference max by the query's maximum over its two lanes
      const float gm = pair_max(mx);
      float sub, alpha;
      if (ub > 0.f) {                                      // query not initialised yet: adopt its first finite maximum
        const bool got = gm > -INFINITY;
        sub = got ? gm : 0.f;
        alpha = 1.f;                                       // (O^T and l are still zero)
        m_run = sub;
        ub = got ? -INFINITY : INFINITY;
      } else {
        sub = fmaxf(gm, 0.f);                              // the reference only moves up
        alpha = __builtin_amdgcn_exp2f(-sub);
        m_run += sub;
      }
      l_run *= alpha;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mb][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] -= sub;
#pragma unroll
      for (int r = 0; r < 8; ++r) bm[r] -= sub;            // (-inf stays -inf)
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = __builtin_amdgcn_exp2f(t[r]);
    l_run += (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
             (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
    Frag8<bf16_t> pa, pb;                                  // P^T of the tile's first / second key row as PV's B operands
#pragma unroll
    for (int r = 0; r < 8; ++r) { pa.v[r] = (short)f32_to_bf16_bits(t[r]); pb.v[r] = (short)f32_to_bf16_bits(t[8 + r]); }
    ds_tr_wait();
    if (G.dbg & 8) return;                                 // (timing ablation: no PV)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      asm volatile("" : "+v"(xv[mb][0]), "+v"(xv[mb][1]), "+v"(xv[mb][2]), "+v"(xv[mb][3]));   // uses stay behind the wait
      Frag8<bf16_t> v0, v1;
      v0.v = __builtin_shufflevector(xv[mb][0], xv[mb][1], 0, 1, 2, 3, 4, 5, 6, 7);
      v1.v = __builtin_shufflevector(xv[mb][2], xv[mb][3], 0, 1, 2, 3, 4, 5, 6, 7);
      o[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0.v, pa.v, o[mb], 0, 0, 0);
      o[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1.v, pb.v, o[mb], 0, 0, 0);
    }
  };

  // two copies of the slab loop (the wave-uniform `lag` decides once): simple control flow inside each
  auto run = [&](auto lagc) {
    constexpr bool LAG = decltype(lagc)::value;
    for (int j = 0; j < nslab; ++j) {
      const char* Sb = smem + (j & (NBUF - 1)) * I::BUF;
      const int pl = pl_c, base = base_c;
      WMZ_TS32(16 * j + 0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of slab j have landed; its LDS reads of slab j-1 are done
      WMZ_TS32(16 * j + 1);
      __builtin_amdgcn_s_barrier();                        // ... everyone's: slab j is complete and slab j - 1's buffer is free
      WMZ_TS32(16 * j + 2);
      if (j + 1 < nslab && staging) issue_slab();
      WMZ_TS32(16 * j + 3);
      int tsi = 0;
      // slab rows this wave needs: plane rows my_lo .. my_hi of this slab's parity
      const int lo = max(0, (my_lo - base + RS - 1) >> LOG_RS), hi_r = min(KC - 1, (my_hi - base) >> LOG_RS);
      if (act && !(G.dbg & 1)) {
        for (int t0 = lo; t0 <= hi_r; t0 += 2) {
          const int ta = min(t0, KC - 2);                  // the tile's slab rows ta, ta + 1 (shifted down at the slab's end)
          TileInfo32 cur;
          cur.rowA = base + RS * ta; cur.rowB = cur.rowA + RS; cur.pl = pl;
          const bool okA = ta >= t0, okB = ta + 1 <= hi_r;
          cur.a0 = okA && cur.rowA >= hA - G.eH && cur.rowA <= hA + G.eH; cur.a1 = okA && cur.rowA >= hB - G.eH && cur.rowA <= hB + G.eH;
          cur.b0 = okB && cur.rowB >= hA - G.eH && cur.rowB <= hA + G.eH; cur.b1 = okB && cur.rowB >= hB - G.eH && cur.rowB <= hB + G.eH;
          WMZ_TS32(16 * j + 4 + 4 * tsi);
          if constexpr (LAG) h2(pend);                     // the tile carried over (the very first one is the empty tile: all masked)
          WMZ_TS32(16 * j + 5 + 4 * tsi);
          h1(lds_addr(Sb + kbase + ta * 16 * I::KROW), lds_addr(Sb + vbase + ta * 16 * I::VROW));
          WMZ_TS32(16 * j + 6 + 4 * tsi);
          if constexpr (LAG) pend = cur; else h2(cur);
          WMZ_TS32(16 * j + 7 + 4 * tsi);
          ++tsi;
        }
      }
      base_c = base_n; pl_c = pl_n;
      advance();
      next_state();
    }
    if constexpr (LAG) { if (act && !(G.dbg & 1)) h2(pend); }   // drain
    WMZ_TS32(250);
  };
  WMZ_TS32(240);
  if (lag) run(std::true_type{}); else run(std::false_type{});

  if (act && !(G.dbg & 64)) {                            // (dbg 64: timing ablation, no epilogue)
    const float l = wave_halves_sum(l_run);
    const float inv = 1.f / l;
    bf16_t* orow = O + (plane_o + hA * 16 + q32) * G.ldo + (long)head * DH;
    // lane (q, hi) holds dh 32 mb + 8 gg + 4 hi .. + 3 in registers 4 gg .. 4 gg + 3; the query's two lanes swap halves of a
    // (gg, gg + 1) pair so that hi = 0 owns dh 32 mb + 16 pp .. + 7 and hi = 1 the following 8: one 16-byte store each
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int ra = 8 * pp, rb = 8 * pp + 4;            // registers of groups gg = 2 pp (dh + 0..3 | + 4..7 by hi) and 2 pp + 1
        unsigned a0 = (unsigned)f32_to_bf16_bits(o[mb][ra] * inv) | ((unsigned)f32_to_bf16_bits(o[mb][ra + 1] * inv) << 16);
        unsigned a1 = (unsigned)f32_to_bf16_bits(o[mb][ra + 2] * inv) | ((unsigned)f32_to_bf16_bits(o[mb][ra + 3] * inv) << 16);
        unsigned b0 = (unsigned)f32_to_bf16_bits(o[mb][rb] * inv) | ((unsigned)f32_to_bf16_bits(o[mb][rb + 1] * inv) << 16);
        unsigned b1 = (unsigned)f32_to_bf16_bits(o[mb][rb + 2] * inv) | ((unsigned)f32_to_bf16_bits(o[mb][rb + 3] * inv) << 16);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);     // upper half of a <-> lower half of b
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        i32x4 pk;
        pk[0] = (int)s0[0]; pk[1] = (int)s1[0]; pk[2] = (int)s0[1]; pk[3] = (int)s1[1];
        *reinterpret_cast<i32x4*>(orow + mb * 32 + 16 * pp + 8 * hi) = pk;
      }
    }
    if (LSE != nullptr && hi == 0) LSE[(plane_o + hA * 16 + q32) * G.heads + head] = m_run * 0.6931471805599453f + logf(l);
  }
}

template <int DH, bool ALIGNED, bool PROBE>
int launch_row32(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg, AttnGeom G, hipStream_t st) {
  G.qgroups = wmz_cdiv(G.H, QROWS);
  const long nwg = (long)G.B * G.heads * G.Sq * G.qgroups;
  hipLaunchKernelGGL((attn_fwd_row32_kernel<DH, ALIGNED, PROBE>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, lse, dbg, G);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_fwd(row32)");
  return WMZ_OK;
}

template <bool ALIGNED, bool PROBE>
int by_dh(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg, const AttnGeom& G, hipStream_t st) {
  if (G.dh == 128) return launch_row32<128, ALIGNED, PROBE>(q, k, v, out, lse, dbg, G, st);
  if (G.dh == 64) return launch_row32<64, ALIGNED, PROBE>(q, k, v, out, lse, dbg, G, st);
  return launch_row32<32, ALIGNED, PROBE>(q, k, v, out, lse, dbg, G, st);
}

}  // namespace

// Called by wmz_local3d_attn_fwd when the shape qualifies (bf16, W == 16, H even, dim_head in {32,64,128}).  dbg: optional
// logits probe [N, heads, window] (natural-log-domain scaled logits of the in-window slots, pre-filled with -1e9).
int wmz_attn_fwd_row32_dispatch(const void* q, const void* k, const void* v, void* out, float* lse, float* dbg,
                                const AttnGeom& G, hipStream_t st) {
  const bool aligned = (G.H & 15) == 0;
  if (dbg != nullptr) return aligned ? by_dh<true, true>(q, k, v, out, lse, dbg, G, st) : by_dh<false, true>(q, k, v, out, lse, dbg, G, st);
  return aligned ? by_dh<true, false>(q, k, v, out, lse, nullptr, G, st) : by_dh<false, false>(q, k, v, out, lse, nullptr, G, st);
}
