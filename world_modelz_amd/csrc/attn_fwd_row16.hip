// Local windowed 3D attention forward, fast path for the reference's canonical latent planes: W == 16 (a 16-query tile is
// exactly one row of the plane), bf16, dim_head in {32, 64, 128}.  Same algorithm and outputs as attn_fwd.hip (which stays
// the general / fp32 / parity path); this version exists because that kernel is VALU-bound (~400 instructions per
// 32-key step against 16 MFMAs):
//   * the row test (|hk-hq| <= eH, plane bounds) is decided per key ROW by the loop bounds, and the column test
//     (|wk-wq| <= eW) is the same for every key row: four additive biases (0 / -inf) per lane, folded into the FMA that
//     moves the logit to the log2 domain -- no coordinate tables, no per-element tests;
//   * K / V slabs live in PADDED LDS rows (K: +16 B, V: +32 B) instead of XOR-swizzled ones, so every fragment address is
//     `lane base + row offset + immediate` (4 address adds per step) and still bank-conflict-free for ds_read_b128 and
//     ds_read_b64_tr_b16; slabs of 8 key rows arrive by LDS-DMA, double-buffered (counted vmcnt; padding is applied on the
//     source-address side);
//   * the online-softmax rescale of O^T is skipped unless some row's max grew by more than 2^8 (deferred max);
//   * raw v_exp_f32 (arguments are <= 8 by construction, underflow to 0 is the masked case).
// Two 16-query rows per wave, 8 waves per workgroup: every K / V fragment read from LDS is used by both rows (with one row
// per wave the LDS array, not the MFMA or the VALU, set the pace).
#include "attn_common.h"
#include <stdlib.h>

namespace {

long long* g_attn_ts = nullptr;     // timing probe buffer (16 waves x 64 int64), see wmz_debug_attn_timestamps

// QT = query rows per wave, NW = 16 / QT waves per workgroup (a workgroup covers 16 query rows of one plane).  With
// QT = 2 every K / V fragment read from LDS feeds two MFMAs (half the LDS traffic), at half the waves per SIMD.
constexpr int KC = 8;                 // key rows per slab
// (Issuing the slab DMA from four loader waves only, so that the other twelve compute meanwhile, measured 1.6x SLOWER:
// LDS-DMA writes landing during the fragment reads cost more than the ~1k cycles of issue they hide.)
constexpr int NBUF = 2;               // LDS slab ring: NBUF-1 slabs in flight (4 x 4-row slabs measured slower: 68 vs 54 us)
constexpr float DEFER = 8.f;          // log2 units

template <int DH> struct Img {
  static constexpr int KROW = DH * 2 + 16, VROW = DH * 2 + 32;
  static constexpr int KIMG = KC * 16 * KROW, VIMG = KC * 16 * VROW;
  static constexpr int BUF = KIMG + VIMG;
  static_assert(KIMG % 1024 == 0 && VIMG % 1024 == 0, "images must be whole 1 KB DMA pieces");
};

// LDS-DMA one padded image: lane landing on (row, 16-byte chunk) fetches that chunk of global row c0*16+row; pad chunks and
// rows past the valid range fetch a valid dummy (never read / masked).
// A slab holds every OTHER row of a 16-row chunk of the plane (plane row = base + 2 * slab row): whatever its own row, a
// wave finds about half of its +-eH key rows in each slab, so all waves of the workgroup are busy between two barriers.
template <int DH, int ROWP, int IMGB, int NW>
__device__ __forceinline__ void stage_padded(char* dst, const bf16_t* plane, long ld, int base, int H, int wave,
                                             int lane) {
  constexpr int PIECES = IMGB / 1024;
#pragma unroll
  for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
    const int piece = wave + NW * i;
    if (piece >= PIECES) break;                       // wave-uniform
    const int off = piece * 1024 + lane * 16;
    const int r = off / ROWP;
    int c = (off - r * ROWP) >> 4;
    c = c < DH / 8 ? c : 0;
    const int prow = min(base + 2 * (r >> 4), H - 1);         // rows past the plane: any valid row (never read)
    const int grow = (prow << 4) + (r & 15);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(plane + (long)grow * ld + c * 8),
                                     (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
  }
}

template <int DH, int QT>
__global__ __launch_bounds__(1024 / QT, 1) void attn_fwd_row16_kernel(const bf16_t* __restrict__ Q,
                                                                    const bf16_t* __restrict__ K,
                                                                    const bf16_t* __restrict__ V,
                                                                    bf16_t* __restrict__ O, float* __restrict__ LSE,
                                                                    AttnGeom G, long long* ts) {
  using I = Img<DH>;
  constexpr int KS = DH / 32, MT = DH / 16, NW = 16 / QT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  // timing probe (wmz_debug_attn_timestamps): workgroup 0, per wave, s_memtime at the phase boundaries
#define WMZ_ATS(slot) do { if (ts != nullptr && blockIdx.x == 0 && lane == 0) ts[wave * 64 + (slot)] = __builtin_readcyclecounter(); } while (0)
  WMZ_ATS(0);

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int sq = lid % G.Sq; lid /= G.Sq;
  const int s = G.qs0 + sq;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  const int HW = G.HW, H = G.H;
  const int hq0 = og * (NW * QT) + wave * QT;           // this wave's first query row; it owns rows hq0 .. hq0+QT-1
  const long plane_q = ((long)b * G.S + s) * HW;
  const long plane_o = ((long)b * G.Sq + sq) * HW;
  const float c2 = G.scale * 1.4426950408889634f;

  // column-window biases of this lane's 4 keys (w = 4g + r) against its query (w = li)
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int d = 4 * g + r - li; bias[r] = (d <= G.eW && -d <= G.eW) ? 0.f : -INFINITY; }

  bool act[QT];
  Frag8<bf16_t> qf[QT][KS];
  f32x4 o[QT][MT];
  float m_run[QT], l_run[QT];                            // m_run in log2 units of the scaled logits
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    act[q] = hq0 + q < H;
    const bf16_t* qrow = Q + (plane_q + (act[q] ? hq0 + q : 0) * 16 + li) * G.ldq + (long)head * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_zero(qf[q][ks]);
      if (act[q]) frag_load(qf[q][ks], qrow + ks * 32 + g * 8);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) o[q][mt] = (f32x4)(0.f);
    m_run[q] = -1e30f;
    l_run[q] = 0.f;
  }

  const int kbase = li * I::KROW + g * 16;
  const int vbase = (4 * g + (li >> 2)) * I::VROW + (li & 3) * 8;
  const int hq1 = min(hq0 + QT - 1, H - 1);             // last query row of the wave
  const int my_lo = max(hq0 - G.eH, 0), my_hi = min(hq1 + G.eH, H - 1);
  // key rows the workgroup stages
  const int h0 = og * (NW * QT);
  const int t_lo = max(h0 - G.eH, 0), t_hi = min(min(h0 + NW * QT - 1, H - 1) + G.eH, H - 1);
  const int sk_lo = max(0, s - G.eS), sk_hi = min(G.S - 1, s + G.eS);
  const int c_first = t_lo >> 4, c_last = t_hi >> 4;
  const int nch = (c_last - c_first + 1) * 2;               // slabs per key plane: (16-row chunk) x (row parity)
  const int nslab = (sk_hi - sk_lo + 1) * nch;

  auto issue = [&](int j) {
    if (G.dbg & 2) return;
    const int pl = j / nch, rem = j - pl * nch;
    const int base = ((c_first + (rem >> 1)) << 4) + (rem & 1);
    const long plane_k = ((long)b * G.S + (sk_lo + pl)) * HW;
    char* buf = smem + (j % NBUF) * I::BUF;
    stage_padded<DH, I::KROW, I::KIMG, NW>(buf, K + plane_k * G.ldk + (long)head * DH, G.ldk, base, H, wave, lane);
    stage_padded<DH, I::VROW, I::VIMG, NW>(buf + I::KIMG, V + plane_k * G.ldv + (long)head * DH, G.ldv, base, H, wave, lane);
  };
  static_assert(NBUF == 2, "one slab in flight: the wait below is vmcnt(0)");
  WMZ_ATS(1);
  if (nslab > 0) issue(0);
  WMZ_ATS(2);
  for (int j = 0; j < nslab; ++j) {
    const int pl = j / nch, rem = j - pl * nch;
    const int base = ((c_first + (rem >> 1)) << 4) + (rem & 1);   // plane row of slab row 0; slab row r <-> base + 2r
    const char* Ks = smem + (j % NBUF) * I::BUF;
    const char* Vs = Ks + I::KIMG;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of slab j landed ...
    if (j < 15) WMZ_ATS(3 + 4 * j);
    __builtin_amdgcn_s_barrier();                        // ... everyone's did, and slab j-1 is retired: refill its slot
    if (j < 15) WMZ_ATS(4 + 4 * j);
    if (j + 1 < nslab) issue(j + 1);
    if (j < 15) WMZ_ATS(5 + 4 * j);
    if (!act[0] || (G.dbg & 1)) continue;
    const int lo = max(0, (my_lo - base + 1) >> 1), hi = min(KC - 1, (my_hi - base) >> 1);   // slab rows this wave needs
    for (int t0 = lo; t0 <= hi; t0 += 2) {
      const bool has1 = t0 + 1 <= hi;
      const int ko0 = kbase + t0 * 16 * I::KROW, ko1 = has1 ? ko0 + 16 * I::KROW : ko0;
      const int vo0 = vbase + t0 * 16 * I::VROW, vo1 = has1 ? vo0 + 16 * I::VROW : vo0;
      const int pr0 = base + 2 * t0, pr1 = pr0 + 2;      // plane rows of the two key rows
      // ---- S^T = K Q^T: two key rows against the wave's QT query rows (every K fragment feeds QT MFMAs)
      f32x4 sc[QT][2];
#pragma unroll
      for (int q = 0; q < QT; ++q) { sc[q][0] = (f32x4)(0.f); sc[q][1] = (f32x4)(0.f); }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag8<bf16_t> ka, kb;
        ka.v = *reinterpret_cast<const s16x8*>(Ks + ko0 + ks * 64);
        kb.v = *reinterpret_cast<const s16x8*>(Ks + ko1 + ks * 64);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          mma16(sc[q][0], ka, qf[q][ks]);
          mma16(sc[q][1], kb, qf[q][ks]);
        }
      }
      // ---- per query row: log2-domain logits (column window + key-row window as additive 0 / -inf), online softmax
      Frag8<bf16_t> pf[QT];
      bool live[QT];
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        const int hq = hq0 + q;
        const bool v0 = act[q] && pr0 - hq <= G.eH && hq - pr0 <= G.eH;
        const bool v1 = act[q] && has1 && pr1 - hq <= G.eH && hq - pr1 <= G.eH;
        live[q] = v0 || v1;                              // wave-uniform
        frag_zero(pf[q]);
        if (!live[q]) continue;
        const float rb0 = v0 ? 0.f : -INFINITY, rb1 = v1 ? 0.f : -INFINITY;
        float t[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r] = fmaf(sc[q][0][r], c2, bias[r] + rb0);
          t[4 + r] = fmaf(sc[q][1][r], c2, bias[r] + rb1);
        }
        float mx = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
        mx = wave_groups_max(mx);
        // deferred max: rescale only when some row's max grew by more than 2^DEFER (wave-uniform decision)
        if (__any(mx > m_run[q] + DEFER)) {
          const float m_new = fmaxf(m_run[q], mx);
          const float alpha = __builtin_amdgcn_exp2f(m_run[q] - m_new);
          m_run[q] = m_new;
          l_run[q] *= alpha;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) o[q][mt] *= alpha;
        }
        float p[8];
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          p[r] = __builtin_amdgcn_exp2f(t[r] - m_run[q]);
          psum += p[r];
        }
        l_run[q] += psum;
        frag_from_f32<bf16_t>(pf[q], p);
      }
      // ---- O^T += V^T P^T   (transposed V fragments by asm reads: see ds_read_tr16_asm; each feeds QT MFMAs)
      {
        const unsigned va0 = lds_addr(Vs + vo0), va1 = lds_addr(Vs + vo1);
        s16x4 x0[MT], x1[MT];
        static_for<MT>([&](auto mt) {
          x0[mt] = ds_read_tr16_asm<mt * 32>(va0);
          x1[mt] = ds_read_tr16_asm<mt * 32>(va1);
        });
        ds_tr_wait();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          asm volatile("" : "+v"(x0[mt]), "+v"(x1[mt]));       // uses stay behind the wait
          Frag8<bf16_t> vf;
          vf.v = __builtin_shufflevector(x0[mt], x1[mt], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int q = 0; q < QT; ++q)
            if (live[q]) mma16(o[q][mt], vf, pf[q]);
        }
      }
    }
    if (j < 15) WMZ_ATS(6 + 4 * j);
  }
  WMZ_ATS(63);

#pragma unroll
  for (int q = 0; q < QT; ++q) {
    if (!act[q]) continue;
    const int h = hq0 + q;
    float l = l_run[q];
    l = wave_groups_sum(l);
    const float inv = 1.f / l;
    bf16_t* orow = O + (plane_o + h * 16 + li) * G.ldo + (long)head * DH;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      s16x4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(o[q][mt][r] * inv);
      *reinterpret_cast<s16x4*>(orow + mt * 16 + 4 * g) = pk;
    }
    if (LSE != nullptr && g == 0) LSE[(plane_o + h * 16 + li) * G.heads + head] = m_run[q] * 0.6931471805599453f + logf(l);
  }
  WMZ_ATS(62);
#undef WMZ_ATS
}

template <int DH, int QT>
int launch_row16(const void* q, const void* k, const void* v, void* out, float* lse, AttnGeom G, hipStream_t st) {
  constexpr int NW = 16 / QT;
  G.qgroups = wmz_cdiv(G.H, 16);
  const long nwg = (long)G.B * G.heads * G.Sq * G.qgroups;
  const size_t smem = NBUF * (size_t)Img<DH>::BUF;
  auto kern = attn_fwd_row16_kernel<DH, QT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), smem, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                     (bf16_t*)out, lse, G, g_attn_ts);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_fwd(row16)");
  return WMZ_OK;
}

}  // namespace

// Called by wmz_local3d_attn_fwd when the shape qualifies (bf16, W == 16, dim_head in {32,64,128}, no logits probe).
extern "C" int wmz_debug_attn_timestamps(void* buf) { g_attn_ts = (long long*)buf; return WMZ_OK; }

int wmz_attn_fwd_row16_dispatch(const void* q, const void* k, const void* v, void* out, float* lse, const AttnGeom& G,
                                hipStream_t st) {
  static const int qt = getenv("WMZ_ATTN_QT") ? atoi(getenv("WMZ_ATTN_QT")) : 1;     // A/B timing switch
  if (qt == 2) {
    if (G.dh == 128) return launch_row16<128, 2>(q, k, v, out, lse, G, st);
    if (G.dh == 64) return launch_row16<64, 2>(q, k, v, out, lse, G, st);
    return launch_row16<32, 2>(q, k, v, out, lse, G, st);
  }
  if (G.dh == 128) return launch_row16<128, 1>(q, k, v, out, lse, G, st);
  if (G.dh == 64) return launch_row16<64, 1>(q, k, v, out, lse, G, st);
  return launch_row16<32, 1>(q, k, v, out, lse, G, st);
}
