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
// One 16-query row per wave, 16 waves per workgroup (four per SIMD): the S -> max -> exp -> PV chain is latency-bound.
#include "attn_common.h"

namespace {

constexpr int NW = 16;
constexpr int KC = 8;                 // key rows per slab
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
template <int DH, int ROWP, int IMGB>
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

template <int DH>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_fwd_row16_kernel(const bf16_t* __restrict__ Q,
                                                                         const bf16_t* __restrict__ K,
                                                                         const bf16_t* __restrict__ V,
                                                                         bf16_t* __restrict__ O, float* __restrict__ LSE,
                                                                         AttnGeom G) {
  using I = Img<DH>;
  constexpr int KS = DH / 32, MT = DH / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;

  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int og = lid % G.qgroups; lid /= G.qgroups;
  const int sq = lid % G.Sq; lid /= G.Sq;
  const int s = G.qs0 + sq;
  const int head = lid % G.heads;
  const int b = lid / G.heads;

  const int HW = G.HW, H = G.H;
  const int h = og * NW + wave;                       // this wave's query row
  const bool active = h < H;
  const long plane_q = ((long)b * G.S + s) * HW;
  const long plane_o = ((long)b * G.Sq + sq) * HW;
  const float c2 = G.scale * 1.4426950408889634f;

  // column-window biases of this lane's 4 keys (w = 4g + r) against its query (w = li)
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int d = 4 * g + r - li; bias[r] = (d <= G.eW && -d <= G.eW) ? 0.f : -INFINITY; }

  Frag8<bf16_t> qf[KS];
  {
    const bf16_t* qrow = Q + (plane_q + (active ? h : 0) * 16 + li) * G.ldq + (long)head * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_zero(qf[ks]);
      if (active) frag_load(qf[ks], qrow + ks * 32 + g * 8);
    }
  }
  f32x4 o[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) o[mt] = (f32x4)(0.f);
  float m_run = -1e30f, l_run = 0.f;                  // m_run in log2 units of the scaled logits

  const int kbase = li * I::KROW + g * 16;
  const int vbase = (4 * g + (li >> 2)) * I::VROW + (li & 3) * 8;
  const int my_lo = max(h - G.eH, 0), my_hi = min(h + G.eH, H - 1);
  // key rows the workgroup stages
  const int h0 = og * NW;
  const int t_lo = max(h0 - G.eH, 0), t_hi = min(min(h0 + NW - 1, H - 1) + G.eH, H - 1);
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
    stage_padded<DH, I::KROW, I::KIMG>(buf, K + plane_k * G.ldk + (long)head * DH, G.ldk, base, H, wave, lane);
    stage_padded<DH, I::VROW, I::VIMG>(buf + I::KIMG, V + plane_k * G.ldv + (long)head * DH, G.ldv, base, H, wave, lane);
  };
  // pieces this wave issues per slab (K image + V image): the counted vmcnt below depends on it
  constexpr int KP = I::KIMG / 1024, VP = I::VIMG / 1024;
  const int my_pieces = (KP - wave + NW - 1) / NW + (VP - wave + NW - 1) / NW;
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j)
    if (j < nslab) issue(j);
  for (int j = 0; j < nslab; ++j) {
    const int pl = j / nch, rem = j - pl * nch;
    const int base = ((c_first + (rem >> 1)) << 4) + (rem & 1);   // plane row of slab row 0; slab row r <-> base + 2r
    const char* Ks = smem + (j % NBUF) * I::BUF;
    const char* Vs = Ks + I::KIMG;
    // this wave's pieces of slab j landed: all but the pieces of the (up to NBUF-2) younger slabs in flight
    {
      const int younger = min(NBUF - 2, nslab - 1 - j) * my_pieces;
      if (younger >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (younger == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // ... everyone's did, and slab j-1 is retired: refill its slot
    if (j + NBUF - 1 < nslab) issue(j + NBUF - 1);
    if (!active || (G.dbg & 1)) continue;
    const int lo = max(0, (my_lo - base + 1) >> 1), hi = min(KC - 1, (my_hi - base) >> 1);   // slab rows this wave needs
    for (int t0 = lo; t0 <= hi; t0 += 2) {
      const bool has1 = t0 + 1 <= hi;
      const int ko0 = kbase + t0 * 16 * I::KROW, ko1 = has1 ? ko0 + 16 * I::KROW : ko0;
      const int vo0 = vbase + t0 * 16 * I::VROW, vo1 = has1 ? vo0 + 16 * I::VROW : vo0;
      // ---- S^T = K Q^T, two key rows
      f32x4 s0 = (f32x4)(0.f), s1 = (f32x4)(0.f);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag8<bf16_t> ka, kb;
        ka.v = *reinterpret_cast<const s16x8*>(Ks + ko0 + ks * 64);
        kb.v = *reinterpret_cast<const s16x8*>(Ks + ko1 + ks * 64);
        mma16(s0, ka, qf[ks]);
        mma16(s1, kb, qf[ks]);
      }
      // ---- log2-domain logits with the column window folded in
      float t[8];
      const float b1 = has1 ? 0.f : -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t[r] = fmaf(s0[r], c2, bias[r]);
        t[4 + r] = fmaf(s1[r], c2, bias[r] + b1);
      }
      float mx = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
      mx = wave_xor_max(mx, 16);
      mx = wave_xor_max(mx, 32);
      // ---- deferred max: rescale only when some row's max grew by more than 2^DEFER (wave-uniform decision)
      if (__any(mx > m_run + DEFER)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o[mt] *= alpha;
      }
      float p[8];
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        p[r] = __builtin_amdgcn_exp2f(t[r] - m_run);
        psum += p[r];
      }
      l_run += psum;
      Frag8<bf16_t> pf;
      frag_from_f32<bf16_t>(pf, p);
      // ---- O^T += V^T P^T
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Vs + vo0 + mt * 32));
        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Vs + vo1 + mt * 32));
        Frag8<bf16_t> vf;
        vf.v = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        mma16(o[mt], vf, pf);
      }
    }
  }

  if (!active) return;
  float l = l_run;
  l = wave_xor_add(l, 16);
  l = wave_xor_add(l, 32);
  const float inv = 1.f / l;
  bf16_t* orow = O + (plane_o + h * 16 + li) * G.ldo + (long)head * DH;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    s16x4 pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_bits(o[mt][r] * inv);
    *reinterpret_cast<s16x4*>(orow + mt * 16 + 4 * g) = pk;
  }
  if (LSE != nullptr && g == 0) LSE[(plane_o + h * 16 + li) * G.heads + head] = m_run * 0.6931471805599453f + logf(l);
}

template <int DH>
int launch_row16(const void* q, const void* k, const void* v, void* out, float* lse, AttnGeom G, hipStream_t st) {
  G.qgroups = wmz_cdiv(G.H, NW);
  const long nwg = (long)G.B * G.heads * G.Sq * G.qgroups;
  const size_t smem = NBUF * (size_t)Img<DH>::BUF;
  auto kern = attn_fwd_row16_kernel<DH>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), smem, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                     (bf16_t*)out, lse, G);
  WMZ_LAUNCH_CHECK("wmz_local3d_attn_fwd(row16)");
  return WMZ_OK;
}

}  // namespace

// Called by wmz_local3d_attn_fwd when the shape qualifies (bf16, W == 16, dim_head in {32,64,128}, no logits probe).
int wmz_attn_fwd_row16_dispatch(const void* q, const void* k, const void* v, void* out, float* lse, const AttnGeom& G,
                                hipStream_t st) {
  if (G.dh == 128) return launch_row16<128>(q, k, v, out, lse, G, st);
  if (G.dh == 64) return launch_row16<64>(q, k, v, out, lse, G, st);
  return launch_row16<32>(q, k, v, out, lse, G, st);
}
