// Per-token layer kernel for widths the register-chained default kernel (layer_fused.hip: 256 / 128 / 256 only) cannot hold:
// the reference's published runs (results/README.md: dim 96 / mlp 256 and dim 384 / mlp 512, one head of 128) and config 5's
// widths.  Same fusion -- everything between two attention launches is one launch (bf16 operands, fp32 accumulation):
//
//   HEAD:  x1 = o Wout^T + bout + x                       (to_out + residual,        local_3d_attention.py:50-53, :160)
//          x2 = W2 GELU(W1' LN(x1) + b1') + b2 + x1       (PreNorm(FeedForward) + x, :11-31, :161; LayerNorm affine folded
//                                                          into W1' / b1' by the host packer, as in layer_fused.hip)
//   TAIL:  q = Wq x2 ,  k | v = Wk' | Wv' LN(x2) + b      (NEXT layer's to_q on the RAW stream, to_k / to_v on LayerNorm(x):
//                                                          quirk Q1, :16-17, :46-48, :106-108)
//
// Design.  A workgroup = 8 waves = 128 tokens, a wave owns 16 tokens end to end; every GEMM is computed transposed,
// D^T[16 features x 16 tokens] += W[16 x 32] act^T[32 x 16] with MFMA 16x16x32 bf16, the token on the lane (l & 15), lane
// group g = l >> 4.  Feature order is FREE on both sides of every GEMM (the host permutes weight rows and columns), so every
// activation vector of width Wd lives "lane-group-major": lane (t, g) owns the CONTIGUOUS features g Wd/4 .. (g + 1) Wd/4 of
// its token -- as accumulator registers (block b, register r <-> feature g Wd/4 + 4 b + r) and, after bf16 packing, as the
// B operand of the next GEMM (k-step s, element j <-> feature g Wd/4 + 8 s + j) with NO exchange between lanes: activations
// never touch LDS, rows are read and written in 16-byte pieces, LayerNorm is lane-local sums plus one wave_groups_sum.
// The fp32 residual stream stays in registers (D/4 per lane: 96 at D = 384, 128 at D = 512 -- why a wave has 16 tokens, not
// 32 as in the default-width kernel, whose 32x32x16 tiles need D/2 per lane).
// Weights: the layer's stream (0.25 - 2.8 MB) is packed by the host as 1 KB pieces = one MFMA A operand each (lane-linear:
// conflict-free ds_read_b128), in consumption order, CSP = 32 pieces to a slab, every GEMM stage padded to whole slabs; an
// LDS ring of 3 slabs is filled by LDS-DMA (4 pieces per wave and slab), one counted vmcnt + one raw s_barrier per slab
// (= 32 MFMAs per wave).  The feed-forward is walked MC hidden units at a time (W1' rows -> GELU -> W2 columns).
#include "wmz_common.h"
#include "chain_widths.h"

#ifndef WMZ_CHAIN_GROUP
#define WMZ_CHAIN_GROUP 0         // this unit's group of width triples (chain_widths.h); group 0 also holds the entry points
#endif
// a group's launcher: WMZ_OK / an error code, or -1 when the widths are not this group's
#ifdef WMZ_OP16_F16
#define CHAIN_GROUP_FN(g) WMZ_CHAIN_CAT(WMZ_CHAIN_CAT(wmz_chain_fwd_group, g), _f16)
#else
#define CHAIN_GROUP_FN(g) WMZ_CHAIN_CAT(wmz_chain_fwd_group, g)
#endif
#define CHAIN_DECLARE_GROUP(g) \
  int CHAIN_GROUP_FN(g)(const void* params, int D, int I, int M, int head, int tail, int train, hipStream_t st);
WMZ_CHAIN_ALL_GROUPS(CHAIN_DECLARE_GROUP)

namespace {

constexpr int CW = 8;             // waves per workgroup
constexpr int CT = 16;            // tokens per wave
#ifndef WMZ_CHAIN_SP
#define WMZ_CHAIN_SP 32
#endif
#ifndef WMZ_CHAIN_RING
#define WMZ_CHAIN_RING 3
#endif
constexpr int CSP = WMZ_CHAIN_SP;           // 1 KB pieces per weight slab (= MFMAs per wave between two barriers)
constexpr int CSLAB = CSP * 1024;           // bytes per weight slab
constexpr int CRING = WMZ_CHAIN_RING;       // ring slots: CRING - 1 slabs of the stream in flight
constexpr int CPW = CSP / CW;               // pieces a wave requests per slab

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct ChainParams {
  const bf16_t* o;      // [B, n_q, HW, I]        attention output (HEAD)
  const bf16_t* x;      // [B, n_in, HW, D]       residual stream in (its trailing n_q planes per clip are used)
  bf16_t* xo;           // [B, n_q, HW, D]        residual stream out (HEAD)
  bf16_t* q;            // [B, n_q, HW, I]        (TAIL)
  bf16_t* kv;           // [2, B, n_q, HW, I]     k planes, then v planes (TAIL)
  const char* wpack;    // packed weights (+ CRING-1 slabs of padding: the prefetch runs past the last real slab)
  const float* vec;     // bout[D] b1'[M] b2[D] bk'[I] bv'[I]
  int B, n_q, n_in, HW;
  float eps;
  long ldkv, voff;      // kv row stride and offset of the v half in elements (inference: I and ntok I; training: 2 I and I)
  // TRAIN only: what the backward reads (all row-major over the launch's tokens)
  bf16_t* x1;           // [N, D]  the feed-forward block's input (raw rows), or NULL
  bf16_t* xn_ff;        // [N, D]  the normalised rows of x1 (no affine: it is folded into the weights)
  bf16_t* z;            // [N, M]  feed-forward pre-activation
  bf16_t* h;            // [N, M]  GELU(z)
  float* st_ff;         // [2, N]  mean | rstd of x1
  bf16_t* xn_attn;      // [N, D]  the normalised rows of x2 (what the next layer's folded to_k / to_v consume)
  float* st_attn;       // [2, N]  mean | rstd of x2
};

constexpr int pad16(int n) { return (n + CSP - 1) / CSP * CSP; }

template <int D, int I, int M, int MC, bool HEAD, bool TAIL>
struct ChainShape {
  static constexpr int P_OUT = pad16((D / 16) * (I / 32));                   // pieces: to_out
  static constexpr int P_FFC = (MC / 16) * (D / 32) + (D / 16) * (MC / 32);  // one feed-forward chunk (W1' rows, then W2 columns)
  static constexpr int P_QKV = pad16((I / 16) * (D / 32));                   // each of q, k, v
  static_assert(P_FFC % CSP == 0, "a feed-forward chunk must be whole slabs: MC * D % (256 CSP) == 0");
  static constexpr int SLABS = ((HEAD ? P_OUT + (M / MC) * P_FFC : 0) + (TAIL ? 3 * P_QKV : 0)) / CSP;
};

// TRAIN: the training forward (round 4): the same launch that also STORES what the step's backward reads -- the normalised rows
// (they ARE the packed B operands of the GEMMs behind the norms: no extra arithmetic), the feed-forward pre-activation, GELU of it,
// both LayerNorm statistics pairs and, on request, the raw feed-forward input.
template <int D, int I, int M, int MC, bool HEAD, bool TAIL, bool TRAIN>
__global__ __launch_bounds__(CW * 64, 2) void layer_chain_kernel(ChainParams P) {
  using S = ChainShape<D, I, M, MC, HEAD, TAIL>;
  constexpr int NBD = D / 16, NBI = I / 16, KSD = D / 32, KSI = I / 32, NBC = MC / 16, KSC = MC / 32;
  constexpr int VEC = 2 * D + M + 2 * I;
  __shared__ __attribute__((aligned(1024))) char ring[CRING * CSLAB];
  __shared__ __attribute__((aligned(16))) float vec[VEC];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const long ntok = (long)P.B * P.n_q * P.HW;
  long tok = (long)blockIdx.x * (CW * CT) + wave * CT + li;
  const bool ok = tok < ntok;
  if (!ok) tok = ntok - 1;                                  // (computed on the last token, never stored)
  const long per_clip = (long)P.n_q * P.HW;
  const long clip = tok / per_clip;
  const long tok_in = clip * ((long)P.n_in * P.HW) + (long)(P.n_in - P.n_q) * P.HW + (tok - clip * per_clip);

  // ---- weight ring: this wave's two pieces of every slab; the first CRING - 1 slabs are requested up front
  const char* wsrc = P.wpack + (wave * CPW) * 1024 + lane * 16;
  auto issue = [&](int slab) {
    char* dst = ring + (slab % CRING) * CSLAB + (wave * CPW) * 1024;
    const char* src = wsrc + (long)slab * CSLAB;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < CRING - 1; ++s) issue(s);
  for (int i = tid; i < VEC; i += CW * 64) vec[i] = P.vec[i];
  int slab = 0;                                             // slab being consumed (wave-uniform, runtime across the FF chunk loop)
  // before piece 0 of a slab: this wave's pieces of it have landed (only the CPW (CRING - 2) younger DMA requests -- and whatever
  // other VMEM traffic is younger still -- may be outstanding), then ONE barrier: every wave's pieces have, and every wave is
  // done with the slab before, whose slot is refilled right away
  auto acquire = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CPW * (CRING - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(slab + CRING - 1);                                // (the stream is padded: requests past the last real slab are harmless)
  };
  // A run of NP consecutive pieces of the stream (a GEMM stage, or a feed-forward chunk = two GEMMs sharing its slabs), the
  // first of which opens a slab: mf(piece index, A fragment) issues the piece's MFMA.  The fragments are read by inline asm
  // (left to hipcc, a plain LDS load with an LDS-DMA in flight drains vmcnt(0) first -- the ring would never run ahead -- and
  // every MFMA waits for its own read: one exposed LDS round trip per MFMA) through a window of PF fragments in flight inside a
  // slab, retired one by one with counted lgkmcnt waits.
  constexpr int PF = 4;
  auto run = [&](auto npc, auto&& mf) {
    constexpr int NP = decltype(npc)::value;
    s16x8 fr[PF];
    unsigned base = 0;
    static_for<NP>([&](auto pc) {
      constexpr int p = decltype(pc)::value, ps = p % CSP;
      constexpr int left = (CSP - 1 - ps) < (NP - 1 - p) ? (CSP - 1 - ps) : (NP - 1 - p);    // pieces behind p in this slab
      if constexpr (ps == 0) {
        slab += (p > 0);
        acquire();
        base = lds_addr(ring + (slab % CRING) * CSLAB) + lane * 16;
        static_for<(left + 1 < PF ? left + 1 : PF)>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          fr[(p + i) % PF] = ds_read_b128_asm<(ps + i) * 1024>(base);
        });
      }
      lgkm_wait_for<(left < PF - 1 ? left : PF - 1)>(fr[p % PF]);
      mf(pc, fr[p % PF]);
      if constexpr (left >= PF) fr[p % PF] = ds_read_b128_asm<(ps + PF) * 1024>(base);
    });
    slab += 1;                                              // the run's last (possibly padded) slab is done
  };
  // one GEMM stage: acc[b] += piece(ks * NB + b) x opnd[ks]
  auto gemm = [&](auto nbc, auto ksc, auto& acc, const auto& opnd) {
    constexpr int NB = decltype(nbc)::value, KS = decltype(ksc)::value;
    run(std::integral_constant<int, NB * KS>{}, [&](auto pc, const s16x8& a) {
      constexpr int p = decltype(pc)::value, ks = p / NB, b = p % NB;
      acc[b] = op16_mfma_16x16x32(a, opnd[ks], acc[b]);
    });
  };
  // fp32 accumulator blocks (lane-group-major) -> bf16 B operands of the next GEMM: k-step s = blocks 2 s, 2 s + 1
  auto pack = [&](auto nbc, const auto& acc, auto& opnd, float scale, float shift) {
    constexpr int NB = decltype(nbc)::value;
#pragma unroll
    for (int s = 0; s < NB / 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) opnd[s][j] = (short)f32_to_bf16_bits(fmaf(acc[2 * s + (j >> 2)][j & 3], scale, shift));
  };
  // 16-byte global accesses of a lane's contiguous quarter row (NB blocks = 4 NB features)
  auto load_rows = [&](auto nbc, const bf16_t* row, auto& acc, bool add) {
    constexpr int NB = decltype(nbc)::value;
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
      const i32x4 c = *reinterpret_cast<const i32x4*>(row + 8 * s);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = bf16_bits_to_f32((unsigned short)((unsigned)c[j] & 0xFFFFu)), hi = bf16_bits_to_f32((unsigned short)((unsigned)c[j] >> 16));   // (the unit's 16-bit format: wmz_common.h)
        const int e = 2 * j;
        if (add) { acc[2 * s + (e >> 2)][e & 3] += lo; acc[2 * s + ((e + 1) >> 2)][(e + 1) & 3] += hi; }
        else { acc[2 * s + (e >> 2)][e & 3] = lo; acc[2 * s + ((e + 1) >> 2)][(e + 1) & 3] = hi; }
      }
    }
  };
  auto store_rows = [&](auto nbc, bf16_t* row, const auto& acc) {
    constexpr int NB = decltype(nbc)::value;
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
      i32x4 c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 2 * j;
        c[j] = (int)((unsigned)f32_to_bf16_bits(acc[2 * s + (e >> 2)][e & 3]) |
                     ((unsigned)f32_to_bf16_bits(acc[2 * s + ((e + 1) >> 2)][(e + 1) & 3]) << 16));
      }
      if (ok) *reinterpret_cast<i32x4*>(row + 8 * s) = c;
    }
  };
  auto add_vec = [&](auto nbc, auto& acc, const float* v) {   // v: the vector's lane-group-major quarter of this lane, in LDS
    constexpr int NB = decltype(nbc)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] += *reinterpret_cast<const f32x4*>(v + 4 * b);
  };
  // LayerNorm statistics of the token's D features (D / 4 in this lane, the rest in lanes l ^ 16, ^ 32, ^ 48)
  auto ln_stats = [&](const f32x4 (&acc)[NBD], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NBD; ++b) s += (acc[b][0] + acc[b][1]) + (acc[b][2] + acc[b][3]);
    mean = wave_groups_sum(s) * (1.f / D);
    float v = 0.f;
#pragma unroll
    for (int b = 0; b < NBD; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[b][r] - mean; v = fmaf(d, d, v); }
    rstd = rsqrtf(wave_groups_sum(v) * (1.f / D) + P.eps);
  };
  using CNBD = std::integral_constant<int, NBD>;
  using CNBI = std::integral_constant<int, NBI>;
  using CNBC = std::integral_constant<int, NBC>;
  using CKSD = std::integral_constant<int, KSD>;
  using CKSI = std::integral_constant<int, KSI>;
  using CKSC = std::integral_constant<int, KSC>;
  f32x4 xr[NBD];                                            // the residual stream, fp32
  load_rows(CNBD{}, P.x + tok_in * D + g * (D / 4), xr, false);
  __syncthreads();                                          // vec is in LDS
  if constexpr (HEAD) {
    // ---- x1 = o Wout^T + bout + x
    s16x8 ob[KSI];
    {
      const bf16_t* orow = P.o + tok * I + g * (I / 4);
#pragma unroll
      for (int s = 0; s < KSI; ++s) ob[s] = *reinterpret_cast<const s16x8*>(orow + 8 * s);
    }
    add_vec(CNBD{}, xr, vec + g * (D / 4));
    gemm(CNBD{}, CKSI{}, xr, ob);
    // ---- x2 = W2 GELU(W1' LN(x1) + b1') + b2 + x1, MC hidden units at a time
    float mean, rstd;
    ln_stats(xr, mean, rstd);
    s16x8 xb[KSD];
    pack(CNBD{}, xr, xb, rstd, -mean * rstd);
    if constexpr (TRAIN) {
      // what the backward reads: the raw rows (op-by-op fallback only), the statistics, and the NORMALISED rows -- which are
      // exactly the B operand just packed: 16 bytes per k-step, no extra arithmetic
      if (P.x1 != nullptr) store_rows(CNBD{}, P.x1 + tok * D + g * (D / 4), xr);
      if (ok && g == 0) { P.st_ff[tok] = mean; P.st_ff[ntok + tok] = rstd; }
#pragma unroll
      for (int s2 = 0; s2 < KSD; ++s2)
        if (ok) *reinterpret_cast<s16x8*>(P.xn_ff + tok * D + g * (D / 4) + 8 * s2) = xb[s2];
    }
    add_vec(CNBD{}, xr, vec + D + M + g * (D / 4));         // + b2 (once)
#pragma unroll 1
    for (int c = 0; c < M / MC; ++c) {
      f32x4 h[NBC];
      {
        const unsigned va = lds_addr(vec + D + c * MC + g * (MC / 4));
        static_for<NBC>([&](auto bc) {
          constexpr int b = decltype(bc)::value;
          s16x8 t = ds_read_b128_asm<16 * b>(va);
          lgkm_wait_for<0>(t);
          h[b] = __builtin_bit_cast(f32x4, t);
        });
      }
      // the chunk's W1' pieces and W2 pieces share its slabs: one run, GELU + packing in front of the first W2 piece
      s16x8 hb[KSC];
      run(std::integral_constant<int, S::P_FFC>{}, [&](auto pc, const s16x8& a) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NBC * KSD) {
          constexpr int ks = p / NBC, b = p % NBC;
          h[b] = op16_mfma_16x16x32(a, xb[ks], h[b]);
        } else {
          if constexpr (p == NBC * KSD) {
            if constexpr (TRAIN) store_rows(CNBC{}, P.z + tok * M + c * MC + g * (MC / 4), h);
#pragma unroll
            for (int s2 = 0; s2 < KSC; ++s2)
#pragma unroll
              for (int j = 0; j < 8; ++j) hb[s2][j] = (short)f32_to_bf16_bits(wmz_gelu_fast(h[2 * s2 + (j >> 2)][j & 3]));
            if constexpr (TRAIN) {                          // GELU(z) as the second GEMM consumes it: 16 bytes per k-step
#pragma unroll
              for (int s2 = 0; s2 < KSC; ++s2)
                if (ok) *reinterpret_cast<s16x8*>(P.h + tok * M + c * MC + g * (MC / 4) + 8 * s2) = hb[s2];
            }
          }
          constexpr int q2 = p - NBC * KSD, ks = q2 / NBD, b = q2 % NBD;
          xr[b] = op16_mfma_16x16x32(a, hb[ks], xr[b]);
        }
      });
    }
    store_rows(CNBD{}, P.xo + tok * D + g * (D / 4), xr);
  }
  if constexpr (TAIL) {
    float mean, rstd;
    ln_stats(xr, mean, rstd);
    s16x8 xq[KSD], xn[KSD];
    pack(CNBD{}, xr, xq, 1.f, 0.f);
    pack(CNBD{}, xr, xn, rstd, -mean * rstd);
    if constexpr (TRAIN) {
      if (ok && g == 0) { P.st_attn[tok] = mean; P.st_attn[ntok + tok] = rstd; }
#pragma unroll
      for (int s2 = 0; s2 < KSD; ++s2)
        if (ok) *reinterpret_cast<s16x8*>(P.xn_attn + tok * D + g * (D / 4) + 8 * s2) = xn[s2];
    }
    f32x4 a[NBI];
#pragma unroll
    for (int b = 0; b < NBI; ++b) a[b] = (f32x4)(0.f);
    gemm(CNBI{}, CKSD{}, a, xq);
    store_rows(CNBI{}, P.q + tok * I + g * (I / 4), a);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int b = 0; b < NBI; ++b) a[b] = *reinterpret_cast<const f32x4*>(vec + 2 * D + M + t * I + g * (I / 4) + 4 * b);
      gemm(CNBI{}, CKSD{}, a, xn);
      store_rows(CNBI{}, P.kv + (long)t * P.voff + tok * P.ldkv + g * (I / 4), a);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the ring's run-ahead requests land before the LDS is handed back
}

template <int D, int I, int M, int MC, bool TRAIN>
int launch_chain(const ChainParams& P, int head, int tail, hipStream_t st) {
  const long ntok = (long)P.B * P.n_q * P.HW;
  const dim3 grid((unsigned)((ntok + CW * CT - 1) / (CW * CT))), block(CW * 64);
  if (head && tail) hipLaunchKernelGGL((layer_chain_kernel<D, I, M, MC, true, true, TRAIN>), grid, block, 0, st, P);
  else if (head) hipLaunchKernelGGL((layer_chain_kernel<D, I, M, MC, true, false, TRAIN>), grid, block, 0, st, P);
  else hipLaunchKernelGGL((layer_chain_kernel<D, I, M, MC, false, true, TRAIN>), grid, block, 0, st, P);
  WMZ_LAUNCH_CHECK("wmz_layer_chain_fwd");
  return WMZ_OK;
}

}  // namespace

int CHAIN_GROUP_FN(WMZ_CHAIN_GROUP)(const void* params, int D, int I, int M, int head, int tail, int train, hipStream_t st) {
  const ChainParams& P = *static_cast<const ChainParams*>(params);
#ifdef WMZ_OP16_F16
#define CHAIN_TRY(d, i, m, mc)                                                                            \
  if (D == d && I == i && M == m) {                                                                       \
    if (train) {                                                                                          \
      wmz_set_error("wmz_layer_chain_fwd_train: the half unit is inference only (precise mode)");         \
      return WMZ_ERR_UNSUPPORTED;                                                                         \
    }                                                                                                     \
    return launch_chain<d, i, m, mc, false>(P, head, tail, st);                                           \
  }
#else
#define CHAIN_TRY(d, i, m, mc)    \
  if (D == d && I == i && M == m) \
    return train ? launch_chain<d, i, m, mc, true>(P, head, tail, st) : launch_chain<d, i, m, mc, false>(P, head, tail, st);
#endif
  WMZ_CHAIN_WIDTHS_OF(WMZ_CHAIN_GROUP)(CHAIN_TRY)
#undef CHAIN_TRY
  return -1;
}

#if WMZ_CHAIN_GROUP == 0          // ---- the entry points (this group's bfloat16 / half unit)
namespace {
int chain_dispatch(const ChainParams& P, int D, int I, int M, int head, int tail, int train, hipStream_t st, const char* who) {
  int r = -1;
#define CHAIN_ASK(g) if (r == -1) r = CHAIN_GROUP_FN(g)(&P, D, I, M, head, tail, train, st);
  WMZ_CHAIN_ALL_GROUPS(CHAIN_ASK)
#undef CHAIN_ASK
  if (r == -1) {
    wmz_set_error("%s: widths (%d, %d, %d) not built (csrc/chain_widths.h)", who, D, I, M);
    return WMZ_ERR_UNSUPPORTED;
  }
  return r;
}
}  // namespace

// pieces (KB) per weight slab: every GEMM stage of the packed stream is padded to a multiple of it
#ifndef WMZ_OP16_F16
extern "C" int wmz_layer_chain_slab_pieces(void) { return CSP; }

// 1 when (D, I, M) has an instantiation; the hidden-chunk size MC of that instantiation through *mc (the host packer needs it)
extern "C" int wmz_layer_chain_supported(int D, int I, int M, int* mc) {
  int c = 0;
#define CHAIN_HAS(d, i, m, mcv) if (D == d && I == i && M == m) c = mcv;
  WMZ_CHAIN_ALL_WIDTHS(CHAIN_HAS)
#undef CHAIN_HAS
  if (mc) *mc = c;
  return c != 0;
}

#endif  // WMZ_OP16_F16 (slab size and supported widths are the bfloat16 unit's exports; they hold for both)

// One launch for everything per-token between two attention launches, widths (D, I, M) of wmz_layer_chain_supported.
//   o [B, n_q, HW, I] attention output (head != 0), x [B, n_in, HW, D] the stream (trailing n_q planes of every clip are read),
//   x_out [B, n_q, HW, D] (head), q_out [B, n_q, HW, I] and kv_out [2, B, n_q, HW, I] (tail != 0): all bf16, row-major.
//   wpack / vec: the weight stream and fp32 vectors in the kernel's order (world_modelz_amd/fused.py::_chain_pack), wpack with
//   CRING - 1 slabs (wmz_layer_chain_slab_pieces() KB each) of readable padding behind the last piece.
extern "C" int WMZ_FN(wmz_layer_chain_fwd_planes)(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                          const float* vec, int B, int n_q, int n_in, int HW, int D, int I, int M, int head,
                                          int tail, float eps, void* stream) {
  WMZ_REQUIRE(x && wpack && vec, "wmz_layer_chain_fwd_planes: null tensor");
  WMZ_REQUIRE(head || tail, "wmz_layer_chain_fwd_planes: nothing to do");
  WMZ_REQUIRE(!head || (o && x_out), "wmz_layer_chain_fwd_planes: head needs o and x_out");
  WMZ_REQUIRE(!tail || (q_out && kv_out), "wmz_layer_chain_fwd_planes: tail needs q_out and kv_out");
  WMZ_REQUIRE(B > 0 && n_q > 0 && n_in >= n_q && HW > 0, "wmz_layer_chain_fwd_planes: bad shape");
  ChainParams P;
  P.o = (const bf16_t*)o; P.x = (const bf16_t*)x; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.B = B; P.n_q = n_q; P.n_in = n_in; P.HW = HW; P.eps = eps;
  P.ldkv = I; P.voff = (long)B * n_q * HW * I;
  P.x1 = P.xn_ff = P.z = P.h = P.xn_attn = nullptr; P.st_ff = P.st_attn = nullptr;
  hipStream_t st = (hipStream_t)stream;
  return chain_dispatch(P, D, I, M, head, tail, 0, st, "wmz_layer_chain_fwd_planes");
}

#ifndef WMZ_OP16_F16      // (the training forward: the bfloat16 unit only)
// The training forward of the same launch (whole grids: ntok tokens, no trailing-planes form); wpack / vec exactly as for
// wmz_layer_chain_fwd_planes (LayerNorm affines folded).  Besides x_out / q_out / kv_out ([ntok, 2 I]: k | v per row) it writes what
// the step's backward reads: x1 [ntok, D] the raw feed-forward input (or NULL), xn_ff [ntok, D] its NORMALISED rows, z / h [ntok, M]
// pre-activation and GELU of it, st_ff [2, ntok] mean | rstd (head != 0); xn_attn [ntok, D] normalised rows, st_attn [2, ntok]
// (tail != 0).
extern "C" int wmz_layer_chain_fwd_train(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                         const float* vec, void* x1, void* xn_ff, void* z, void* h, float* st_ff, void* xn_attn,
                                         float* st_attn, long ntok, int D, int I, int M, int head, int tail, float eps,
                                         void* stream) {
  WMZ_REQUIRE(x && wpack && vec, "wmz_layer_chain_fwd_train: null tensor");
  WMZ_REQUIRE(head || tail, "wmz_layer_chain_fwd_train: nothing to do");
  WMZ_REQUIRE(!head || (o && x_out && xn_ff && z && h && st_ff), "wmz_layer_chain_fwd_train: head needs o, x_out and the saved tensors");
  WMZ_REQUIRE(!tail || (q_out && kv_out && xn_attn && st_attn), "wmz_layer_chain_fwd_train: tail needs q_out, kv_out, xn_attn, st_attn");
  WMZ_REQUIRE(ntok > 0 && ntok < (1L << 31), "wmz_layer_chain_fwd_train: bad token count");
  ChainParams P;
  P.o = (const bf16_t*)o; P.x = (const bf16_t*)x; P.xo = (bf16_t*)x_out; P.q = (bf16_t*)q_out; P.kv = (bf16_t*)kv_out;
  P.wpack = (const char*)wpack; P.vec = vec; P.B = 1; P.n_q = 1; P.n_in = 1; P.HW = (int)ntok; P.eps = eps;
  P.ldkv = 2 * I; P.voff = I;
  P.x1 = (bf16_t*)x1; P.xn_ff = (bf16_t*)xn_ff; P.z = (bf16_t*)z; P.h = (bf16_t*)h; P.st_ff = st_ff;
  P.xn_attn = (bf16_t*)xn_attn; P.st_attn = st_attn;
  hipStream_t st = (hipStream_t)stream;
  return chain_dispatch(P, D, I, M, head, tail, 1, st, "wmz_layer_chain_fwd_train");
}
#endif  // WMZ_OP16_F16
#endif  // WMZ_CHAIN_GROUP == 0
