// Backward of everything per-token between two attention launches for the widths of layer_chain.hip (the reference's published
// runs: dim 96 / mlp 256 and dim 384 / mlp 512, one head of 128), on that kernel's machinery: a workgroup = 8 waves = 128 tokens,
// a wave owns 16 tokens end to end, every GEMM transposed on MFMA 16x16x32 bf16 with the token on the lane, activations
// "lane-group-major" in registers (lane (t, g) owns the contiguous features g Wd/4 .. of its token: accumulator blocks and, after
// bf16 packing, B operands of the next GEMM with no exchange between lanes), the TRANSPOSED weights streamed as 1 KB pieces through
// an LDS ring filled by LDS-DMA.  Two kernels replace the five data-gradient GEMMs and the two LayerNorm-backward launches of a
// layer's op-by-op backward (main.py:278 `loss.backward()` through local_3d_attention.py:11-31, :46-53, :160-161):
//
//   wmz_chain_ff_bwd:   dy -> dz = (W2^T dy) * gelu'(z)  (MC hidden units at a time, stored: operand of dW1)
//                             dxh = W1'^T dz ; dx1 = dy + LayerNorm'(dxh ; xhat1, rstd1) ; do = Wout^T dx1
//   wmz_chain_qkv_bwd:  dx  = dx1 + Wq^T dq + LayerNorm'(Wk'^T dk + Wv'^T dv ; xhat, rstd)
// (W1' = W1 diag(g_ff), Wk' / Wv' = Wk / Wv diag(g_attn): the norms' weights folded in by the host packer, as in the forward)
//
// The forward (layer_chain.hip, TRAIN) leaves the NORMALISED rows xhat (no affine) and rstd; the weight gradients behind a norm are
// taken against xhat as plain GEMMs and converted by wmz_ln_affine_grads (as on the default-width path, layer_fused_bwd.hip).
// GELU' is the derivative of the forward's fitted GELU (wmz_gelu_fast_both).  z is fetched by loads the compiler does not track
// (asm + a counted wait: a tracked load in front of the weight ring makes hipcc drain the ring at its first use).
#include "wmz_common.h"
#include "chain_widths.h"

#ifndef WMZ_CHAIN_GROUP
#define WMZ_CHAIN_GROUP 0         // this unit's group of width triples (chain_widths.h); group 0 also holds the entry points
#endif
// a group's launchers: WMZ_OK / an error code, or -1 when the widths are not this group's
#define CHAIN_BWD_DECLARE_GROUP(g)                                                                                     \
  int WMZ_CHAIN_CAT(wmz_chain_ff_bwd_group, g)(const void* params, int D, int I, int M, hipStream_t st); \
  int WMZ_CHAIN_CAT(wmz_chain_qkv_bwd_group, g)(const void* params, int D, int I, hipStream_t st);
WMZ_CHAIN_ALL_GROUPS(CHAIN_BWD_DECLARE_GROUP)

namespace {

constexpr int CW = 8, CT = 16;
constexpr int CSP = 32;                     // 1 KB pieces per weight slab (wmz_layer_chain_slab_pieces())
constexpr int CSLAB = CSP * 1024;
constexpr int CRING = 3;
constexpr int CPW = CSP / CW;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int pad_slab(int n) { return (n + CSP - 1) / CSP * CSP; }

struct ChainBwdParams {
  // ff
  const bf16_t* dy;       // [N, D]  gradient of the layer's output rows
  const bf16_t* z;        // [N, M]  feed-forward pre-activation (forward)
  const bf16_t* xhat;     // [N, D]  normalised rows of the block's LayerNorm input (forward)
  const float* rstd;      // [N]
  bf16_t* dz;             // [N, M]
  bf16_t* dx1;            // [N, D]  ff: out.  qkv: in (the residual path's gradient)
  bf16_t* dout;           // [N, I]  ff: do
  // qkv
  const bf16_t* dq;       // [N, I]
  const bf16_t* dkv;      // [N, 2 I]  dk | dv per row
  bf16_t* dx;             // [N, D]
  const char* wpack;
  long ntok;
};

// The weight ring and the piece runner of layer_chain.hip (see there for the protocol).
struct Ring {
  char* ring;
  const char* wsrc;
  int slab, wave, lane;
  __device__ __forceinline__ void issue(int s) const {
    char* dst = ring + (s % CRING) * CSLAB + (wave * CPW) * 1024;
    const char* src = wsrc + (long)s * CSLAB;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void acquire() const {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CPW * (CRING - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(slab + CRING - 1);
  }
  template <int NP, typename F>
  __device__ __forceinline__ void run(F&& mf) {
    constexpr int PF = 4;
    s16x8 fr[PF];
    unsigned base = 0;
    static_for<NP>([&](auto pc) {
      constexpr int p = decltype(pc)::value, ps = p % CSP;
      constexpr int left = (CSP - 1 - ps) < (NP - 1 - p) ? (CSP - 1 - ps) : (NP - 1 - p);
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
    slab += 1;
  }
};

// acc[b] (NB blocks of 4 features, lane-group-major) <-> the lane's contiguous quarter row, 16-byte accesses
template <int NB>
__device__ __forceinline__ void load_rows(const bf16_t* row, f32x4 (&acc)[NB], bool add) {
#pragma unroll
  for (int s = 0; s < NB / 2; ++s) {
    const i32x4 c = *reinterpret_cast<const i32x4*>(row + 8 * s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = __uint_as_float(((unsigned)c[j]) << 16), hi = __uint_as_float(((unsigned)c[j]) & 0xFFFF0000u);
      const int e = 2 * j;
      if (add) { acc[2 * s + (e >> 2)][e & 3] += lo; acc[2 * s + ((e + 1) >> 2)][(e + 1) & 3] += hi; }
      else { acc[2 * s + (e >> 2)][e & 3] = lo; acc[2 * s + ((e + 1) >> 2)][(e + 1) & 3] = hi; }
    }
    if (add && (s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (adding into live accumulators: four 16-byte loads in flight, not all)
  }
}
template <int NB>
__device__ __forceinline__ void store_rows(bf16_t* row, const f32x4 (&acc)[NB], bool ok) {
#pragma unroll
  for (int s = 0; s < NB / 2; ++s) {
    const s16x4 a = cvt_pk4_bf16(acc[2 * s][0], acc[2 * s][1], acc[2 * s][2], acc[2 * s][3]);
    const s16x4 b = cvt_pk4_bf16(acc[2 * s + 1][0], acc[2 * s + 1][1], acc[2 * s + 1][2], acc[2 * s + 1][3]);
    if (ok) *reinterpret_cast<s16x8*>(row + 8 * s) = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
// fp32 blocks -> bf16 B operands (k-step s = blocks 2 s, 2 s + 1)
template <int NB>
__device__ __forceinline__ void pack(const f32x4 (&acc)[NB], s16x8 (&opnd)[NB / 2]) {
#pragma unroll
  for (int s = 0; s < NB / 2; ++s) {
    const s16x4 a = cvt_pk4_bf16(acc[2 * s][0], acc[2 * s][1], acc[2 * s][2], acc[2 * s][3]);
    const s16x4 b = cvt_pk4_bf16(acc[2 * s + 1][0], acc[2 * s + 1][1], acc[2 * s + 1][2], acc[2 * s + 1][3]);
    opnd[s] = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
// the lane's quarter row straight into B operands (a bf16 row IS the operand layout: 8 consecutive features per k-step)
template <int KS>
__device__ __forceinline__ void load_operand(const bf16_t* row, s16x8 (&opnd)[KS]) {
#pragma unroll
  for (int s = 0; s < KS; ++s) opnd[s] = *reinterpret_cast<const s16x8*>(row + 8 * s);
}

// LayerNorm backward of one token in registers: dn = gradient w.r.t. the NORMALISED rows (the norm's weight is folded into the
// transposed weights that produced it), xh = the normalised rows (bf16 operands); dn <- rstd (dn - mean(dn) - xh mean(dn xh)).
template <int D>
__device__ __forceinline__ void ln_backward(f32x4 (&dn)[D / 16], s16x8 (&xh)[D / 32], float rstd) {
  constexpr int NBD = D / 16;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int b = 0; b < NBD; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = bf16_bits_to_f32((unsigned short)xh[b >> 1][4 * (b & 1) + r]);
      s1 += dn[b][r];
      s2 = fmaf(dn[b][r], x, s2);
    }
    if ((b & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  const float m1 = wave_groups_sum(s1) * (1.f / D), m2 = wave_groups_sum(s2) * (1.f / D);
  const float a1 = -m1 * rstd, a2 = -m2 * rstd;
  // second pass IN PLACE, block by block: the operand rows are made opaque first (or hipcc keeps the first pass's D / 4 converted
  // values alive across the reduction) and every block is pinned (or it forms all of dn - m1 in fresh registers up front)
#pragma unroll
  for (int s = 0; s < D / 32; ++s) asm volatile("" : "+v"(xh[s]));
#pragma unroll
  for (int b = 0; b < NBD; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = bf16_bits_to_f32((unsigned short)xh[b >> 1][4 * (b & 1) + r]);
      dn[b][r] = fmaf(x, a2, fmaf(dn[b][r], rstd, a1));
    }
    asm volatile("" : "+v"(dn[b]));
  }
}

// ---------------------------------------------------------------------------------------------------------------- feed-forward side
template <int D, int I, int M, int MC>
__global__ __launch_bounds__(CW * 64, 2) void chain_ff_bwd_kernel(ChainBwdParams P) {
  constexpr int NBD = D / 16, NBI = I / 16, KSD = D / 32, NBC = MC / 16, KSC = MC / 32;
  constexpr int P_FFC = NBC * KSD + NBD * KSC;           // one chunk: W2^T rows (MC x D), then W1^T rows (D x MC)
  constexpr int P_OUT = pad_slab(NBI * KSD);             // Wout^T (I x D)
  static_assert(P_FFC % CSP == 0, "a feed-forward chunk must be whole slabs");
  constexpr int NACQ = (NBC * KSD + CSP - 1) / CSP;      // ring acquisitions between a chunk's start and its W2^T -> W1^T transition
  constexpr int ZL = MC / 32;                            // 16-byte z loads per lane and chunk
  __shared__ __attribute__((aligned(1024))) char ringmem[CRING * CSLAB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  long tok = (long)blockIdx.x * (CW * CT) + wave * CT + li;
  const bool ok = tok < P.ntok;
  if (!ok) tok = P.ntok - 1;

  Ring R{ringmem, P.wpack + (wave * CPW) * 1024 + lane * 16, 0, wave, lane};
#pragma unroll
  for (int s = 0; s < CRING - 1; ++s) R.issue(s);

  s16x8 dyb[KSD];
  load_operand<KSD>(P.dy + tok * D + g * (D / 4), dyb);
  f32x4 dn[NBD];
#pragma unroll
  for (int b = 0; b < NBD; ++b) dn[b] = (f32x4)(0.f);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < M / MC; ++c) {
    // this chunk's z, by loads hipcc does not track; retired by a counted wait at the transition (only the ring's requests are
    // younger: the previous chunk's dz stores were issued before)
    i32x4 zr[ZL];
    {
      const bf16_t* zp = P.z + tok * M + c * MC + g * (MC / 4);
      static_for<ZL>([&zr, zp](auto ic) {                  // (explicit captures: an asm operand alone does not capture)
        constexpr int i = decltype(ic)::value;
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(zr[i]) : "v"(zp), "n"(i * 16) : "memory");
      });
    }
    f32x4 dz[NBC];
#pragma unroll
    for (int b = 0; b < NBC; ++b) dz[b] = (f32x4)(0.f);
    s16x8 dzb[KSC];
    R.run<P_FFC>([&](auto pc, const s16x8& a) {
      constexpr int p = decltype(pc)::value;
      if constexpr (p < NBC * KSD) {
        constexpr int ks = p / NBC, b = p % NBC;
        dz[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dyb[ks], dz[b], 0, 0, 0);
      } else {
        if constexpr (p == NBC * KSD) {
          static_for<ZL>([&zr](auto ic) {
            constexpr int i = decltype(ic)::value;
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(zr[i]) : "n"(NACQ * CPW) : "memory");
          });
#pragma unroll
          for (int b = 0; b < NBC; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = 4 * b + r;                       // feature e of the lane's quarter: word e / 2 of the loaded row
              const unsigned w = (unsigned)zr[e >> 3][(e >> 1) & 3];
              const float zv = (e & 1) ? __uint_as_float(w & 0xFFFF0000u) : __uint_as_float(w << 16);
              float gv, dv;
              wmz_gelu_fast_both(zv, gv, dv);
              dz[b][r] *= dv;
            }
          store_rows<NBC>(P.dz + tok * M + c * MC + g * (MC / 4), dz, ok);
          pack<NBC>(dz, dzb);
        }
        constexpr int q2 = p - NBC * KSD, ks = q2 / NBD, b = q2 % NBD;
        dn[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dzb[ks], dn[b], 0, 0, 0);
      }
    });
  }
  // ---- dx1 = dy + LayerNorm'(dn)
  __builtin_amdgcn_sched_barrier(0);                       // (the loads below must not be hoisted over the chunk loop: registers)
  {
    s16x8 xh[KSD];
    load_operand<KSD>(P.xhat + tok * D + g * (D / 4), xh);
    ln_backward<D>(dn, xh, P.rstd[tok]);
  }
  __builtin_amdgcn_sched_barrier(0);
  load_rows<NBD>(P.dy + tok * D + g * (D / 4), dn, true);
  store_rows<NBD>(P.dx1 + tok * D + g * (D / 4), dn, ok);
  // ---- do = Wout^T dx1
  __builtin_amdgcn_sched_barrier(0);
  s16x8 xb[KSD];
  pack<NBD>(dn, xb);
  f32x4 a[NBI];
#pragma unroll
  for (int b = 0; b < NBI; ++b) a[b] = (f32x4)(0.f);
  R.run<NBI * KSD>([&](auto pc, const s16x8& w) {
    constexpr int p = decltype(pc)::value, ks = p / NBI, b = p % NBI;
    a[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xb[ks], a[b], 0, 0, 0);
  });
  store_rows<NBI>(P.dout + tok * I + g * (I / 4), a, ok);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------- attention side
template <int D, int I>
__global__ __launch_bounds__(CW * 64, 2) void chain_qkv_bwd_kernel(ChainBwdParams P) {
  constexpr int NBD = D / 16, KSD = D / 32, KSI = I / 32;
  __shared__ __attribute__((aligned(1024))) char ringmem[CRING * CSLAB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  long tok = (long)blockIdx.x * (CW * CT) + wave * CT + li;
  const bool ok = tok < P.ntok;
  if (!ok) tok = P.ntok - 1;

  Ring R{ringmem, P.wpack + (wave * CPW) * 1024 + lane * 16, 0, wave, lane};
#pragma unroll
  for (int s = 0; s < CRING - 1; ++s) R.issue(s);
  s16x8 kb[KSI], vb[KSI];
  load_operand<KSI>(P.dkv + tok * 2 * I + g * (I / 4), kb);
  load_operand<KSI>(P.dkv + tok * 2 * I + I + g * (I / 4), vb);
  __syncthreads();
  // ---- gradient w.r.t. the normalised rows: Wk^T dk + Wv^T dv
  f32x4 dn[NBD];
#pragma unroll
  for (int b = 0; b < NBD; ++b) dn[b] = (f32x4)(0.f);
  // (every stage of the stream is padded to whole slabs: a run leaves the padding pieces of its last slab unread)
  R.run<NBD * KSI>([&](auto pc, const s16x8& w) {
    constexpr int p = decltype(pc)::value, ks = p / NBD, b = p % NBD;
    dn[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, kb[ks], dn[b], 0, 0, 0);
  });
  R.run<NBD * KSI>([&](auto pc, const s16x8& w) {
    constexpr int p = decltype(pc)::value, ks = p / NBD, b = p % NBD;
    dn[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, vb[ks], dn[b], 0, 0, 0);
  });
  __builtin_amdgcn_sched_barrier(0);                       // (the loads below must not be hoisted over the GEMMs: registers)
  {
    s16x8 xh[KSD];
    load_operand<KSD>(P.xhat + tok * D + g * (D / 4), xh);
    ln_backward<D>(dn, xh, P.rstd[tok]);
  }
  // ---- + the residual path's gradient + Wq^T dq (the raw stream feeds to_q: quirk Q1)
  __builtin_amdgcn_sched_barrier(0);
  load_rows<NBD>(P.dx1 + tok * D + g * (D / 4), dn, true);
  __builtin_amdgcn_sched_barrier(0);
  s16x8 qb[KSI];
  load_operand<KSI>(P.dq + tok * I + g * (I / 4), qb);
  R.run<NBD * KSI>([&](auto pc, const s16x8& w) {
    constexpr int p = decltype(pc)::value, ks = p / NBD, b = p % NBD;
    dn[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, qb[ks], dn[b], 0, 0, 0);
  });
  store_rows<NBD>(P.dx + tok * D + g * (D / 4), dn, ok);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D, int I, int M, int MC>
int launch_ff(const ChainBwdParams& P, hipStream_t st) {
  const dim3 grid((unsigned)((P.ntok + CW * CT - 1) / (CW * CT))), block(CW * 64);
  hipLaunchKernelGGL((chain_ff_bwd_kernel<D, I, M, MC>), grid, block, 0, st, P);
  WMZ_LAUNCH_CHECK("wmz_chain_ff_bwd");
  return WMZ_OK;
}
template <int D, int I>
int launch_qkv(const ChainBwdParams& P, hipStream_t st) {
  const dim3 grid((unsigned)((P.ntok + CW * CT - 1) / (CW * CT))), block(CW * 64);
  hipLaunchKernelGGL((chain_qkv_bwd_kernel<D, I>), grid, block, 0, st, P);
  WMZ_LAUNCH_CHECK("wmz_chain_qkv_bwd");
  return WMZ_OK;
}

}  // namespace

int WMZ_CHAIN_CAT(wmz_chain_ff_bwd_group, WMZ_CHAIN_GROUP)(const void* params, int D, int I, int M, hipStream_t st) {
  const ChainBwdParams& P = *static_cast<const ChainBwdParams*>(params);
#define CHAIN_TRY(d, i, m, mc) if (D == d && I == i && M == m) return launch_ff<d, i, m, mc>(P, st);
  WMZ_CHAIN_WIDTHS_OF(WMZ_CHAIN_GROUP)(CHAIN_TRY)
#undef CHAIN_TRY
  return -1;
}
int WMZ_CHAIN_CAT(wmz_chain_qkv_bwd_group, WMZ_CHAIN_GROUP)(const void* params, int D, int I, hipStream_t st) {
  const ChainBwdParams& P = *static_cast<const ChainBwdParams*>(params);
  // (triples that share (D, I) name the same instantiation: the first match launches it)
#define CHAIN_TRY(d, i, m, mc) if (D == d && I == i) return launch_qkv<d, i>(P, st);
  WMZ_CHAIN_WIDTHS_OF(WMZ_CHAIN_GROUP)(CHAIN_TRY)
#undef CHAIN_TRY
  return -1;
}

#if WMZ_CHAIN_GROUP == 0          // ---- the entry points
// dy [ntok, D] -> dz [ntok, M], dx1 [ntok, D], dout [ntok, I] (all bf16, row-major).  z / xhat / rstd: what wmz_layer_chain_fwd_train
// left (xhat: its xn_ff output).  wpack: per hidden chunk of MC (wmz_layer_chain_supported) the pieces of W2[:, chunk]^T then of
// W1'[chunk, :]^T (W1' = W1 diag(g_ff)), then Wout^T padded to whole slabs, + 2 slabs of readable padding.
extern "C" int wmz_chain_ff_bwd(const void* dy, const void* z, const void* xhat, const float* rstd, void* dz, void* dx1, void* dout,
                                const void* wpack, long ntok, int D, int I, int M, void* stream) {
  WMZ_REQUIRE(dy && z && xhat && rstd && dz && dx1 && dout && wpack, "wmz_chain_ff_bwd: null tensor");
  WMZ_REQUIRE(ntok > 0, "wmz_chain_ff_bwd: bad token count");
  ChainBwdParams P = {};
  P.dy = (const bf16_t*)dy; P.z = (const bf16_t*)z; P.xhat = (const bf16_t*)xhat; P.rstd = rstd; P.dz = (bf16_t*)dz;
  P.dx1 = (bf16_t*)dx1; P.dout = (bf16_t*)dout; P.wpack = (const char*)wpack; P.ntok = ntok;
  hipStream_t st = (hipStream_t)stream;
  int r = -1;
#define CHAIN_ASK(g) if (r == -1) r = WMZ_CHAIN_CAT(wmz_chain_ff_bwd_group, g)(&P, D, I, M, st);
  WMZ_CHAIN_ALL_GROUPS(CHAIN_ASK)
#undef CHAIN_ASK
  if (r != -1) return r;
  wmz_set_error("wmz_chain_ff_bwd: widths (%d, %d, %d) not built (csrc/chain_widths.h)", D, I, M);
  return WMZ_ERR_UNSUPPORTED;
}

// dx [ntok, D] = dx1 + Wq^T dq + LayerNorm'(Wk'^T dk + Wv'^T dv; xhat, rstd).  dq [ntok, I], dkv [ntok, 2 I] (dk | dv per row), dx1
// [ntok, D] the residual path's gradient; wpack: pieces of Wk'^T, Wv'^T (Wk' = Wk diag(g_attn); each padded to whole slabs), Wq^T,
// + 2 slabs of padding.
extern "C" int wmz_chain_qkv_bwd(const void* dq, const void* dkv, const void* xhat, const float* rstd, const void* dx1, void* dx,
                                 const void* wpack, long ntok, int D, int I, void* stream) {
  WMZ_REQUIRE(dq && dkv && xhat && rstd && dx1 && dx && wpack, "wmz_chain_qkv_bwd: null tensor");
  WMZ_REQUIRE(ntok > 0, "wmz_chain_qkv_bwd: bad token count");
  ChainBwdParams P = {};
  P.dq = (const bf16_t*)dq; P.dkv = (const bf16_t*)dkv; P.xhat = (const bf16_t*)xhat; P.rstd = rstd;
  P.dx1 = (bf16_t*)const_cast<void*>(dx1); P.dx = (bf16_t*)dx; P.wpack = (const char*)wpack; P.ntok = ntok;
  hipStream_t st = (hipStream_t)stream;
  int r = -1;
#define CHAIN_ASK(g) if (r == -1) r = WMZ_CHAIN_CAT(wmz_chain_qkv_bwd_group, g)(&P, D, I, st);
  WMZ_CHAIN_ALL_GROUPS(CHAIN_ASK)
#undef CHAIN_ASK
  if (r != -1) return r;
  wmz_set_error("wmz_chain_qkv_bwd: widths (%d, %d) not built (csrc/chain_widths.h)", D, I);
  return WMZ_ERR_UNSUPPORTED;
}
#endif  // WMZ_CHAIN_GROUP == 0
