// Direct 3x3 (stride 1, pad 1) convolution of the VQ auto-encoder's conv encoder / decoder (vq-video-diffusion/autoencoder.py:8-15
// conv3x3 inside Residual :18-42, UpscaleResidual :89-131, SimpleResidualDecoder :134-152) in bf16, NHWC, for gfx950.
//
// Same GEMM and the same epilogue arithmetic as conv2d.hip's implicit-GEMM kernel (C[px, co] = sum_k A[px, k] W[co, k], k tap-major),
// re-staged around the LDS-DMA engine:
//   * a workgroup (4 waves) owns 256 output pixels of one image (8 x 32 or 16 x 16); its haloed input patch goes to LDS ONCE by
//     global_load_lds (16-byte pieces, out-of-image pixels fetched from a zero chunk: padding on the source side), 64 channels at
//     a pixel pitch of 144 bytes (128 + 16: 16 consecutive pixels cover the 64 banks -- conflict-free ds_read_b128 with the tap
//     and k-step offsets as instruction immediates / one scalar add, no swizzle arithmetic in the loop);
//   * the weights arrive PRE-PACKED in fragment order (wmz_conv3x3_direct_pack: one contiguous KB per MFMA B fragment), one
//     (tap x 64 channels) slab per step through a double-buffered LDS ring, again by DMA: no staging registers, one counted wait
//     and one s_barrier per 32 (Cout <= 64: 16) MFMAs of a wave;
//   * a wave owns 64 pixels x all output channels (2 x 4 blocks of 32 x 32, MFMA 32x32x16); fragment reads are inline asm with
//     counted lgkmcnt waits, the next k-step's six reads in flight under the current eight MFMAs;
//   * Cin = 128 runs as two passes over the patch (channels 0-63, then 64-127) into the same accumulators;
//   * epilogue per wave and in parallel (conv2d.hip: one wave at a time): without a residual the affine / LeakyReLU / rounding /
//     BatchNorm statistics happen in the accumulator layout (a lane owns a channel: the sums are lane-local) and the bf16 tile is
//     transposed through a wave-private LDS image into 16-byte row stores; with a residual the fp32 tile is staged and finished
//     on 16-byte row chunks exactly as conv2d.hip does.
// Two workgroups per CU (<= 80 KB of LDS each): one computes while the other loads its patch or stores its tile.
#include "wmz_common.h"
#include "wmz_debug.h"
#ifdef WMZ_CONV_STAMPS         // diagnostic build (tools/build_variant.py -DWMZ_CONV_STAMPS): s_memtime stamps of wave 0 of every workgroup of
                               // convr_kernel + the CU it ran on (tools/conv_stamps.py)
__device__ unsigned long long conv_stamps[8192 * 8];
#define CONV_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) conv_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CONV_STAMP(i) do {} while (0)
#endif
#ifndef WMZ_CONV_ABL
#define WMZ_CONV_ABL 0      // timing ablations of convr_kernel (tools/build_variant.py; results are garbage): 1 = a quarter of the A-fragment LDS reads, 2 = every weight fragment from the first two slabs (cache-resident)
#endif

namespace {

typedef const __attribute__((address_space(1))) void* cq_gptr_t;
typedef __attribute__((address_space(3))) void* cq_lptr_t;

__device__ __attribute__((aligned(16))) unsigned cq_zero_chunk[4] = {0u, 0u, 0u, 0u};

struct DirectParams {
  const bf16_t* x; const bf16_t* wpack; bf16_t* out;
  const float* bias; const float* scale; const float* shift; const bf16_t* res;
  float* stat_sum; float* stat_sq;
  int B, H, W, Cin, Cout, tiles_x, tiles_y;
  int Ho, Wo;           // output plane (= H, W at stride 1)
  float slope; int leaky;
  int skew, dbg;        // development knobs (wmz_debug_conv_knobs)
  int ncb_pack;         // channel blocks of the packed weight stream (wmz_conv3x3_direct_pack)
};

constexpr int CQ_PITCH = 144;          // bytes per patch pixel: 64 channels + one dead 16-byte slot
constexpr int CQ_SPP = 9;              // 16-byte slots per patch pixel

// counted wait that names the fragments it releases (their consumers cannot be scheduled above it)
template <int N>
__device__ __forceinline__ void cq_wait(s16x8& a, s16x8& b, s16x8& c, s16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void cq_wait(s16x8& a, s16x8& b, s16x8& c, s16x8& d, s16x8& e, s16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}

template <int NCB, int TW>
struct CqShape {
  static constexpr int TH = 256 / TW, PH = TH + 2, PW = TW + 2, NPX = PH * PW;
  static constexpr int PATCH = (NPX * CQ_PITCH + 1023) / 1024 * 1024;
  static constexpr int SLAB = 4 * NCB * 1024;                      // one tap x 64 channels x NCB*32 output channels
  static constexpr int LDS = PATCH + 2 * SLAB;
  static constexpr int OPITCH = NCB * 64 + 16;                     // bf16 output row of a wave's staging image (bytes)
  static constexpr int FPITCH = NCB * 128 + 16;                    // fp32 staging row (bytes)
  static constexpr int STAGE_A = 64 * OPITCH, STAGE_B = 32 * FPITCH;
  static constexpr int STAT_OFF = 4 * (STAGE_A > STAGE_B ? STAGE_A : STAGE_B);
  static_assert(STAT_OFF + 2 * NCB * 32 * 4 <= LDS, "epilogue staging fits the main loop's LDS");
  static_assert(LDS <= 81920, "two workgroups per CU");
};

template <int NCB, int TW>
__global__ __launch_bounds__(256, 2) void convq_kernel(DirectParams P) {
  using S = CqShape<NCB, TW>;
  __shared__ __attribute__((aligned(1024))) char lds[S::LDS];
  char* patch = lds;
  char* ring = lds + S::PATCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = lid % P.tiles_x; lid /= P.tiles_x;
  const int ty = lid % P.tiles_y;
  const int b = lid / P.tiles_y;
  const int oy0 = ty * S::TH, ox0 = tx * TW;
  const int npass = P.Cin >> 6, nslab = npass * 9;

  // ---- DMA issue: the patch of one channel pass, the weight slab of one (pass, tap)
  auto issue_patch = [&](int pass) {
    const bf16_t* xb = P.x + (long)b * P.H * P.W * P.Cin + pass * 64;
    for (int pc = wave; pc < S::PATCH / 1024; pc += 4) {
      const int q = pc * 64 + lane;
      const int pix = q / CQ_SPP, c = q - pix * CQ_SPP;
      const int py = pix / S::PW, px = pix - py * S::PW;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      const bool ok = c < 8 && pix < S::NPX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      const void* src = ok ? (const void*)(xb + ((long)iy * P.W + ix) * P.Cin + c * 8) : (const void*)cq_zero_chunk;
      __builtin_amdgcn_global_load_lds((cq_gptr_t)src, (cq_lptr_t)(patch + pc * 1024), 16, 0, 0);
    }
  };
  // weight stream: HALF slabs (2 k-steps x NCB fragments = 2 NCB KB) through a ring of four -- three half slabs in flight
  constexpr int HSLAB = S::SLAB / 2, PPW = NCB / 2;                // bytes per half slab; DMA pieces per wave and half slab
  const char* wsrc = reinterpret_cast<const char*>(P.wpack) + wave * (PPW * 1024) + lane * 16;
  auto issue_half = [&](int h) {
    char* dst = ring + (h & 3) * HSLAB + wave * (PPW * 1024);
    const char* src = wsrc + (long)h * HSLAB;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((cq_gptr_t)(src + i * 1024), (cq_lptr_t)(dst + i * 1024), 16, 0, 0);
  };

  // second workgroup of a CU (dispatch order: workgroups 256..511 of the first round): a start delay takes the two workgroups of a
  // CU -- and with them the chip's load / multiply / store phases -- out of step (measured: -13 % on the 64 -> 128 layer)
  if (P.skew > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < P.skew; ++i) __builtin_amdgcn_s_sleep(127);
  issue_patch(0);
  const int nhalf = 2 * nslab;
  issue_half(0);
  issue_half(1);
  issue_half(2);

  // this lane's two pixels (one per 32-pixel block of the wave's 64): tile pixel t -> patch pixel (py + kh, px + kw)
  unsigned abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = 64 * wave + 32 * i + l31;
    const int py = t / TW, px = t - py * TW;
    abase[i] = lds_addr(patch) + (unsigned)((py * S::PW + px) * CQ_PITCH + hh * 16);
  }
  const unsigned bbase = lds_addr(ring) + lane * 16;

  // the epilogue's per-channel constants (a lane owns channel 32 j + l31 of every block j): requested now, used after the loop
  float cbias[NCB], cscale[NCB], cshift[NCB];
#pragma unroll
  for (int j = 0; j < NCB; ++j) {
    const int col = 32 * j + l31;
    const bool cok = col < P.Cout;
    cbias[j] = (cok && P.bias) ? P.bias[col] : 0.f;
    cscale[j] = (cok && P.scale) ? P.scale[col] : 1.f;
    cshift[j] = (cok && P.shift) ? P.shift[col] : 0.f;
  }

  f32x16 acc[2][NCB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCB; ++j) acc[i][j] = (f32x16)(0.f);

  // patch + half slab 0 landed (only half slabs 1, 2 may still be in flight: 2 PPW pieces of this wave)
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
  __builtin_amdgcn_s_barrier();

  int tap = 0, pass = 0;
  for (int s = 0; s < ((P.dbg & 2) ? 1 : nslab); ++s) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const unsigned toff = (unsigned)((kh * S::PW + kw) * CQ_PITCH);
    const unsigned a0 = abase[0] + toff, a1 = abase[1] + toff;
    s16x8 af[2][2], bf[2][NCB];
    // the two half slabs of this (pass, tap): k-steps 0, 1 and 2, 3
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      const int h = 2 * s + hs;
      // (the buffer of half slab h + 3 held h - 1: every wave left it at the barrier that ended h - 1)
      const bool ahead = h + 3 < nhalf && !(P.dbg & 4);
      if (ahead) issue_half(h + 3);
      const unsigned bb = bbase + (unsigned)((h & 3) * HSLAB);
      auto reads = [&](auto kkc, auto setc) {
        constexpr int kk = decltype(kkc)::value, set = decltype(setc)::value;      // kk: k-step inside the half slab
        if (hs == 0) { af[set][0] = ds_read_b128_asm<kk * 32>(a0); af[set][1] = ds_read_b128_asm<kk * 32>(a1); }
        else { af[set][0] = ds_read_b128_asm<(kk + 2) * 32>(a0); af[set][1] = ds_read_b128_asm<(kk + 2) * 32>(a1); }
        static_for<NCB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          bf[set][j] = ds_read_b128_asm<(kk * NCB + j) * 1024>(bb);
        });
      };
      auto mfmas = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NCB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i], bf[set][j], acc[i][j], 0, 0, 0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      reads(I0{}, I0{});
      reads(I1{}, I1{});
      if constexpr (NCB == 4) cq_wait<6>(af[0][0], af[0][1], bf[0][0], bf[0][1], bf[0][2], bf[0][3]);
      else cq_wait<4>(af[0][0], af[0][1], bf[0][0], bf[0][1]);
      mfmas(I0{});
      if constexpr (NCB == 4) cq_wait<0>(af[1][0], af[1][1], bf[1][0], bf[1][1], bf[1][2], bf[1][3]);
      else cq_wait<0>(af[1][0], af[1][1], bf[1][0], bf[1][1]);
      mfmas(I1{});
      // half slab h + 1 landed (this wave's pieces; h + 2, h + 3 may be in flight) ...
      // (the last three half slabs of the stream wait for everything: nothing is requested behind them)
      if (ahead) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(P.dbg & 8)) __builtin_amdgcn_s_barrier();              // ... everyone's; and everyone is done with half slab h
    }
    if (++tap == 9) {
      tap = 0;
      if (++pass < npass) {                            // next 64 input channels: reload the patch (every wave left it above)
        issue_patch(pass);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  // ------------------------------------------------------------------------------------------------ epilogue
  if (P.dbg & 1) { if (acc[0][0][0] == 12345.f) P.out[0] = __float2bfloat16(acc[1][1][3]); return; }
  // (the last barrier above: every wave is done with the patch and the ring -- their LDS is the staging space now)
  bf16_t* O = P.out;
  constexpr int CH = NCB * 4;                                      // 16-byte chunks per output row of the wave's tile
  constexpr int RPI = 64 / CH;                                     // tile rows one store instruction of the wave covers
  static_assert(TW % RPI == 0, "a store instruction stays inside one row of pixels");
  float* stat_l = reinterpret_cast<float*>(lds + S::STAT_OFF);     // [2][NCB * 32]
  const bool want_stats = P.stat_sum != nullptr;
  if (want_stats) {
    if (tid < 2 * NCB * 32) stat_l[tid] = 0.f;
    __syncthreads();
  }
  // output addressing of the 16-byte row chunks: lane -> (row lane / CH of an iteration's RPI rows, chunk lane % CH); iteration
  // `it` covers tile rows RPI it ..: pixel row (RPI it) / TW of the wave's 64 / TW rows, pixel column (RPI it) % TW + lane / CH
  const int ch = lane & (CH - 1), col0 = ch * 8, lrow = lane / CH;
  const long pix00 = ((long)b * P.H + oy0 + wave * (64 / TW)) * P.W + ox0 + lrow;
  bf16_t* const obase = O + pix00 * P.Cout + col0;
  const bf16_t* const rbase = P.res ? P.res + pix00 * P.Cout + col0 : nullptr;
  auto it_offset = [&](int it) {                                   // elements from obase / rbase (wave-uniform)
    const int r0 = RPI * it;
    return ((long)(r0 / TW) * P.W + (r0 % TW)) * P.Cout;
  };
  if (P.res == nullptr) {
    // ---- mode A: finish in the accumulator layout (a lane owns a channel), transpose the bf16 tile through LDS
    char* stage = lds + wave * S::STAGE_A;
    auto finish = [&](auto affc, auto leakyc, auto statsc) {
      constexpr bool AFF = decltype(affc)::value, LEAKY = decltype(leakyc)::value, STATS = decltype(statsc)::value;
#pragma unroll
      for (int j = 0; j < NCB; ++j) {
        const int col = 32 * j + l31;
        const float bv = cbias[j], sc = cscale[j], sh = cshift[j];
        f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
        char* const wp = stage + (4 * hh) * S::OPITCH + col * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rp = 0; rp < 8; ++rp) {
            const int reg = 2 * rp;                                 // registers reg, reg + 1 = rows rl, rl + 1
            const int rl = 32 * i + (reg & 3) + 8 * (reg >> 2);
            float v0 = acc[i][j][reg] + bv, v1 = acc[i][j][reg + 1] + bv;
            if constexpr (AFF) { v0 = v0 * sc + sh; v1 = v1 * sc + sh; }
            if constexpr (LEAKY) { v0 = fmaxf(v0, v0 * P.slope); v1 = fmaxf(v1, v1 * P.slope); }
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, bf16x2_t));
            *reinterpret_cast<unsigned short*>(wp + rl * S::OPITCH) = (unsigned short)pk;
            *reinterpret_cast<unsigned short*>(wp + (rl + 1) * S::OPITCH) = (unsigned short)(pk >> 16);
            if constexpr (STATS) {                                  // statistics of what the next stage will read
              const f32x2 q = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xFFFF0000u)};
              s1 += q;
              s2 = q * q + s2;
            }
          }
        if constexpr (STATS) {
          const float t1 = wave_halves_sum(s1[0] + s1[1]), t2 = wave_halves_sum(s2[0] + s2[1]);
          if (hh == 0 && col < P.Cout) { atomicAdd(&stat_l[col], t1); atomicAdd(&stat_l[NCB * 32 + col], t2); }
        }
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    const int sel = (P.scale ? 4 : 0) | (P.leaky ? 2 : 0) | (want_stats ? 1 : 0);
    switch (sel) {
      case 0: finish(F_{}, F_{}, F_{}); break;
      case 1: finish(F_{}, F_{}, T_{}); break;
      case 2: finish(F_{}, T_{}, F_{}); break;
      case 3: finish(F_{}, T_{}, T_{}); break;
      case 4: finish(T_{}, F_{}, F_{}); break;
      case 5: finish(T_{}, F_{}, T_{}); break;
      case 6: finish(T_{}, T_{}, F_{}); break;
      default: finish(T_{}, T_{}, T_{}); break;
    }
    // (the same wave wrote what it reads back: LDS operations of a wave complete in order)
    __builtin_amdgcn_wave_barrier();
    if (col0 < P.Cout) {
      const char* const rp_ = stage + lrow * S::OPITCH + ch * 16;
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const i32x4 v = *reinterpret_cast<const i32x4*>(rp_ + it * RPI * S::OPITCH);
        *reinterpret_cast<i32x4*>(obase + it_offset(it)) = v;
      }
    }
  } else {
    // ---- mode B: residual add in fp32 on 16-byte row chunks (conv2d.hip's staged epilogue, per wave)
    char* stage = lds + wave * S::STAGE_B;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < NCB; ++j) {
        const float bv = cbias[j], sc = cscale[j], sh = cshift[j];
        char* const wp = stage + (4 * hh) * S::FPITCH + (32 * j + l31) * 4;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int rl = (reg & 3) + 8 * (reg >> 2);
          *reinterpret_cast<float*>(wp + rl * S::FPITCH) = (acc[i][j][reg] + bv) * sc + sh;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (col0 < P.Cout) {
        const char* const rp_ = stage + lrow * S::FPITCH + ch * 32;
        i32x4 rv[CH / 2];
#pragma unroll
        for (int it = 0; it < CH / 2; ++it) rv[it] = *reinterpret_cast<const i32x4*>(rbase + it_offset(i * (CH / 2) + it));
#pragma unroll
        for (int it = 0; it < CH / 2; ++it) {
          const f32x4 va = *reinterpret_cast<const f32x4*>(rp_ + it * RPI * S::FPITCH);
          const f32x4 vb = *reinterpret_cast<const f32x4*>(rp_ + it * RPI * S::FPITCH + 16);
          float f[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[2 * e] += __uint_as_float(((unsigned)rv[it][e]) << 16);
            f[2 * e + 1] += __uint_as_float(((unsigned)rv[it][e]) & 0xFFFF0000u);
          }
          if (P.leaky) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], f[e] * P.slope);
          }
          i32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            pk[e] = (int)__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){f[2 * e], f[2 * e + 1]}, bf16x2_t));
          *reinterpret_cast<i32x4*>(obase + it_offset(i * (CH / 2) + it)) = pk;
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float q0 = __uint_as_float(((unsigned)pk[e]) << 16), q1 = __uint_as_float(((unsigned)pk[e]) & 0xFFFF0000u);
              s1[2 * e] += q0; s2[2 * e] = fmaf(q0, q0, s2[2 * e]);
              s1[2 * e + 1] += q1; s2[2 * e + 1] = fmaf(q1, q1, s2[2 * e + 1]);
            }
          }
        }
      }
    }
    if (want_stats) {
      // the lanes of a chunk column sit CH lanes apart
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o); s2[e] += __shfl_xor(s2[e], o); }
      }
      if (lane < CH && col0 < P.Cout) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { atomicAdd(&stat_l[col0 + e], s1[e]); atomicAdd(&stat_l[NCB * 32 + col0 + e], s2[e]); }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < NCB * 32 && tid < P.Cout) {
      const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * P.Cout;
      atomicAdd(P.stat_sum + rep + tid, stat_l[tid]);
      atomicAdd(P.stat_sq + rep + tid, stat_l[NCB * 32 + tid]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// convr_kernel: the same convolution with DECOUPLED waves.  convq_kernel's four waves share every weight slab (one s_barrier per
// slab, a DMA wait in front of it) -- measured on the 64 -> 128 layer of 256 frames: 37 of 220 us in those waits, and the chip's
// workgroups march through load / multiply / store in step.  Here a wave owns ONE block of 32 output channels for 2 NCB blocks
// of 32 pixels (Cout = 128: all 256 pixels of the tile), so its weight fragments are private: they come straight from global
// memory (L2-resident, one contiguous KB per fragment) into a register ring two slabs deep -- no LDS ring, no barrier and no
// shared wait inside the loop; the patch is the only LDS operand (one ds_read_b128 per MFMA, a window of four in flight across
// slab boundaries).  Same k order per accumulator as convq_kernel / conv2d.hip: the same bits.
// STRIDE = 2 (3x3 / stride 2 / pad 1, Cout = 128: the first convolution of the down-sampling Residual, autoencoder.py:27-33): the
// same machine on the four PARITY planes of the input -- x[2 i + p, 2 j + q] for (p, q) in {0, 1}^2: tap (kh, kw) reads plane
// (kh != 1, kw != 1) at the output pixel's own (i, j) or one row / column before it, so inside a plane the lanes of a fragment
// again walk consecutive pixels (the DMA de-interleaves on the source side).  Round 6: every input pixel belongs to ONE plane, so
// the taps are grouped by the planes they read and a pixel's 64 channels (one 128-byte line) are fetched ONCE: group A = the four
// corner taps (plane (1, 1)), group B = the four edge taps (planes (0, 1), (1, 0)), group C = the centre tap (plane (0, 0)); a
// 64-channel pass is [stage A's plane, 4 taps][stage B's two planes, 4 taps][stage C's plane, 1 tap] -- at most two planes = 46 KB
// resident, THREE workgroups per CU at 161 VGPRs (round 5 staged all four planes 32 channels at a time: every line
// fetched as two halves in two passes -- fabric traffic 1.86 x the input, pmc_conv.json).  An 8 x 16 output tile needs 9 x 17
// pixels of a plane at the stride-1 pixel pitch of 144 bytes; a plane ROW is padded to 160 sixteen-byte slots (= 0 mod 16): the 16
// lanes of a ds_read_b128 group are 8 pixels of one tile row and 8 of the next, and with 17-pixel rows two of them met in a bank
// (round 5: 0.47 of the LDS cycles were conflicts) -- with rows a multiple of 16 slots apart the 16 slot residues are distinct.
// The packed weight stream is the stride-1 one, its fragments visited in this order.
template <int NCB, int TW, int STRIDE>
struct CrShape {
  static constexpr int TILE_PX = STRIDE == 1 ? 256 : 128;
  static constexpr int TH = TILE_PX / TW, PH = TH + 2, PW = TW + 2;
  static constexpr int PITCH = CQ_PITCH;                           // bytes per patch pixel: 64 channels + one dead 16-byte slot
  static constexpr int SPP = PITCH / 16;                           // 16-byte slots per patch pixel
  static constexpr int ROW_SLOTS = STRIDE == 1 ? PW * SPP : 160;   // 16-byte slots per patch row (STRIDE 2: 17 pixels padded, see above)
  static constexpr int ROW_BYTES = ROW_SLOTS * 16;
  static constexpr int PLANE_BYTES = (TH + 1) * ROW_BYTES;         // STRIDE 2: one parity plane of the patch (TH + 1 rows)
  static constexpr int KS = 4;                                     // k-steps per (pass, tap)
  static constexpr int PATCH = ((STRIDE == 1 ? PH * PW * PITCH : 2 * PLANE_BYTES) + 1023) / 1024 * 1024;
  static constexpr int NPB = TILE_PX * NCB / 128;                  // 32-pixel blocks per wave
  static constexpr int BPR = 2;                                    // blocks per epilogue round (bf16 staging, double-buffered)
  static constexpr int OPITCH = 80, FPITCH = 144;                  // staging rows: 32 channels bf16 / fp32 + 16 bytes
  static constexpr int STAGE_A = 2 * BPR * 32 * OPITCH, STAGE_B = 64 * FPITCH;
  static constexpr int STAGE = STAGE_A > STAGE_B ? STAGE_A : STAGE_B;
  static constexpr int LDS = PATCH > 4 * STAGE ? PATCH : 4 * STAGE;
  static_assert(STRIDE == 1 || (TW == 16 && NCB == 4), "the stride-2 form is built for Cout = 128 on 8 x 16 tiles");
  static_assert(STRIDE == 1 || (ROW_SLOTS >= (TW + 1) * SPP && ROW_SLOTS % 16 == 0), "conflict-free plane rows");
  static_assert(LDS <= 81920, "two workgroups per CU");
  static_assert(STRIDE == 1 || LDS <= 53248, "stride 2: three workgroups per CU");
};
// STRIDE 2: tap order of a 64-channel pass (group A: the 4 corner taps, group B: the 4 edge taps, group C: the centre tap), the
// group of the idx-th tap and the slot its plane is staged in
__host__ __device__ constexpr int cr2_tap(int idx) { constexpr int o[9] = {0, 2, 6, 8, 3, 5, 1, 7, 4}; return o[idx]; }
__host__ __device__ constexpr int cr2_group(int idx) { return idx < 4 ? 0 : (idx < 8 ? 1 : 2); }
__host__ __device__ constexpr int cr2_slot(int kh, int kw) { return (kh != 1 && kw == 1) ? 1 : 0; }   // B: (0, 1) -> slot 0, (1, 0) -> slot 1

// NPASS = Cin / 64
template <int NCB, int TW, int NPASS, int STRIDE = 1>
__global__ __launch_bounds__(256, STRIDE == 1 ? 2 : 3) void convr_kernel(DirectParams P) {
  using S = CrShape<NCB, TW, STRIDE>;
  constexpr int NPB = S::NPB, KS = S::KS, NSEQ = KS * NPB, WIN = 4;
  __shared__ __attribute__((aligned(1024))) char lds[S::LDS];
  char* patch = lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int cb = wave % NCB, i0 = (wave / NCB) * NPB;              // channel block; first pixel block
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = lid % P.tiles_x; lid /= P.tiles_x;
  const int ty = lid % P.tiles_y;
  const int b = lid / P.tiles_y;
  const int oy0 = ty * S::TH, ox0 = tx * TW;
  constexpr int nslab = NPASS * 9;                                 // (64-channel pass, tap) slabs

  // stride 1: pass = 64-channel pass.  stride 2: pass = 3 * (64-channel pass) + group (0: A = plane (1, 1); 1: B = planes (0, 1), (1, 0);
  // 2: C = plane (0, 0))
  auto issue_patch = [&](int pass) {
    if constexpr (STRIDE == 1) {
      const bf16_t* xb = P.x + (long)b * P.H * P.W * P.Cin + pass * 64;
      for (int pc = wave; pc < S::PATCH / 1024; pc += 4) {
        const int q = pc * 64 + lane;
        const int pix = q / S::SPP, c = q - pix * S::SPP;
        const int py = pix / S::PW, px = pix - py * S::PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool ok = c < S::SPP - 1 && pix < S::PH * S::PW && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        const void* src = ok ? (const void*)(xb + ((long)iy * P.W + ix) * P.Cin + c * 8) : (const void*)cq_zero_chunk;
        __builtin_amdgcn_global_load_lds((cq_gptr_t)src, (cq_lptr_t)(patch + pc * 1024), 16, 0, 0);
      }
    } else {
      const int c64 = pass / 3, grp = pass - 3 * c64;
      const bf16_t* xb = P.x + (long)b * P.H * P.W * P.Cin + c64 * 64;
      const int nplanes = grp == 1 ? 2 : 1;
      const int npc = (nplanes * S::PLANE_BYTES + 1023) / 1024;
      for (int pc = wave; pc < npc; pc += 4) {
        const int q = pc * 64 + lane;
        const int slot = q / ((S::TH + 1) * S::ROW_SLOTS), r = q - slot * ((S::TH + 1) * S::ROW_SLOTS);
        const int pi = r / S::ROW_SLOTS, sl = r - pi * S::ROW_SLOTS;
        const int pj = sl / S::SPP, c = sl - pj * S::SPP;
        // plane of the slot: A: (1, 1); B: slot 0 -> (0, 1), slot 1 -> (1, 0); C: (0, 0)
        const int pr = grp == 0 ? 1 : (grp == 1 ? slot : 0), pq = grp == 0 ? 1 : (grp == 1 ? 1 - slot : 0);
        const int iy = 2 * (oy0 - 1 + pi) + pr, ix = 2 * (ox0 - 1 + pj) + pq;
        const bool ok = c < S::SPP - 1 && pj <= TW && slot < nplanes && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        const void* src = ok ? (const void*)(xb + ((long)iy * P.W + ix) * P.Cin + c * 8) : (const void*)cq_zero_chunk;
        __builtin_amdgcn_global_load_lds((cq_gptr_t)src, (cq_lptr_t)(patch + pc * 1024), 16, 0, 0);
      }
    }
  };
  CONV_STAMP(0);
#ifdef WMZ_CONV_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    conv_stamps[blockIdx.x * 8 + 6] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  if (P.skew > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < P.skew; ++i) __builtin_amdgcn_s_sleep(127);
  issue_patch(0);
  CONV_STAMP(1);

  // this wave's weight fragments: fragment row f of the packed stream ([64-channel pass][tap][k-step 0..3]), block cb of ncb_pack
  const s16x8* const wp = reinterpret_cast<const s16x8*>(P.wpack) + cb * 64 + lane;
  const long wstep = (long)P.ncb_pack * 64;                        // fragment rows are ncb_pack KB apart
  // fragment row of (slab s, k-step kk): stride 1: 4 s + kk; stride 2: slab s = (64-channel pass s / 9, the (s % 9)-th tap of cr2_tap's order)
  auto frag_row = [](int s, int kk) {
    if (STRIDE == 1) return 4 * s + kk;
    return ((s / 9) * 9 + cr2_tap(s % 9)) * 4 + kk;
  };
  // Loads hipcc does not track (inline asm): its own wait in front of a fragment's first use comes out as vmcnt(0) -- a drain of
  // the whole ring every other slab -- where the issue order says exactly 2 KS - 1 younger loads may still be in flight: fragment
  // (s, kk) is requested behind k-step kk of slab s - 2, followed by KS - 1 - kk more of that slab, KS of slab s - 1 and kk of slab s.
  // Nothing may touch a destination register between its load and the counted wait (straight-line code, no copies: checked on
  // the ISA by tools/check_untracked_conv.py).
  auto wload = [&](long frag) {
    s16x8 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(wp + frag * wstep) : "memory");
    return v;
  };
  s16x8 bq[2][KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) bq[0][kk] = wload(frag_row(0, kk));
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) bq[1][kk] = wload(frag_row(1, kk));

  float cbias, cscale, cshift;
  {
    const int col = 32 * cb + l31;
    const bool cok = col < P.Cout;
    cbias = (cok && P.bias) ? P.bias[col] : 0.f;
    cscale = (cok && P.scale) ? P.scale[col] : 1.f;
    cshift = (cok && P.shift) ? P.shift[col] : 0.f;
  }

  unsigned abase[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int t = 32 * (i0 + i) + l31;
    const int py = t / TW, px = t - py * TW;
    abase[i] = lds_addr(patch) + (unsigned)(py * S::ROW_BYTES + px * S::PITCH + hh * 16);
  }
  f32x16 acc[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) acc[i] = (f32x16)(0.f);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // patch pieces (and the first weight fragments) landed
  __builtin_amdgcn_s_barrier();
  CONV_STAMP(2);

  // A-fragment sequence of a slab (= tap t of the pass): n = kk * NPB + i -> ds_read_b128 at abase[i] + tap offset + 32 kk, the
  // offsets instruction immediates (all slabs are unrolled: straight-line code); WIN reads in flight, carried across slabs (the
  // first WIN reads of tap t + 1 are issued under the last MFMAs of tap t).
  s16x8 fr[WIN];
  auto read_n = [&](auto tc, auto nc) {
    constexpr int t = STRIDE == 1 ? decltype(tc)::value % 9 : cr2_tap(decltype(tc)::value % 9);
    constexpr int n = decltype(nc)::value, kk = n / NPB, i = n % NPB;
    constexpr int kh = t / 3, kw = t % 3;
    constexpr int off = (STRIDE == 1 ? kh * S::ROW_BYTES + kw * S::PITCH
                                     : cr2_slot(kh, kw) * S::PLANE_BYTES + (kh != 0) * S::ROW_BYTES + (kw != 0) * S::PITCH) + kk * 32;
#if WMZ_CONV_ABL & 1
    // timing ablation (garbage results): three of four A-fragment reads become register copies -- every destination is still WRITTEN
    // (a skipped asm read leaves its register unassigned: hipcc re-uses it and the counted waits protect nothing -- that variant faulted)
    if constexpr (i % 4 != 0) {
      fr[n % WIN] = (s16x8)((short)abase[i]);
      asm volatile("" : "+v"(fr[n % WIN]));
      return;
    }
#endif
    fr[n % WIN] = ds_read_b128_asm<off>(abase[i]);
  };
  using I0 = std::integral_constant<int, 0>;
  static_for<WIN>([&](auto nc) { read_n(I0{}, nc); });

  static_for<nslab>([&](auto sc) {
    constexpr int s = decltype(sc)::value, t = s % 9;
    constexpr int snext = s + 2 < nslab ? s + 2 : nslab - 1;       // (the tail re-loads the last slab: uniform counts)
    static_for<NSEQ>([&](auto nc) {
      constexpr int n = decltype(nc)::value, kk = n / NPB, i = n % NPB;
      if constexpr (i == 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bq[s & 1][kk]) : "n"(2 * KS - 1) : "memory");
      lgkm_wait_for<WIN - 1>(fr[n % WIN]);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[n % WIN], bq[s & 1][kk], acc[i], 0, 0, 0);
      if constexpr (n + WIN < NSEQ) read_n(sc, std::integral_constant<int, n + WIN>{});
      else read_n(std::integral_constant<int, s + 1>{}, std::integral_constant<int, n + WIN - NSEQ>{});    // next tap
#if WMZ_CONV_ABL & 2
      if constexpr (i == NPB - 1) bq[s & 1][kk] = wload(frag_row(snext & 1, kk));   // timing ablation: the same two slabs over and over (cache-resident)
#else
      if constexpr (i == NPB - 1) bq[s & 1][kk] = wload(frag_row(snext, kk));   // k-step kk done: its register takes slab s + 2
#endif
    });
    if constexpr ((t == 8 || (STRIDE == 2 && (t == 3 || t == 7))) && s + 1 < nslab) {   // the patch of the next 64 channels (stride 2: of the next tap group)
      // (the window's run-ahead reads -- of the OLD patch: dead values -- retire here; their registers stay named until then:
      //  hipcc hands the register of a dead asm result to the next instruction while the LDS return is still in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]) :: "memory");
      __builtin_amdgcn_s_barrier();                                // every wave is done with the patch
      issue_patch(STRIDE == 1 ? (s + 1) / 9 : 3 * ((s + 1) / 9) + cr2_group((s + 1) % 9));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      static_for<WIN>([&](auto nc) { read_n(std::integral_constant<int, s + 1>{}, nc); });   // (the window was read from the old patch)
    }
  });
  // the window's run-ahead reads and the ring's (redundant) tail loads retire here: their registers stay named until then
  if constexpr (KS == 4)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]),
                   "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]) :: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]) :: "memory");

  // ------------------------------------------------------------------------------------------------ epilogue
  CONV_STAMP(3);
  if (P.dbg & 1) { if (acc[0][0] == 12345.f) P.out[0] = __float2bfloat16(acc[1][3]); return; }
  __syncthreads();                                                 // every wave is done with the patch: its LDS is the staging space
  CONV_STAMP(4);
  char* const stage = lds + wave * S::STAGE;
  const int col = 32 * cb + l31;
  const bool want_stats = P.stat_sum != nullptr;
  const int ch = lane & 3, lpx = lane >> 2;                        // store phase: 16-byte chunk of the 32 channels, pixel of 16
  const int ccol = 32 * cb + ch * 8;
  const bool chunk_ok = ccol < P.Cout;
  // pixel address of the 16-pixel group that starts at tile pixel t0 (t0 % 16 == 0: one row of pixels)
  auto group_ptr = [&](const bf16_t* base, int t0) {
    const int py = t0 / TW, px = t0 - py * TW + lpx;
    return base + (((long)b * P.Ho + oy0 + py) * P.Wo + ox0 + px) * P.Cout + ccol;
  };
  const bool nt = (P.dbg & 128) != 0;
  auto st_out = [&](const bf16_t* p, const i32x4& v) {
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<i32x4*>(const_cast<bf16_t*>(p)));
    else *reinterpret_cast<i32x4*>(const_cast<bf16_t*>(p)) = v;
  };
  if (P.res == nullptr) {
    f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
    auto finish = [&](auto affc, auto leakyc, auto statsc) {
      constexpr bool AFF = decltype(affc)::value, LEAKY = decltype(leakyc)::value, STATS = decltype(statsc)::value;
      // rounds of 64 pixels through a double-buffered staging image: the rows of round r are read back right behind their
      // writes and STORED one round later, under the arithmetic of round r + 1 (no wait for the LDS round trip)
      constexpr int NR = NPB / S::BPR, HALF = S::BPR * 32 * S::OPITCH, NIT = S::BPR * 2;
      i32x4 prev[NIT];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        char* const buf = stage + (r & 1) * HALF;
        char* const wpos = buf + (4 * hh) * S::OPITCH + l31 * 2;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ib = 0; ib < S::BPR; ++ib) {
#pragma unroll
          for (int rp = 0; rp < 8; ++rp) {
            const int reg = 2 * rp, rl = 32 * ib + (reg & 3) + 8 * (reg >> 2);
            float v0 = acc[r * S::BPR + ib][reg] + cbias, v1 = acc[r * S::BPR + ib][reg + 1] + cbias;
            if constexpr (AFF) { v0 = v0 * cscale + cshift; v1 = v1 * cscale + cshift; }
            if constexpr (LEAKY) { v0 = fmaxf(v0, v0 * P.slope); v1 = fmaxf(v1, v1 * P.slope); }
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, bf16x2_t));
            *reinterpret_cast<unsigned short*>(wpos + rl * S::OPITCH) = (unsigned short)pk;
            *reinterpret_cast<unsigned short*>(wpos + (rl + 1) * S::OPITCH) = (unsigned short)(pk >> 16);
            if constexpr (STATS) {                                  // statistics of what the next stage will read
              const f32x2 q = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xFFFF0000u)};
              s1 += q;
              s2 = q * q + s2;
            }
          }
          if (ib == 0 && r > 0 && chunk_ok && !(P.dbg & 32)) {      // the previous round's rows leave under this round's arithmetic
#pragma unroll
            for (int it = 0; it < NIT; ++it)
              st_out(group_ptr(P.out, 32 * (i0 + (r - 1) * S::BPR) + 16 * it), prev[it]);
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) prev[it] = *reinterpret_cast<const i32x4*>(buf + (it * 16 + lpx) * S::OPITCH + ch * 16);
      }
      if (chunk_ok && !(P.dbg & 32)) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          st_out(group_ptr(P.out, 32 * (i0 + (NR - 1) * S::BPR) + 16 * it), prev[it]);
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    const int sel = (P.scale ? 4 : 0) | (P.leaky ? 2 : 0) | (want_stats ? 1 : 0);
    if (P.dbg & 64) {
      if (chunk_ok) {
#pragma unroll
        for (int it = 0; it < NPB * 2; ++it) {
          const i32x4 v = *reinterpret_cast<const i32x4*>(stage + ((it & 7) * 16 + lpx) * S::OPITCH + ch * 16);
          *reinterpret_cast<i32x4*>(const_cast<bf16_t*>(group_ptr(P.out, 32 * i0 + 16 * it))) = v;
        }
      }
    } else
    switch (sel) {
      case 0: finish(F_{}, F_{}, F_{}); break;
      case 1: finish(F_{}, F_{}, T_{}); break;
      case 2: finish(F_{}, T_{}, F_{}); break;
      case 3: finish(F_{}, T_{}, T_{}); break;
      case 4: finish(T_{}, F_{}, F_{}); break;
      case 5: finish(T_{}, F_{}, T_{}); break;
      case 6: finish(T_{}, T_{}, F_{}); break;
      default: finish(T_{}, T_{}, T_{}); break;
    }
    if (want_stats) {
      const float t1 = wave_halves_sum(s1[0] + s1[1]), t2 = wave_halves_sum(s2[0] + s2[1]);
      if (hh == 0 && col < P.Cout) {
        const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * P.Cout;
        atomicAdd(P.stat_sum + rep + col, t1);
        atomicAdd(P.stat_sq + rep + col, t2);
      }
    }
  } else {
    // residual: fp32 rows of 64 pixels x 32 channels staged per round, finished on 16-byte chunks (conv2d.hip's arithmetic)
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    char* const wpos = stage + (4 * hh) * S::FPITCH + l31 * 4;
#pragma unroll
    for (int r = 0; r < NPB / 2; ++r) {
      i32x4 rv[4];
      if (chunk_ok) {
#pragma unroll
        for (int it = 0; it < 4; ++it) rv[it] = *reinterpret_cast<const i32x4*>(group_ptr(P.res, 32 * (i0 + 2 * r) + 16 * it));
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int rl = 32 * ib + (reg & 3) + 8 * (reg >> 2);
          *reinterpret_cast<float*>(wpos + rl * S::FPITCH) = (acc[2 * r + ib][reg] + cbias) * cscale + cshift;
        }
      __builtin_amdgcn_wave_barrier();
      if (chunk_ok) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const char* const rp_ = stage + (it * 16 + lpx) * S::FPITCH + ch * 32;
          const f32x4 va = *reinterpret_cast<const f32x4*>(rp_);
          const f32x4 vb = *reinterpret_cast<const f32x4*>(rp_ + 16);
          float f[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[2 * e] += __uint_as_float(((unsigned)rv[it][e]) << 16);
            f[2 * e + 1] += __uint_as_float(((unsigned)rv[it][e]) & 0xFFFF0000u);
          }
          if (P.leaky) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], f[e] * P.slope);
          }
          i32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            pk[e] = (int)__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){f[2 * e], f[2 * e + 1]}, bf16x2_t));
          *reinterpret_cast<i32x4*>(const_cast<bf16_t*>(group_ptr(P.out, 32 * (i0 + 2 * r) + 16 * it))) = pk;
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float q0 = __uint_as_float(((unsigned)pk[e]) << 16), q1 = __uint_as_float(((unsigned)pk[e]) & 0xFFFF0000u);
              s1[2 * e] += q0; s2[2 * e] = fmaf(q0, q0, s2[2 * e]);
              s1[2 * e + 1] += q1; s2[2 * e + 1] = fmaf(q1, q1, s2[2 * e + 1]);
            }
          }
        }
      }
    }
    if (want_stats) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o); s2[e] += __shfl_xor(s2[e], o); }
      }
      if (lane < 4 && chunk_ok) {
        const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * P.Cout;
#pragma unroll
        for (int e = 0; e < 8; ++e) { atomicAdd(P.stat_sum + rep + ccol + e, s1[e]); atomicAdd(P.stat_sq + rep + ccol + e, s2[e]); }
      }
    }
  }
  CONV_STAMP(5);
}

// GEMM operand [Cout, 9 * Cin] (tap-major, channels inside: autoencoder.py:_w_op) -> the fragment-order stream convq_kernel's
// DMA reads: [pass = Cin / 64][tap 9][k-step 4][Cout block of 32][lane 64][8]: lane (l31, hh) of fragment (pass, tap, kk, j) holds
// W[32 j + l31][tap * Cin + 64 pass + 16 kk + 8 hh + 0..7]; rows past Cout are zero.
__global__ __launch_bounds__(256) void convq_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ dst, int Cin, int Cout,
                                                         int ncb, long nchunk) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunk) return;
  const int lane = (int)(i & 63);
  long r = i >> 6;
  const int j = (int)(r % ncb); r /= ncb;
  const int kk = (int)(r & 3); r >>= 2;
  const int tap = (int)(r % 9);
  const int pass = (int)(r / 9);
  const int co = 32 * j + (lane & 31);
  const int k = tap * Cin + 64 * pass + 16 * kk + 8 * (lane >> 5);
  i32x4 v = (i32x4)(0);
  if (co < Cout) v = *reinterpret_cast<const i32x4*>(w + (long)co * 9 * Cin + k);
  *reinterpret_cast<i32x4*>(dst + i * 8) = v;
}

static int g_conv_skew = 1, g_conv_dbg = 0;

}  // namespace

#ifdef WMZ_CONV_STAMPS
extern "C" int wmz_debug_conv_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(conv_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" int wmz_debug_conv_knobs(int skew, int dbg) { g_conv_skew = skew; g_conv_dbg = dbg; return WMZ_OK; }

extern "C" int wmz_conv3x3_direct_supported(int H, int W, int Cin, int Cout) {
  if ((Cin != 64 && Cin != 128) || Cout <= 0 || Cout > 128 || (Cout & 7) != 0) return 0;
  if (W >= 32 && (W & 31) == 0 && (H & 7) == 0) return 1;
  if (W == 16 && (H & 15) == 0) return 1;
  return 0;
}

extern "C" long wmz_conv3x3_direct_pack_elems(int Cin, int Cout) {
  const int ncb = Cout <= 64 ? 2 : 4;
  return (long)9 * Cin * ncb * 32;
}

extern "C" int wmz_conv3x3_direct_pack(const void* w_op, void* wpack, int Cin, int Cout, void* stream) {
  WMZ_REQUIRE(w_op && wpack, "wmz_conv3x3_direct_pack: null tensor");
  WMZ_REQUIRE(Cin > 0 && (Cin & 63) == 0 && Cout > 0 && Cout <= 128, "wmz_conv3x3_direct_pack: Cin %% 64 == 0 and Cout <= 128 required (got %d, %d)", Cin, Cout);
  const int ncb = Cout <= 64 ? 2 : 4;
  const long nchunk = wmz_conv3x3_direct_pack_elems(Cin, Cout) / 8;
  hipLaunchKernelGGL(convq_pack_kernel, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w_op, (bf16_t*)wpack, Cin, Cout, ncb, nchunk);
  WMZ_LAUNCH_CHECK("wmz_conv3x3_direct_pack");
  return WMZ_OK;
}

// stride 2 (pad 1): H, W the INPUT plane (even); the output is H / 2 x W / 2.  Cout = 128, output planes of 8 k x 16 m pixels.
extern "C" int wmz_conv3x3_direct_supported_strided(int H, int W, int Cin, int Cout, int stride) {
  if (stride == 1) return wmz_conv3x3_direct_supported(H, W, Cin, Cout);
  if (stride != 2 || (Cin != 64 && Cin != 128) || Cout != 128 || (H & 1) || (W & 1)) return 0;
  return ((H / 2) & 7) == 0 && ((W / 2) & 15) == 0;
}

extern "C" int wmz_conv3x3_direct_fwd_strided(const void* x, const void* wpack, void* out, const float* bias, const float* scale,
                                              const float* shift, const void* residual, float* stat_sum, float* stat_sq, int B,
                                              int H, int W, int Cin, int Cout, int stride, int leaky, float slope, void* stream) {
  if (stride == 1)
    return wmz_conv3x3_direct_fwd(x, wpack, out, bias, scale, shift, residual, stat_sum, stat_sq, B, H, W, Cin, Cout, leaky, slope, stream);
  WMZ_REQUIRE(x && wpack && out, "wmz_conv3x3_direct_fwd_strided: null tensor");
  WMZ_REQUIRE(B > 0 && wmz_conv3x3_direct_supported_strided(H, W, Cin, Cout, stride),
              "wmz_conv3x3_direct_fwd_strided: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d stride=%d", B, H, W, Cin, Cout, stride);
  WMZ_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "wmz_conv3x3_direct_fwd_strided: stat_sum and stat_sq go together");
  WMZ_REQUIRE((scale == nullptr) == (shift == nullptr), "wmz_conv3x3_direct_fwd_strided: scale and shift go together");
  WMZ_REQUIRE(slope >= 0.f && slope <= 1.f, "wmz_conv3x3_direct_fwd_strided: LeakyReLU slope in [0, 1] expected");
  DirectParams P;
  P.x = (const bf16_t*)x; P.wpack = (const bf16_t*)wpack; P.out = (bf16_t*)out;
  P.bias = bias; P.scale = scale; P.shift = shift; P.res = (const bf16_t*)residual;
  P.stat_sum = stat_sum; P.stat_sq = stat_sq;
  P.B = B; P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout; P.Ho = H / 2; P.Wo = W / 2;
  P.slope = slope; P.leaky = leaky;
  P.skew = g_conv_skew; P.dbg = g_conv_dbg & 1;
  P.ncb_pack = 4;
  P.tiles_x = P.Wo / 16; P.tiles_y = P.Ho / 8;
  const long tiles = (long)B * P.tiles_x * P.tiles_y;
  WMZ_REQUIRE(tiles < (1L << 31), "wmz_conv3x3_direct_fwd_strided: too many tiles");
  dim3 grid((unsigned)tiles), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64) hipLaunchKernelGGL((convr_kernel<4, 16, 1, 2>), grid, block, 0, st, P);      // <.., NPASS = Cin / 64, STRIDE>
  else hipLaunchKernelGGL((convr_kernel<4, 16, 2, 2>), grid, block, 0, st, P);
  WMZ_LAUNCH_CHECK("wmz_conv3x3_direct_fwd_strided");
  return WMZ_OK;
}

extern "C" int wmz_conv3x3_direct_fwd(const void* x, const void* wpack, void* out, const float* bias, const float* scale,
                                      const float* shift, const void* residual, float* stat_sum, float* stat_sq, int B, int H,
                                      int W, int Cin, int Cout, int leaky, float slope, void* stream) {
  WMZ_REQUIRE(x && wpack && out, "wmz_conv3x3_direct_fwd: null tensor");
  WMZ_REQUIRE(B > 0 && wmz_conv3x3_direct_supported(H, W, Cin, Cout),
              "wmz_conv3x3_direct_fwd: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d (wmz_conv3x3_direct_supported)", B, H, W, Cin, Cout);
  WMZ_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "wmz_conv3x3_direct_fwd: stat_sum and stat_sq go together");
  WMZ_REQUIRE((scale == nullptr) == (shift == nullptr), "wmz_conv3x3_direct_fwd: scale and shift go together");
  DirectParams P;
  P.x = (const bf16_t*)x; P.wpack = (const bf16_t*)wpack; P.out = (bf16_t*)out;
  P.bias = bias; P.scale = scale; P.shift = shift; P.res = (const bf16_t*)residual;
  P.stat_sum = stat_sum; P.stat_sq = stat_sq;
  P.B = B; P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout; P.Ho = H; P.Wo = W;
  P.slope = slope; P.leaky = leaky;
  P.skew = g_conv_skew; P.dbg = g_conv_dbg;
  const bool wide = W >= 32;
  P.tiles_x = wide ? W / 32 : 1;
  P.tiles_y = wide ? H / 8 : H / 16;
  const long tiles = (long)B * P.tiles_x * P.tiles_y;
  WMZ_REQUIRE(tiles < (1L << 31), "wmz_conv3x3_direct_fwd: too many tiles");
  dim3 grid((unsigned)tiles), block(256);
  hipStream_t st = (hipStream_t)stream;
  P.ncb_pack = Cout <= 64 ? 2 : 4;
  if (g_conv_dbg & 16) {                                           // development: the shared-slab kernel (A/B timing)
    if (Cout <= 64) {
      if (wide) hipLaunchKernelGGL((convq_kernel<2, 32>), grid, block, 0, st, P);
      else hipLaunchKernelGGL((convq_kernel<2, 16>), grid, block, 0, st, P);
    } else {
      if (wide) hipLaunchKernelGGL((convq_kernel<4, 32>), grid, block, 0, st, P);
      else hipLaunchKernelGGL((convq_kernel<4, 16>), grid, block, 0, st, P);
    }
  } else if (Cout <= 32) {
    if (wide) { if (Cin == 64) hipLaunchKernelGGL((convr_kernel<1, 32, 1>), grid, block, 0, st, P); else hipLaunchKernelGGL((convr_kernel<1, 32, 2>), grid, block, 0, st, P); }
    else { if (Cin == 64) hipLaunchKernelGGL((convr_kernel<1, 16, 1>), grid, block, 0, st, P); else hipLaunchKernelGGL((convr_kernel<1, 16, 2>), grid, block, 0, st, P); }
  } else if (Cout <= 64) {
    if (wide) { if (Cin == 64) hipLaunchKernelGGL((convr_kernel<2, 32, 1>), grid, block, 0, st, P); else hipLaunchKernelGGL((convr_kernel<2, 32, 2>), grid, block, 0, st, P); }
    else { if (Cin == 64) hipLaunchKernelGGL((convr_kernel<2, 16, 1>), grid, block, 0, st, P); else hipLaunchKernelGGL((convr_kernel<2, 16, 2>), grid, block, 0, st, P); }
  } else {
    if (wide) { if (Cin == 64) hipLaunchKernelGGL((convr_kernel<4, 32, 1>), grid, block, 0, st, P); else hipLaunchKernelGGL((convr_kernel<4, 32, 2>), grid, block, 0, st, P); }
    else { if (Cin == 64) hipLaunchKernelGGL((convr_kernel<4, 16, 1>), grid, block, 0, st, P); else hipLaunchKernelGGL((convr_kernel<4, 16, 2>), grid, block, 0, st, P); }
  }
  WMZ_LAUNCH_CHECK("wmz_conv3x3_direct_fwd");
  return WMZ_OK;
}
