// Small-K convolutions of the VQ auto-encoder in bf16, NHWC, for gfx950 (vq-video-diffusion/autoencoder.py): the 1x1 convolutions
// of Residual (:18-42, with the training-mode BatchNorm + LeakyReLU of the layer in front as an input prologue) and UpscaleResidual's
// conv_residual (:89-131), the 2x2 / stride 2 down-sampling convolution (:29-33) and the 3-channel conv_1 (:60-86) -- K = KH KW Cin
// <= 256.  These layers are memory-bound (0.3-1 flop per byte of the roofline's ridge): the job is to stream the tensors.
//
//   * persistent workgroups (grid = a few per CU), every WAVE on its own: a wave takes runs of 64 consecutive output pixels x all
//     output channels; nothing is shared between waves but the weights, which sit in LDS in MFMA fragment order for the whole launch
//     (<= 32 KB, loaded once per workgroup: ONE barrier per launch; conv2d.hip pays ~12 per 256-pixel tile);
//   * the im2col rows are never built: per 64-channel chunk a lane fetches 16-byte granules whose source address it computes (tap,
//     channel group, padding), coalesced (8 lanes = one 128-byte line), applies the optional prologue in registers, and writes them
//     to the wave's private LDS image (pixel pitch 144 bytes: conflict-free ds_write_b128 / ds_read_b128); the next chunk's granules
//     are in flight while the current chunk multiplies (MFMA 32x32x16, 2 x NCB blocks);
//   * epilogue per run in the accumulator layout (a lane owns a channel: bias / folded BatchNorm / LeakyReLU / rounding, the
//     BatchNorm statistics as lane-local sums kept in registers ACROSS runs -- one atomic per channel and workgroup per launch),
//     the bf16 tile transposed through the same LDS image into 16-byte stores of whole contiguous rows.
// Same arithmetic as conv2d.hip's kernel (k order, epilogue): the same bits; statistics to fp32 summation order.
#include "wmz_common.h"
#include "bn_lazy.h"

namespace {

struct PointParams {
  const bf16_t* x; const bf16_t* wpack; bf16_t* out;
  const float* bias; const float* scale; const float* shift;
  float* stat_sum; float* stat_sq;
  const float* in_scale; const float* in_shift; float in_slope;
  int Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int nq;            // 64-channel chunks of the (zero-padded) reduction: ceil(KH KW Cin / 64)
  int kreal;         // KH KW Cin: the k-steps of the last chunk behind it multiply zeros and are skipped
  int gpt;           // 16-byte granules per tap = Cin / 8
  int nruns;         // M / 64
  float slope; int leaky;
  BnStats in_bn;     // the input prologue's BatchNorm from raw statistics (sum != nullptr; bn_lazy.h) instead of in_scale / in_shift
};

__device__ __attribute__((aligned(16))) unsigned cp_zero_chunk[4] = {0u, 0u, 0u, 0u};      // what a padding / empty granule loads

constexpr int CP_PITCH = 144;                    // bytes per pixel of a wave's LDS image: 64 channels + 16
constexpr int CP_IMG = 64 * CP_PITCH;            // one wave's image: 64 pixels

// PAD: the layer has zero padding (conv_1, the 3-channel data gradient): per-pixel bounds checks; without it every tap of every
// output pixel is inside the image and the lane keeps only its eight input offsets.
// NPB: 32-pixel blocks per run -- 2 (64 pixels) at Cout <= 64, 1 at Cout <= 128: 64 accumulator registers either way.
template <int NCB, int NPB, bool PAD>
__global__ __launch_bounds__(256, 2) void convp_kernel(PointParams P) {
  constexpr int RPX = 32 * NPB, NLD = 4 * NPB;                     // pixels per run; granules a lane stages per chunk
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  // layout: [weights: nq * 4 * NCB KB][prologue table: 2 * 128 floats][4 wave images]
  const int wbytes = P.nq * 4 * NCB * 1024;
  char* const wlds = lds;
  float* const ptab = reinterpret_cast<float*>(lds + wbytes);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const img = lds + wbytes + 1024 + wave * CP_IMG;
  const int l31 = lane & 31, hh = lane >> 5;
  const bool pre = P.in_scale != nullptr || P.in_bn.sum != nullptr;

  // ---- once per workgroup: weights (fragment order, contiguous) and the prologue's per-channel constants into LDS
  for (int i = tid; i < wbytes / 16; i += 256)
    *reinterpret_cast<i32x4*>(wlds + i * 16) = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(P.wpack) + (long)i * 16);
  if (pre && tid < P.Cin) {
    float sc, sh;
    if (P.in_bn.sum != nullptr) bn_channel(P.in_bn, P.Cin, tid, blockIdx.x == 0, sc, sh);      // (workgroup 0 publishes)
    else { sc = P.in_scale[tid]; sh = P.in_shift[tid]; }
    ptab[tid] = sc; ptab[128 + tid] = sh;
  }
  // the epilogue's per-channel constants (a lane owns channel 32 j + l31 of every block j)
  float cbias[NCB], cscale[NCB], cshift[NCB];
#pragma unroll
  for (int j = 0; j < NCB; ++j) {
    const int col = 32 * j + l31;
    const bool cok = col < P.Cout;
    cbias[j] = (cok && P.bias) ? P.bias[col] : 0.f;
    cscale[j] = (cok && P.scale) ? P.scale[col] : 1.f;
    cshift[j] = (cok && P.shift) ? P.shift[col] : 0.f;
  }
  __syncthreads();

  // staging role of a lane: granule column gc of the chunk, rows sr + 8 i
  const int gc = lane & 7, sr = lane >> 3;
  const int ntaps = P.KH * P.KW;
  const char* const a_rd = img + l31 * CP_PITCH + hh * 16;                           // fragment reads: rows l31, l31 + 32
  char* const a_wr = img + sr * CP_PITCH + gc * 16;
  const char* const b_rd = wlds + lane * 16;
  const bool want_stats = P.stat_sum != nullptr;
  f32x2 s1[NCB], s2[NCB];
#pragma unroll
  for (int j = 0; j < NCB; ++j) { s1[j] = (f32x2){0.f, 0.f}; s2[j] = (f32x2){0.f, 0.f}; }

  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  // the 8 pixels this lane stages for a run: input offset of tap (0, 0) and the coordinates for the bounds checks
  int poff[NLD];
  unsigned pyx[PAD ? NLD : 1];                                        // (iy0 + 1) << 16 | (ix0 + 1): pad <= 1 keeps both non-negative
  auto setup = [&](int run) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int m = run * RPX + sr + 8 * i;
      const int ox = m % P.Wo, t = m / P.Wo;
      const int oy = t % P.Ho, b = t / P.Ho;
      const int iy0 = oy * P.stride - P.pad, ix0 = ox * P.stride - P.pad;
      poff[i] = ((b * P.Hi + iy0) * P.Wi + ix0) * P.Cin;
      if constexpr (PAD) pyx[i] = ((unsigned)(iy0 + 1) << 16) | (unsigned)(ix0 + 1);
    }
  };
  auto gather = [&](int qc, i32x4 (&ld)[NLD]) {
    const int g = qc * 8 + gc;                                      // granule of the im2col row
    const int tap = g / P.gpt, c8 = g - tap * P.gpt;
    const int kh = tap / P.KW, kw = tap - kh * P.KW;
    const int toff = (kh * P.Wi + kw) * P.Cin + c8 * 8;
    const bool tok = tap < ntaps;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      bool ok = tok;
      if constexpr (PAD) {
        const unsigned iy = (pyx[i] >> 16) + kh - 1, ix = (pyx[i] & 0xFFFFu) + kw - 1;       // (wraps to a huge value when negative)
        ok = ok && iy < (unsigned)P.Hi && ix < (unsigned)P.Wi;
      }
      if constexpr (PAD) {
        ld[i] = (i32x4)(0);
        if (ok) ld[i] = *reinterpret_cast<const i32x4*>(P.x + (long)(poff[i] + toff));
      } else {
        // (round 6: an unconditional load from a SELECTED address -- the granules behind the last tap read a zero chunk: the
        //  predicated form made hipcc branch around the loads; 1x1 128 -> 64 at 256 x 64^2: 103.9 -> 93.5 us.  The padded layers
        //  keep the predicate: there a third of the granules would hammer the one zero line, measured 9-15 % slower)
        const i32x4* const src = ok ? reinterpret_cast<const i32x4*>(P.x + (long)(poff[i] + toff)) : reinterpret_cast<const i32x4*>(cp_zero_chunk);
        ld[i] = *src;
      }
    }
  };
  i32x4 ld[NLD];
  if (gw < P.nruns) { setup(gw); gather(0, ld); }
  for (int run = gw; run < P.nruns; run += nw) {
    f32x16 acc[NPB][NCB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
      for (int j = 0; j < NCB; ++j) acc[i][j] = (f32x16)(0.f);

    for (int qc = 0; qc < P.nq; ++qc) {
      // ---- granules -> (prologue) -> the wave's LDS image
      if (pre) {
        const int c0 = (qc * 8 + gc) * 8;                           // 1x1 convolution: the granule's first channel
        const f32x4 sa = *reinterpret_cast<const f32x4*>(ptab + c0), sb = *reinterpret_cast<const f32x4*>(ptab + c0 + 4);
        const f32x4 ta = *reinterpret_cast<const f32x4*>(ptab + 128 + c0), tb = *reinterpret_cast<const f32x4*>(ptab + 128 + c0 + 4);
        const float sc[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
        const float sh[8] = {ta[0], ta[1], ta[2], ta[3], tb[0], tb[1], tb[2], tb[3]};
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y0 = fmaf(__uint_as_float(((unsigned)ld[i][e]) << 16), sc[2 * e], sh[2 * e]);
            float y1 = fmaf(__uint_as_float(((unsigned)ld[i][e]) & 0xFFFF0000u), sc[2 * e + 1], sh[2 * e + 1]);
            y0 = y0 > 0.f ? y0 : y0 * P.in_slope;
            y1 = y1 > 0.f ? y1 : y1 * P.in_slope;
            ld[i][e] = (int)((unsigned)f32_to_bf16_bits(y0) | ((unsigned)f32_to_bf16_bits(y1) << 16));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NLD; ++i) *reinterpret_cast<i32x4*>(a_wr + i * 8 * CP_PITCH) = ld[i];
      // the next chunk's granules -- or, behind the last chunk, the NEXT RUN's first -- are in flight while this chunk multiplies
      // (and the epilogue runs)
      if (qc + 1 < P.nq) gather(qc + 1, ld);
      else if (run + nw < P.nruns) { setup(run + nw); gather(0, ld); }
      __builtin_amdgcn_wave_barrier();
      // ---- 4 k-steps x (2 pixel blocks x NCB channel blocks)
      const char* const bq = b_rd + qc * 4 * NCB * 1024;
      // (conv_1: K = 3 x 3 x 8 = 72 -- the second chunk holds one tap: one k-step of its four)
      const int ksteps = __builtin_amdgcn_readfirstlane((P.kreal - qc * 64 + 15) >> 4);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk >= ksteps) break;
        s16x8 af[NPB], bf[NCB];
#pragma unroll
        for (int i = 0; i < NPB; ++i) af[i] = *reinterpret_cast<const s16x8*>(a_rd + i * 32 * CP_PITCH + kk * 32);
#pragma unroll
        for (int j = 0; j < NCB; ++j) bf[j] = *reinterpret_cast<const s16x8*>(bq + (kk * NCB + j) * 1024);
#pragma unroll
        for (int i = 0; i < NPB; ++i)
#pragma unroll
          for (int j = 0; j < NCB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        // (keeps hipcc from hoisting all four k-steps' fragments in front of the first MFMA: 16-24 fragments = 64-96 registers,
        //  which spilled; the LDS latency of a k-step is covered by the other waves of the CU -- the kernel is memory-bound)
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: accumulator layout -> bf16 -> the wave's image -> 16-byte chunks of whole rows
    constexpr int CH = NCB * 4;                                     // 16-byte chunks per output row
    constexpr int OP = NCB * 64 + 16;                               // output row pitch in the image
    static_assert(RPX * OP <= CP_IMG, "the output tile fits the wave's image");
    bf16_t* const orow = P.out + (long)run * RPX * P.Cout;
    auto finish = [&](auto affc, auto leakyc, auto statsc) {
      constexpr bool AFF = decltype(affc)::value, LEAKY = decltype(leakyc)::value, STATS = decltype(statsc)::value;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
          char* const wp = img + (32 * i + 4 * hh) * OP + (32 * j + l31) * 2;
#pragma unroll
          for (int rp = 0; rp < 8; ++rp) {
            const int reg = 2 * rp, rl = (reg & 3) + 8 * (reg >> 2);
            float v0 = acc[i][j][reg] + cbias[j], v1 = acc[i][j][reg + 1] + cbias[j];
            if constexpr (AFF) { v0 = v0 * cscale[j] + cshift[j]; v1 = v1 * cscale[j] + cshift[j]; }
            if constexpr (LEAKY) { v0 = fmaxf(v0, v0 * P.slope); v1 = fmaxf(v1, v1 * P.slope); }
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, bf16x2_t));
            *reinterpret_cast<unsigned short*>(wp + rl * OP) = (unsigned short)pk;
            *reinterpret_cast<unsigned short*>(wp + (rl + 1) * OP) = (unsigned short)(pk >> 16);
            if constexpr (STATS) {                                  // statistics of what the next stage will read
              const f32x2 q = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xFFFF0000u)};
              s1[j] += q;
              s2[j] = q * q + s2[j];
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int ch = lane & (CH - 1), lrow = lane / CH;
      if (ch * 8 < P.Cout) {
#pragma unroll
        for (int it = 0; it < RPX * CH / 64; ++it) {
          const int rl = it * (64 / CH) + lrow;
          const i32x4 v = *reinterpret_cast<const i32x4*>(img + rl * OP + ch * 16);
          *reinterpret_cast<i32x4*>(orow + (long)rl * P.Cout + ch * 8) = v;
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
    __builtin_amdgcn_wave_barrier();                                // (the image is the next run's staging space)
  }

  if (want_stats) {
    // lane-local sums of all this wave's runs -> channel sums of the workgroup (LDS) -> one atomic per channel
    __syncthreads();                                                // every wave is done with its image: reuse the first KB pair
    float* const red = reinterpret_cast<float*>(lds + wbytes + 1024);   // [2][NCB * 32]
    if (tid < 2 * NCB * 32) red[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const float t1 = wave_halves_sum(s1[j][0] + s1[j][1]), t2 = wave_halves_sum(s2[j][0] + s2[j][1]);
      if (hh == 0) { atomicAdd(&red[32 * j + l31], t1); atomicAdd(&red[NCB * 32 + 32 * j + l31], t2); }
    }
    __syncthreads();
    if (tid < NCB * 32 && tid < P.Cout) {
      const long rep = (long)(blockIdx.x % WMZ_STAT_REPLICAS) * P.Cout;
      atomicAdd(P.stat_sum + rep + tid, red[tid]);
      atomicAdd(P.stat_sq + rep + tid, red[NCB * 32 + tid]);
    }
  }
}

// GEMM operand [Cout, K] (K = KH KW Cin, tap-major) -> [k-step = ceil64(K) / 16][channel block ncb][lane 64][8]: lane (l31, hh) of
// fragment (ks, j) holds W[32 j + l31][16 ks + 8 hh + 0..7]; zero past Cout / K.
__global__ __launch_bounds__(256) void convp_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ dst, int K, int Cout, int ncb,
                                                         long nchunk) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunk) return;
  const int lane = (int)(i & 63);
  long r = i >> 6;
  const int j = (int)(r % ncb);
  const int ks = (int)(r / ncb);
  const int co = 32 * j + (lane & 31);
  const int k = 16 * ks + 8 * (lane >> 5);
  i32x4 v = (i32x4)(0);
  if (co < Cout && k < K) v = *reinterpret_cast<const i32x4*>(w + (long)co * K + k);        // (K % 8 == 0: whole granules)
  *reinterpret_cast<i32x4*>(dst + i * 8) = v;
}

// logical NCHW frames (contiguous, fp32 or bf16) -> NHWC with the channels zero-padded to a multiple of 8, in the compute dtype:
// the layout flip + pad + cast in front of conv_1 (autoencoder.py:83) as ONE pass (F.pad + copy + cast were three launches,
// 34 us for 256 frames of 64 x 64).  A thread owns one pixel and one group of 8 channels: plane reads coalesce along W, the
// write is one 16-byte (bf16) / two 16-byte (fp32) chunks.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int C8, long HW, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;              // pixel-major: consecutive threads = consecutive pixels
  if (i >= total) return;
  const int ngrp = C8 >> 3;
  const long pix = i % (total / ngrp);                               // (channel groups outermost: a warp stays in one plane set)
  const int grp = (int)(i / (total / ngrp));
  const long b = pix / HW, p = pix - b * HW;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = grp * 8 + e;
    f[e] = c < C ? Elem<TI>::to_f32(x[(b * C + c) * HW + p]) : 0.f;
  }
  TO* const dst = y + pix * C8 + grp * 8;
  if constexpr (sizeof(TO) == 2) {
    i32x4 pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = (int)((unsigned)f32_to_bf16_bits(f[2 * e]) | ((unsigned)f32_to_bf16_bits(f[2 * e + 1]) << 16));
    *reinterpret_cast<i32x4*>(dst) = pk;
  } else {
    *reinterpret_cast<f32x4*>(dst) = (f32x4){f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){f[4], f[5], f[6], f[7]};
  }
}

int point_ncb(int Cout) { return Cout <= 64 ? 2 : 4; }

}  // namespace

extern "C" int wmz_conv_point_supported(int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || (Cin & 7) != 0 || Cout <= 0 || Cout > 128 || (Cout & 7) != 0) return 0;
  if (KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || pad > 1 || Hi >= 32768 || Wi >= 32768) return 0;
  const long K = (long)KH * KW * Cin;
  if (K > 256) return 0;
  const int Ho = (Hi + 2 * pad - KH) / stride + 1, Wo = (Wi + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const long M = (long)B * Ho * Wo;
  if ((M & 63) != 0 || M * (Cout > Cin ? Cout : Cin) >= (1L << 31) || (long)B * Hi * Wi * Cin >= (1L << 31)) return 0;
  return 1;
}

extern "C" long wmz_conv_point_pack_elems(int K, int Cout) {
  return (long)((K + 63) / 64) * 64 * point_ncb(Cout) * 32;
}

extern "C" int wmz_conv_point_pack(const void* w_op, void* wpack, int K, int Cout, void* stream) {
  WMZ_REQUIRE(w_op && wpack && K > 0 && (K & 7) == 0 && K <= 256 && Cout > 0 && Cout <= 128, "wmz_conv_point_pack: bad arguments (K %d, Cout %d)", K, Cout);
  const long nchunk = wmz_conv_point_pack_elems(K, Cout) / 8;
  hipLaunchKernelGGL(convp_pack_kernel, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w_op,
                     (bf16_t*)wpack, K, Cout, point_ncb(Cout), nchunk);
  WMZ_LAUNCH_CHECK("wmz_conv_point_pack");
  return WMZ_OK;
}

// in_bn (optional, HOST pointer to a wmz_bn_stats): the prologue's BatchNorm from its raw batch statistics (replaces in_scale / in_shift)
extern "C" int wmz_conv_point_fwd_bn(const void* x, const void* wpack, void* out, const float* bias, const float* scale, const float* shift,
                                     float* stat_sum, float* stat_sq, const float* in_scale, const float* in_shift,
                                     const wmz_bn_stats* in_bn, float in_slope, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                                     int stride, int pad, int leaky, float slope, void* stream) {
  WMZ_REQUIRE(x && wpack && out, "wmz_conv_point_fwd: null tensor");
  WMZ_REQUIRE(wmz_conv_point_supported(B, Hi, Wi, Cin, Cout, KH, KW, stride, pad), "wmz_conv_point_fwd: unsupported shape (wmz_conv_point_supported)");
  WMZ_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "wmz_conv_point_fwd: stat_sum and stat_sq go together");
  WMZ_REQUIRE((scale == nullptr) == (shift == nullptr), "wmz_conv_point_fwd: scale and shift go together");
  WMZ_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wmz_conv_point_fwd: in_scale and in_shift go together");
  WMZ_REQUIRE(bn_stats_ok(in_bn) && (in_bn == nullptr || in_scale == nullptr), "wmz_conv_point_fwd_bn: incomplete wmz_bn_stats (or both prologue forms given)");
  WMZ_REQUIRE((in_scale == nullptr && in_bn == nullptr) || (KH == 1 && KW == 1 && pad == 0 && Cin <= 128), "wmz_conv_point_fwd: the input prologue is built for 1x1 convolutions of <= 128 channels");
  WMZ_REQUIRE(slope >= 0.f && slope <= 1.f, "wmz_conv_point_fwd: LeakyReLU slope in [0, 1] expected");
  PointParams P;
  P.x = (const bf16_t*)x; P.wpack = (const bf16_t*)wpack; P.out = (bf16_t*)out;
  P.bias = bias; P.scale = scale; P.shift = shift; P.stat_sum = stat_sum; P.stat_sq = stat_sq;
  P.in_scale = in_scale; P.in_shift = in_shift; P.in_slope = in_slope;
  P.in_bn = bn_stats_from(in_bn);
  P.Hi = Hi; P.Wi = Wi; P.Cin = Cin; P.Cout = Cout; P.KH = KH; P.KW = KW; P.stride = stride; P.pad = pad;
  P.Ho = (Hi + 2 * pad - KH) / stride + 1;
  P.Wo = (Wi + 2 * pad - KW) / stride + 1;
  const int K = KH * KW * Cin;
  P.nq = (K + 63) / 64; P.gpt = Cin / 8; P.kreal = K;
  const int ncb = point_ncb(Cout);
  P.nruns = (int)((long)B * P.Ho * P.Wo / (ncb == 2 ? 64 : 32));
  P.slope = slope; P.leaky = leaky;
  const int lds_bytes = P.nq * 4 * ncb * 1024 + 1024 + 4 * CP_IMG;
  const int per_cu = lds_bytes <= 81920 ? 2 : 1;
  int grid = 256 * per_cu;
  const int need = (P.nruns + 3) / 4;
  if (grid > need) grid = need;
  hipStream_t st = (hipStream_t)stream;
  static std::atomic<uint64_t> attr_devs{0};                       // (> 64 KB of dynamic LDS has to be asked for once per device)
  if (wmz_first_use_on_device(attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convp_kernel<2, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convp_kernel<4, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convp_kernel<2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convp_kernel<4, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const bool padded = pad > 0;
  if (ncb == 2) {
    if (padded) hipLaunchKernelGGL((convp_kernel<2, 2, true>), dim3(grid), dim3(256), lds_bytes, st, P);
    else hipLaunchKernelGGL((convp_kernel<2, 2, false>), dim3(grid), dim3(256), lds_bytes, st, P);
  } else {
    if (padded) hipLaunchKernelGGL((convp_kernel<4, 1, true>), dim3(grid), dim3(256), lds_bytes, st, P);
    else hipLaunchKernelGGL((convp_kernel<4, 1, false>), dim3(grid), dim3(256), lds_bytes, st, P);
  }
  WMZ_LAUNCH_CHECK("wmz_conv_point_fwd");
  return WMZ_OK;
}

extern "C" int wmz_conv_point_fwd(const void* x, const void* wpack, void* out, const float* bias, const float* scale, const float* shift,
                                  float* stat_sum, float* stat_sq, const float* in_scale, const float* in_shift, float in_slope, int B,
                                  int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int leaky, float slope,
                                  void* stream) {
  return wmz_conv_point_fwd_bn(x, wpack, out, bias, scale, shift, stat_sum, stat_sq, in_scale, in_shift, nullptr, in_slope, B, Hi, Wi, Cin,
                               Cout, KH, KW, stride, pad, leaky, slope, stream);
}

extern "C" int wmz_nchw_to_nhwc8(const void* x, void* y, int B, int C, int H, int W, int in_dtype, int out_dtype, void* stream) {
  WMZ_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "wmz_nchw_to_nhwc8: bad arguments");
  WMZ_REQUIRE((in_dtype == WMZ_F32 || in_dtype == WMZ_BF16) && (out_dtype == WMZ_F32 || out_dtype == WMZ_BF16), "wmz_nchw_to_nhwc8: bad dtype");
  const int C8 = (C + 7) & ~7;
  const long HW = (long)H * W, total = (long)B * HW * (C8 >> 3);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == WMZ_F32 && out_dtype == WMZ_BF16) hipLaunchKernelGGL((nchw_to_nhwc8_kernel<float, bf16_t>), grid, block, 0, st, (const float*)x, (bf16_t*)y, C, C8, HW, total);
  else if (in_dtype == WMZ_F32) hipLaunchKernelGGL((nchw_to_nhwc8_kernel<float, float>), grid, block, 0, st, (const float*)x, (float*)y, C, C8, HW, total);
  else if (out_dtype == WMZ_BF16) hipLaunchKernelGGL((nchw_to_nhwc8_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, C, C8, HW, total);
  else hipLaunchKernelGGL((nchw_to_nhwc8_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)x, (float*)y, C, C8, HW, total);
  WMZ_LAUNCH_CHECK("wmz_nchw_to_nhwc8");
  return WMZ_OK;
}
