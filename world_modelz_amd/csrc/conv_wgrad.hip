// Direct weight gradient of the VQ auto-encoder's 3x3 / stride 1 / pad 1 convolutions in bf16 (VQ-AE training, train_vqae.py:125-192;
// the reference gets it from autograd): dW[co, tap, ci] = sum over pixels dy[b, y, x, co] * x[b, y + kh - 1, x + kw - 1, ci].
//
// linear_bwd.hip's wgrad2_kernel treats it as a plain GEMM dy^T . im2col(x): every 128 x 128 output tile (one tap) re-reads dy and
// re-gathers x through L2 -- 18 passes over the two tensors per layer, 0.3 PFLOP/s (258 us for the 128 -> 128 layers of 64 frames
// of 64 x 64, a third of the VQ-AE training step).  Here the REDUCTION dimension (pixels) is tiled instead and all nine taps are
// computed from one staged copy:
//   * a persistent workgroup (one per CU, 8 waves) walks tiles of 8 x 16 pixels of one image; per tile the haloed 10 x 18 patch of
//     x (64 input channels) and the 128 pixels x 128 channels of dy go to LDS ONCE by global_load_lds, double-buffered -- the next
//     tile lands while this one multiplies (one barrier per tile = per 72 MFMAs of a wave);
//   * a wave owns 32 output channels x 32 input channels x 9 taps = 9 accumulator blocks of 32 x 32 (144 registers, all in the AGPR
//     half) for the WHOLE launch; a tile row of 16 pixels is one k-step: the dy fragment (pixels on the reduction axis: transposing
//     ds_read_b64_tr_b16 reads of the pixel-major image) is read once and multiplied against the 9 shifted x fragments -- the taps
//     are address offsets into the same patch (10 fragment reads per 9 MFMAs; all offsets instruction immediates, unrolled).
//     (Round 5 gave a wave 64 input channels = 18 blocks = 288 registers at one wave per SIMD: more than the 256 AGPRs, and hipcc
//     kept every MFMA in its AGPR form and ROTATED accumulator blocks between the two register halves -- 2 753 v_accvgpr moves
//     for 144 MFMAs per tile, 19 per MFMA: the kernel issued moves 61 % of the time at MFMA-busy 0.15.  Two waves per SIMD with half
//     the blocks each need none.)
//   * Cin = 128: two groups of workgroups, one per half of the input channels;
//   * Cout <= 32 (the decoder's 3-channel last layer, autoencoder.py:134-152 -- 524 us on wgrad2_kernel, whose 128-wide tile holds 8
//     useful columns): the NCOB = 1 form -- two waves per workgroup, a wave owns one block of 32 input channels x 9 taps against the
//     one (zero-padded) block of output channels;
//   * each workgroup stores its partial result once, in the [block][split][256] layout linear_bwd.hip's wgrad_reduce_kernel sums
//     (deterministic two-stage reduction, nn.Conv2d gradient layout and bias gradient included).
#include "wmz_common.h"
#include "wmz_internal.h"

namespace {

typedef const __attribute__((address_space(1))) void* cw_gptr_t;
typedef __attribute__((address_space(3))) void* cw_lptr_t;

__device__ __attribute__((aligned(16))) unsigned cw_zero_chunk[4] = {0u, 0u, 0u, 0u};

struct CwParams {
  const bf16_t* x; const bf16_t* dy; float* ws;
  int H, W, Cin, Cout, nsplit, ncig, tiles_x, tiles_y, ntiles;
};

constexpr int CW_PW = 18, CW_NPX = 180;           // haloed patch of an 8 x 16 tile
constexpr int CW_XIMG = 23 * 1024;                // 180 pixels x 128 bytes (64 channels) = 23040 -> 23 DMA pieces
constexpr int CW_WIN = 8;                         // fragments in flight

// NCOB: blocks of 32 output channels (4: Cout = 128, eight waves = four output blocks x two blocks of 32 input channels; 1: Cout <= 32,
// two waves, one block of input channels each)
template <int NCOB>
__global__ __launch_bounds__(NCOB == 4 ? 512 : 128, NCOB == 4 ? 1 : 2) void convw_kernel(CwParams P) {
  constexpr int NW = NCOB == 4 ? 8 : 2;                             // waves
  constexpr int NB = 9;                                             // accumulator blocks of a wave: the taps of its (output, input) channel block pair
  constexpr int DROW = NCOB == 4 ? 256 : 64;                        // bytes per pixel of the dy image (128 | 32 channels)
  constexpr int CW_DIMG = 128 * DROW;
  constexpr int CW_BUF = CW_XIMG + CW_DIMG;
  constexpr int CW_NITEM = 8 * (NB + 1);                            // fragment sequence of a tile: per tile row one dy fragment + NB x fragments
  __shared__ __attribute__((aligned(1024))) char lds[2 * CW_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cig = blockIdx.x % P.ncig, split = blockIdx.x / P.ncig;
  const int l31 = lane & 31, hh = lane >> 5;
  const int cob = NCOB == 4 ? (wave & 3) : 0, cib = NCOB == 4 ? (wave >> 2) : wave;   // this wave's output / input channel block

  auto issue_tile = [&](int t, int buf) {
    const int tx = t % P.tiles_x, r1 = t / P.tiles_x;
    const int ty = r1 % P.tiles_y, img = r1 / P.tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 16;
    char* const xi = lds + buf * CW_BUF;
    char* const di = xi + CW_XIMG;
    const bf16_t* const xb = P.x + (long)img * P.H * P.W * P.Cin + cig * 64;
    for (int pc = wave; pc < CW_XIMG / 1024; pc += NW) {
      const int q = pc * 64 + lane;
      const int r = q >> 3, s = q & 7;                              // patch pixel, 16-byte slot of its 128 bytes
      const int py = r / CW_PW, px = r - py * CW_PW;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      const bool ok = r < CW_NPX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      const int c = s ^ (((r >> 1) & 1) << 2);                      // the image's 64-byte swizzle, applied on the source side
      const void* src = ok ? (const void*)(xb + ((long)iy * P.W + ix) * P.Cin + c * 8) : (const void*)cw_zero_chunk;
      __builtin_amdgcn_global_load_lds((cw_gptr_t)src, (cw_lptr_t)(xi + pc * 1024), 16, 0, 0);
    }
    const bf16_t* const db = P.dy + (long)img * P.H * P.W * P.Cout;
    for (int pc = wave; pc < CW_DIMG / 1024; pc += NW) {
      const int q = pc * 64 + lane;
      if constexpr (NCOB == 4) {
        const int r = q >> 4, s = q & 15;                           // tile pixel (row-major 8 x 16), slot of its 256 bytes
        const int c = s ^ ((r & 3) << 2);
        const void* src = db + ((long)(oy0 + (r >> 4)) * P.W + ox0 + (r & 15)) * 128 + c * 8;
        __builtin_amdgcn_global_load_lds((cw_gptr_t)src, (cw_lptr_t)(di + pc * 1024), 16, 0, 0);
      } else {
        const int r = q >> 2, s = q & 3;                            // 64 bytes = 32 channels per pixel: the real ones, then zeros
        const void* src = s * 8 < P.Cout ? (const void*)(db + ((long)(oy0 + (r >> 4)) * P.W + ox0 + (r & 15)) * P.Cout + s * 8)
                                         : (const void*)cw_zero_chunk;
        __builtin_amdgcn_global_load_lds((cw_gptr_t)src, (cw_lptr_t)(di + pc * 1024), 16, 0, 0);
      }
    }
  };

  // ---- transposed-fragment addressing (linear_bwd.hip wg_col_frag): element j of the operand = image[m0 + 8 (lane >> 5) + j][c0 +
  // (lane & 31)], read as two ds_read_b64_tr_b16 (rows .. + 0-3 and + 4-7); a 16-lane group covers 4 rows x 16 columns
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int r0 = 8 * (gi >> 1) + q4;
  // dy image: rows of 256 bytes, 64-byte granule XOR (row & 3) -- the tile rows start at multiples of 16: (row & 3) = q4
  // (NCOB = 1: rows of 64 bytes = one granule: four consecutive rows already cover the 64 banks)
  const unsigned offA = NCOB == 4 ? (unsigned)(r0 * 256 + (((32 * cob + 16 * (gi & 1) + 4 * p4) * 2) ^ (q4 << 6)))
                                  : (unsigned)(r0 * 64 + (16 * (gi & 1) + 4 * p4) * 2);
  // x image: rows of 128 bytes, 64-byte granule XOR ((row >> 1) & 1): a fragment starting at patch pixel m0 reads rows m0 + r0 (+ 4);
  // with v = m0 & 3 the lane's offset is offB[v][channel block] + 128 (m0 - v) -- the second term an immediate
  unsigned offB[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = r0 + v;
    offB[v] = (unsigned)(r * 128 + (((32 * cib + 16 * (gi & 1) + 4 * p4) * 2) ^ (((r >> 1) & 1) << 6)));
  }

  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.f);
  float bias_acc = 0.f;

  int t = split;
  if (t < P.ntiles) issue_tile(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int it = 0; t < P.ntiles; t += P.nsplit, ++it) {
    if (t + P.nsplit < P.ntiles) issue_tile(t + P.nsplit, (it + 1) & 1);    // (that buffer: every wave left it at the last barrier)
    const unsigned xbase = lds_addr(lds + (it & 1) * CW_BUF);
    const unsigned aA = xbase + CW_XIMG + offA;
    unsigned aB[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) aB[v] = xbase + offB[v];

    // fragment queue: item n = 10 ty + j; j = 0: the dy fragment of tile row ty, j = 1 + tap: the x fragment of that tap for the row.
    // CW_WIN items (two reads each) in flight; every wait names the registers it releases.
    s16x4 ar[2][2], br[CW_WIN][2];
    auto issue = [&](auto nc) {
      constexpr int n = decltype(nc)::value, ty = n / (NB + 1), j = n % (NB + 1);
      if constexpr (j == 0) {
        ar[ty & 1][0] = ds_read_tr16_asm<ty * 16 * DROW>(aA);
        ar[ty & 1][1] = ds_read_tr16_asm<ty * 16 * DROW + 4 * DROW>(aA);
      } else {
        constexpr int b = j - 1, tap = b, kh = tap / 3, kw = tap % 3;
        constexpr int m0 = (ty + kh) * CW_PW + kw, v = m0 & 3, imm = (m0 - v) * 128, slot = (ty * NB + b) % CW_WIN;
        br[slot][0] = ds_read_tr16_asm<imm>(aB[v]);
        br[slot][1] = ds_read_tr16_asm<imm + 512>(aB[v]);
      }
    };
    static_for<CW_WIN>([&](auto nc) { issue(nc); });
    static_for<CW_NITEM>([&](auto nc) {
      constexpr int n = decltype(nc)::value, ty = n / (NB + 1), j = n % (NB + 1);
      constexpr int younger = (CW_NITEM - 1 - n) < (CW_WIN - 1) ? (CW_NITEM - 1 - n) : (CW_WIN - 1);
      if constexpr (j == 0) {
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ar[ty & 1][0]), "+v"(ar[ty & 1][1]) : "n"(2 * younger) : "memory");
        // bias gradient: this lane's 8 pixels of its output channel
        const s16x8 af = __builtin_shufflevector(ar[ty & 1][0], ar[ty & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_acc += bf16_bits_to_f32((unsigned short)af[e]);
      } else {
        constexpr int b = j - 1, slot = (ty * NB + b) % CW_WIN;
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(br[slot][0]), "+v"(br[slot][1]) : "n"(2 * younger) : "memory");
        const s16x8 af = __builtin_shufflevector(ar[ty & 1][0], ar[ty & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 bf = __builtin_shufflevector(br[slot][0], br[slot][1], 0, 1, 2, 3, 4, 5, 6, 7);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[b], 0, 0, 0);
      }
      if constexpr (n + CW_WIN < CW_NITEM) issue(std::integral_constant<int, n + CW_WIN>{});
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the next tile landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                   // ... everyone's; and everyone is done with this tile
  }

  // ---- this workgroup's partial result -> workspace[block of 256][split][256] (the layout wgrad_reduce_kernel sums)
  const int K = 9 * P.Cin;
  const long NK = (long)P.Cout * K;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int tap = b;
    const int k = tap * P.Cin + 64 * cig + 32 * cib + l31;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int co = 32 * cob + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
      const long q = (long)co * K + k;
      if (co < P.Cout) P.ws[(((q >> 8) * P.nsplit + split) << 8) + (q & 255)] = acc[b][reg];
    }
  }
  if (cig == 0 && cib == 0) {
    const float tb = wave_halves_sum(bias_acc);
    const int co = 32 * cob + l31;
    if (hh == 0 && co < P.Cout) P.ws[(((NK + 255) >> 8) * P.nsplit << 8) + (long)split * P.Cout + co] = tb;
  }
}

int convw_nsplit(int B, int H, int W, int Cin) {
  const int ncig = Cin / 64;
  const long ntiles = (long)B * (H / 8) * (W / 16);
  long ns = 256 / ncig;
  if (ns > ntiles) ns = ntiles;
  return (int)ns;
}

}  // namespace

int wmz_convw_supported(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype) {
  return dtype == WMZ_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && (Cout == 128 || (Cout <= 32 && (Cout & 7) == 0)) &&
         (Cin == 64 || Cin == 128) && B > 0 &&
         H > 0 && W > 0 && (H & 7) == 0 && (W & 15) == 0 && (long)B * H * W * 128 < (1L << 31);
}

long wmz_convw_workspace_floats(int B, int H, int W, int Cin, int Cout) {
  const long NK = (long)Cout * 9 * Cin;
  return (long)convw_nsplit(B, H, W, Cin) * ((((NK + 255) >> 8) << 8) + Cout);
}

int wmz_convw_launch(const void* x, const void* dy, float* dW, float* dbias, int B, int H, int W, int Cin, int Cout, int overwrite,
                     int conv_layout_co, int conv_layout_ci, float* workspace, long workspace_floats, hipStream_t stream) {
  WMZ_REQUIRE(workspace_floats >= wmz_convw_workspace_floats(B, H, W, Cin, Cout), "wmz_conv2d_nhwc_wgrad_ws: workspace too small for the direct kernel");
  if (conv_layout_co > 0)
    WMZ_REQUIRE(conv_layout_co <= Cout && conv_layout_ci > 0 && conv_layout_ci <= Cin, "wmz_conv2d_nhwc_wgrad_ws: bad nn.Conv2d layout sizes");
  CwParams P;
  P.x = (const bf16_t*)x; P.dy = (const bf16_t*)dy; P.ws = workspace;
  P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout; P.ncig = Cin / 64;
  P.tiles_x = W / 16; P.tiles_y = H / 8; P.ntiles = B * P.tiles_x * P.tiles_y;
  P.nsplit = convw_nsplit(B, H, W, Cin);
  if (Cout == 128) hipLaunchKernelGGL(convw_kernel<4>, dim3((unsigned)(P.nsplit * P.ncig)), dim3(512), 0, stream, P);
  else hipLaunchKernelGGL(convw_kernel<1>, dim3((unsigned)(P.nsplit * P.ncig)), dim3(128), 0, stream, P);
  WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_wgrad_ws");
  const long NK = (long)Cout * 9 * Cin;
  wmz_wgrad_reduce_launch(workspace, dW, dbias, NK, P.nsplit, Cout, overwrite, conv_layout_co > 0 ? 9 : 0, Cin, conv_layout_co, conv_layout_ci, stream);
  WMZ_LAUNCH_CHECK("wmz_conv2d_nhwc_wgrad_ws");
  return WMZ_OK;
}
