// Backward of the token + 3-axis position embedding (local_3d_attention.py:140-157) WITHOUT one float atomic per element:
// the tokens are counting-sorted by class and the embedding table's gradient is a gather.
//
//   dE[c]    = sum over the tokens t with z[t] = c of dx[t]
//   dPs[s], dPh[h], dPw[w] = sums of dx over the tokens of plane s / plane row h / plane column w
//
// The scatter form (capi_core.hip: embed_pos3d_bwd16_kernel) issues N*D float atomics -- 16.8 M at config 4, executed at the
// memory side at ~1.3 TB/s of added bytes: 77 us for a 33 MB read.  Here:
//   A  embed_bwd_pos_hist_kernel   one workgroup per plane: the three position sums from registers (8-byte loads, one row
//                                  of 256 features per wave instruction), and the class histogram (one int atomic per token)
//   B  class_fill_kernel       every workgroup scans the histogram (C <= 12 288 counters: cheaper than a launch) and
//                                  drops its 256 tokens into their class's segment of the sorted list
//   C  embed_bwd_gather_kernel     one wave per 64 consecutive entries of the sorted list: the 64 rows are requested at once
//                                  (row addresses are wave-uniform: v_readlane), summed in list order, and a partial row is
//                                  flushed whenever the class changes -- ~2 atomic row adds per 64 tokens instead of 64, and
//                                  every wave has the same amount of work whatever the class distribution (a frame that is
//                                  half mask tokens, real VQ codes with a few dominant classes); re-zeroes the counters.
// Built for the denoiser's shapes: W = 16, D = 256, bf16 gradients; anything else stays on the scatter kernels.
#include "class_sort.h"

namespace {

constexpr int EB_D = 256;
constexpr int EB_MAXC = CS_MAXC;

__device__ __forceinline__ f32x4 bf4_to_f32(const i32x2& p) {
  f32x4 v;
  v[0] = __builtin_bit_cast(float, p[0] << 16);
  v[1] = __builtin_bit_cast(float, p[0] & 0xffff0000);
  v[2] = __builtin_bit_cast(float, p[1] << 16);
  v[3] = __builtin_bit_cast(float, p[1] & 0xffff0000);
  return v;
}

// A: per-plane partial sums of the three position tables -- written to the workspace, NOT added to the tables: all B*S planes
// add into the same H + 16 rows, and 256-deep chains of same-address float atomics cost ~70 us however little data they
// carry (measured: this kernel with the atomics, and nothing else, ran 72-74 us).  part[plane][0] = plane sum, [1 + h] = plane
// row h, [1 + H + w] = plane column w; the reduce blocks of kernel C fold them.  Also the class histogram: counted in LDS
// first, so that a class which fills half a plane (the mask token) costs one global atomic per plane, not one per token.
__global__ __launch_bounds__(256) void embed_bwd_pos_hist_kernel(const int64_t* __restrict__ z, const bf16_t* __restrict__ dx,
                                                                 float* __restrict__ part, int* __restrict__ cnt, int H,
                                                                 int num_classes) {
  __shared__ f32x4 red[3][17][64];                                        // column + plane sums of waves 1..3 (52 KB)
  extern __shared__ int lh[];                                             // [num_classes] this plane's histogram
  const int plane = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < num_classes; c += 256) lh[c] = 0;
  __syncthreads();
  float* mine = part + (long)plane * (17 + H) * EB_D;
  f32x4 aw[16];
#pragma unroll
  for (int w = 0; w < 16; ++w) aw[w] = (f32x4)(0.f);
  f32x4 as = (f32x4)(0.f);
  i32x2 v[2][16];                                                         // two plane rows in flight per wave
  auto request = [&](int buf, int h) {
    const long row = ((long)plane * H + (h < H ? h : H - 1)) * 16;
#pragma unroll
    for (int w = 0; w < 16; ++w) v[buf][w] = *reinterpret_cast<const i32x2*>(dx + (row + w) * EB_D + lane * 4);
  };
  auto consume = [&](int buf, int h) {
    f32x4 ah = (f32x4)(0.f);
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const f32x4 f = bf4_to_f32(v[buf][w]);
      aw[w] += f;
      ah += f;
    }
    as += ah;
    *reinterpret_cast<f32x4*>(mine + (1 + h) * EB_D + lane * 4) = ah;
  };
  request(0, wave);
  for (int h = wave; h < H; h += 8) {                                     // a wave = a plane row at a time, rows wave, wave + 4, ..
    request(1, h + 4);
    if (lane < 16) atomicAdd(lh + clamp_class(z[((long)plane * H + h) * 16 + lane], num_classes), 1);
    consume(0, h);
    if (h + 4 < H) {
      request(0, h + 8);
      if (lane < 16) atomicAdd(lh + clamp_class(z[((long)plane * H + h + 4) * 16 + lane], num_classes), 1);
      consume(1, h + 4);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int w = 0; w < 16; ++w) red[wave - 1][w][lane] = aw[w];
    red[wave - 1][16][lane] = as;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 16; ++w)
      *reinterpret_cast<f32x4*>(mine + (1 + H + w) * EB_D + lane * 4) = (aw[w] + red[0][w][lane]) + (red[1][w][lane] + red[2][w][lane]);
    *reinterpret_cast<f32x4*>(mine + lane * 4) = (as + red[0][16][lane]) + (red[1][16][lane] + red[2][16][lane]);
  }
  for (int c = threadIdx.x; c < num_classes; c += 256) {
    const int n = lh[c];
    if (n > 0) atomicAdd(cnt + c, n);
  }
}

// C: blocks [0, gather_blocks): one wave per 64 list entries (see the header); blocks behind them: the position tables from
// the per-plane partial sums -- block (y, chunk of 32 planes), thread = feature: 32 loads in flight, one atomic per feature
// (chains of planes / 32; y = 0, the plane sums, goes to dps[plane % S] plane by plane: chains of B).
__global__ __launch_bounds__(256) void embed_bwd_gather_kernel(const i32x2* __restrict__ sorted, const bf16_t* __restrict__ dx,
                                                               float* __restrict__ demb, int* __restrict__ cnt,
                                                               int* __restrict__ fill, long ntok, int num_classes,
                                                               int gather_blocks, const float* __restrict__ part,
                                                               float* __restrict__ dps, float* __restrict__ dph,
                                                               float* __restrict__ dpw, int planes, int S, int H) {
  if ((int)blockIdx.x >= gather_blocks) {
    const int r = (int)blockIdx.x - gather_blocks;
    const int y = r % (17 + H), p0 = (r / (17 + H)) * 32, t = threadIdx.x;
    const long stride = (long)(17 + H) * EB_D;
    const float* src = part + (long)y * EB_D + t;
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = src[(long)min(p0 + i, planes - 1) * stride];
    if (y == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if (p0 + i < planes) atomicAdd(dps + (long)((p0 + i) % S) * EB_D + t, x[i]);
      return;
    }
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) a += p0 + i < planes ? x[i] : 0.f;
    float* dst = y <= H ? dph + (long)(y - 1) * EB_D : dpw + (long)(y - 1 - H) * EB_D;
    atomicAdd(dst + t, a);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0)                                                    // counters back to zero for the next call
    for (int c = threadIdx.x; c < num_classes; c += 256) { cnt[c] = 0; fill[c] = 0; }
  const long e0 = ((long)blockIdx.x * 4 + wave) * 64;
  if (e0 >= ntok) return;
  int tl = 0, cl = -1;
  if (e0 + lane < ntok) {
    const i32x2 e = sorted[e0 + lane];
    tl = e[0];
    cl = e[1];
  }
  i32x2 v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const long t = __builtin_amdgcn_readlane(tl, i);                      // (entries past the end read row 0 and are skipped below)
    v[i] = *reinterpret_cast<const i32x2*>(dx + t * EB_D + lane * 4);
  }
  f32x4 acc = (f32x4)(0.f);
  int cur = __builtin_amdgcn_readlane(cl, 0);
  auto flush = [&](int c) {
    float* p = demb + (long)c * EB_D + lane * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(p + k, acc[k]);
  };
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = __builtin_amdgcn_readlane(cl, i);
    if (c != cur) {
      if (cur >= 0) flush(cur);
      acc = (f32x4)(0.f);
      cur = c;
    }
    if (c >= 0) acc += bf4_to_f32(v[i]);
  }
  if (cur >= 0) flush(cur);
}

}  // namespace

// Layout: counters at FIXED offsets (cnt[EB_MAXC] | fill[EB_MAXC]), whatever num_classes is -- one workspace serves calls with
// different class counts, and the region that must be zero between calls never overlaps another call's scratch -- then the list.
extern "C" long wmz_embed_pos3d_bwd_workspace_ints(int B, int S, int H, int W, int num_classes) {
  (void)num_classes;
  (void)W;
  return 2L * EB_MAXC + 2 * (long)B * S * H * 16 + (long)B * S * (17 + H) * EB_D;
}

extern "C" int wmz_embed_pos3d_bwd_sorted(const int64_t* z, const void* dx, float* demb, float* dpos_s, float* dpos_h,
                                          float* dpos_w, int B, int S, int H, int W, int D, int num_classes, int* workspace,
                                          long workspace_ints, int dtype, void* stream) {
  WMZ_REQUIRE(z && dx && demb && dpos_s && dpos_h && dpos_w && workspace, "wmz_embed_pos3d_bwd_sorted: null tensor");
  WMZ_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D > 0 && num_classes > 0, "wmz_embed_pos3d_bwd_sorted: bad shape");
  if (!(W == 16 && D == EB_D && dtype == WMZ_BF16 && num_classes <= EB_MAXC)) {
    wmz_set_error("wmz_embed_pos3d_bwd_sorted: built for W = 16, D = 256, bf16, <= 12288 classes (got W %d, D %d, dtype %d, %d classes); "
                  "use wmz_embed_pos3d_bwd", W, D, dtype, num_classes);
    return WMZ_ERR_UNSUPPORTED;
  }
  const long ntok = (long)B * S * H * W;
  WMZ_REQUIRE(ntok < (1L << 31), "wmz_embed_pos3d_bwd_sorted: too many tokens");
  WMZ_REQUIRE(workspace_ints >= wmz_embed_pos3d_bwd_workspace_ints(B, S, H, W, num_classes),
              "wmz_embed_pos3d_bwd_sorted: workspace too small (%ld ints needed)",
              wmz_embed_pos3d_bwd_workspace_ints(B, S, H, W, num_classes));
  int* cnt = workspace;
  int* fill = workspace + EB_MAXC;
  i32x2* sorted = reinterpret_cast<i32x2*>(workspace + 2 * EB_MAXC);
  float* part = reinterpret_cast<float*>(workspace + 2 * EB_MAXC + 2 * ntok);
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * S, gblocks = wmz_cdiv(ntok, 256L), rblocks = (17 + H) * wmz_cdiv(planes, 32);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(embed_bwd_pos_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(class_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL(embed_bwd_pos_hist_kernel, dim3((unsigned)planes), dim3(256), (size_t)num_classes * 4, st, z,
                     (const bf16_t*)dx, part, cnt, H, num_classes);
  hipLaunchKernelGGL(class_fill_kernel, dim3((unsigned)gblocks), dim3(256), (size_t)num_classes * 8, st, z, cnt, fill, sorted,
                     ntok, num_classes);
  hipLaunchKernelGGL(embed_bwd_gather_kernel, dim3((unsigned)(gblocks + rblocks)), dim3(256), 0, st, sorted, (const bf16_t*)dx,
                     demb, cnt, fill, ntok, num_classes, gblocks, part, dpos_s, dpos_h, dpos_w, planes, S, H);
  WMZ_LAUNCH_CHECK("wmz_embed_pos3d_bwd_sorted");
  return WMZ_OK;
}
