// VectorQuantizerEMA kernels (vq-video-diffusion/vq.py).  Compiled with -ffp-contract=off: the distance
// arithmetic must round exactly like the reference's separate sub / mul / add tensor ops.
#include "class_sort.h"
#include <limits.h>

namespace {

constexpr int VQ_ROWS = 64;    // rows per workgroup: lane = row
constexpr int VQ_WAVES = 4;    // each wave scans a quarter of every codebook tile
constexpr int VQ_NT = VQ_ROWS * VQ_WAVES;

// sum_e (x[e]-c[e])^2 in the order ATen's CPU `vectorized_inner_sum` uses for the reference expression
// (vq.py:30): 8 SIMD lanes x 4 interleaved accumulators, leftover vectors added to a0, combined ((a0+a1)+a2)+a3, then
// the scalar tail first and the 8 lanes in order.  Pinned by oracle.vq.distances_avx_order / tests.
template <int E, typename XF, typename CF>
__device__ __forceinline__ float dist_exact(XF xf, CF cf) {
  constexpr int NV = E / 8, NI = NV / 4;
  float t[8];
  if constexpr (NI > 0) {
    float acc[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xf(8 * k + j) - cf(8 * k + j); acc[k][j] = d * d; }
#pragma unroll
    for (int i = 1; i < NI; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = 8 * (4 * i + k) + j;
          const float d = xf(e) - cf(e);
          acc[k][j] = acc[k][j] + d * d;
        }
    // leftover vectors join accumulator 0 BEFORE the accumulators are combined (ATen row_sum)
#pragma unroll
    for (int v = NI * 4; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xf(8 * v + j) - cf(8 * v + j); acc[0][j] = acc[0][j] + d * d; }
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xf(8 * v + j) - cf(8 * v + j); t[j] = t[j] + d * d; }
  }
  float fin = 0.f;
#pragma unroll
  for (int e = NV * 8; e < E; ++e) { const float d = xf(e) - cf(e); fin = fin + d * d; }
#pragma unroll
  for (int j = 0; j < 8; ++j) fin = fin + t[j];
  return fin;
}

// runtime-E variant of the same order (x and code rows read through pointers)
__device__ __forceinline__ float dist_exact_rt(const float* x, int xs, const float* c, int E) {
  const int NV = E / 8, NI = NV / 4;
  float t[8];
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = 8 * (4 * i + k) + j;
        const float d = x[e * xs] - c[e];
        acc[k][j] = acc[k][j] + d * d;
      }
  for (int v = NI * 4; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = x[(8 * v + j) * xs] - c[8 * v + j]; acc[0][j] = acc[0][j] + d * d; }
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
  float fin = 0.f;
  for (int e = NV * 8; e < E; ++e) { const float d = x[e * xs] - c[e]; fin = fin + d * d; }
#pragma unroll
  for (int j = 0; j < 8; ++j) fin = fin + t[j];
  return fin;
}

// E > 0: compile-time embedding dim, x row in registers.  E == 0: runtime dim, x tile in LDS ([e][row] so the
// lane = row read is conflict-free).
template <int E>
__global__ __launch_bounds__(VQ_NT) void vq_argmin_kernel(const float* __restrict__ X, long ldx,
                                                          const float* __restrict__ CB, int64_t* __restrict__ IDX,
                                                          float* __restrict__ DMIN, int N, int C, int Ert, int CT) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Ed = E > 0 ? E : Ert;
  float* cbs = sm;                                   // [CT][Ed]
  float* xs = sm + (size_t)CT * Ed;                  // E==0 only: [Ed][VQ_ROWS]
  float* red_d = xs + (E > 0 ? 0 : (size_t)Ed * VQ_ROWS);   // [VQ_WAVES][VQ_ROWS]
  int* red_c = reinterpret_cast<int*>(red_d + VQ_WAVES * VQ_ROWS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = (long)blockIdx.x * VQ_ROWS + lane;
  const bool rok = row < N;

  float xr[E > 0 ? E : 1];
  if constexpr (E > 0) {
    // unconditional, vectorised loads from a clamped row (a predicated load per element makes hipcc branch and wait per
    // element: 64 L2 round trips in a row); rows past N are computed on the last row and never stored
    const float* xrow = X + (rok ? row : (long)N - 1) * ldx;
    if ((ldx & 3) == 0 && (reinterpret_cast<size_t>(X) & 15) == 0) {
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + e);
        xr[e] = v[0]; xr[e + 1] = v[1]; xr[e + 2] = v[2]; xr[e + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) xr[e] = xrow[e];
    }
  } else {
    for (int i = tid; i < Ed * VQ_ROWS; i += VQ_NT) {
      const int r = i / Ed, e = i - r * Ed;
      const long gr = (long)blockIdx.x * VQ_ROWS + r;
      xs[e * VQ_ROWS + r] = gr < N ? X[gr * ldx + e] : 0.f;
    }
  }

  float best_d = INFINITY;
  int best_c = INT_MAX;
  if constexpr (E > 0) {
    // Compile-time E: the code rows come through the SCALAR data path.  Every lane of a wave needs the same code row at
    // the same time, so its address is wave-uniform: hipcc emits s_load_dwordx16 and the VALU instructions take the code
    // values as SGPR operands -- no LDS tile, no workgroup barrier per tile, no 16 broadcast ds_read_b128 per code and
    // wave (which, at 4 waves per SIMD, kept the CU's LDS as busy as its vector ALUs: 330 -> 282 us at C = 1024).
    // Wave w scans the contiguous quarter w of the codebook; the (d, c) merge below restores the global order.
    const int per = (C + VQ_WAVES - 1) / VQ_WAVES;
    const int cb0 = __builtin_amdgcn_readfirstlane(wave * per);
    const int cb1 = __builtin_amdgcn_readfirstlane(min(C, wave * per + per));
    for (int c = cb0; c < cb1; ++c) {
      const float* crow = CB + (long)c * E;
      const float d = dist_exact<E>([&](int e) { return xr[e]; }, [&](int e) { return crow[e]; });
      if (d < best_d) { best_d = d; best_c = c; }
    }
  }
  const int per_wave = CT / VQ_WAVES;
  for (int c0 = 0; E == 0 && c0 < C; c0 += CT) {
    __syncthreads();
    const int nc = min(CT, C - c0);
    for (int i = tid * 4; i < nc * Ed; i += VQ_NT * 4) {
      if (((Ed & 3) == 0)) *reinterpret_cast<f32x4*>(cbs + i) = *reinterpret_cast<const f32x4*>(CB + (long)c0 * Ed + i);
      else for (int k = 0; k < 4 && i + k < nc * Ed; ++k) cbs[i + k] = CB[(long)c0 * Ed + i + k];
    }
    __syncthreads();
    const int cw0 = wave * per_wave, cw1 = min(nc, cw0 + per_wave);
    for (int c = cw0; c < cw1; ++c) {
      const float* crow = cbs + c * Ed;
      float d;
      if constexpr (E > 0) d = dist_exact<E>([&](int e) { return xr[e]; }, [&](int e) { return crow[e]; });
      else d = dist_exact_rt(xs + lane, VQ_ROWS, crow, Ed);
      if (d < best_d) { best_d = d; best_c = c0 + c; }
    }
  }
  red_d[wave * VQ_ROWS + lane] = best_d;
  red_c[wave * VQ_ROWS + lane] = best_c;
  __syncthreads();
  if (wave == 0 && rok) {
    float bd = red_d[lane];
    int bc = red_c[lane];
#pragma unroll
    for (int w = 1; w < VQ_WAVES; ++w) {
      const float d = red_d[w * VQ_ROWS + lane];
      const int c = red_c[w * VQ_ROWS + lane];
      if (d < bd || (d == bd && c < bc)) { bd = d; bc = c; }
    }
    if (bc == INT_MAX) bc = 0;
    IDX[row] = bc;
    if (DMIN) DMIN[row] = bd;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vq_gather_kernel(const int64_t* __restrict__ idx, const float* __restrict__ cb,
                                                        T* __restrict__ out, long ldo, long N, int C, int E) {
  const long total = N * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / E;
    const int e = (int)(i - n * E);
    long c = idx[n];
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    out[n * ldo + e] = Elem<T>::from_f32(cb[c * E + e]);
  }
}

// one lane per (row, channel): a wave's atomics land on <= 64 contiguous floats of one dw row
__global__ __launch_bounds__(256) void vq_stats_kernel(const float* __restrict__ X, long ldx,
                                                       const int64_t* __restrict__ idx, const float* __restrict__ cb,
                                                       float* __restrict__ counts, float* __restrict__ dw,
                                                       float* __restrict__ sqerr, long N, int C, int E) {
  const long total = N * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / E;
    const int e = (int)(i - n * E);
    long c = idx[n];
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    const float x = X[n * ldx + e];
    if (dw) atomicAdd(dw + c * E + e, x);
    if (sqerr) { const float d = cb[c * E + e] - x; atomicAdd(sqerr + c, d * d); }
    if (counts && e == 0) atomicAdd(counts + c, 1.0f);
  }
}

// The same statistics as a GATHER over the class-sorted row list (class_sort.h): one wave per 64 consecutive list entries
// walks them 16 rows at a time (16 row loads in flight; lane = feature e, e + 64, ..), keeps a running row sum, squared error
// and count of the current code, and flushes them -- E atomics for the dw row, ONE for the count, ONE for the error after a
// wave reduction -- whenever the code changes.  The scatter kernel above issues N*E atomics on sqerr[c] alone, 64 lanes of
// a wave on ONE address: 1.77 ms at N = 65 536, C = 1 024, E = 64 (0.48 ms without sqerr when one code holds 30 % of the
// rows); this one is insensitive to the code distribution.  Also re-zeroes the sort's counters.
template <int EK>     // ceil(E / 64)
__global__ __launch_bounds__(256) void vq_stats_gather_kernel(const i32x2* __restrict__ sorted, const float* __restrict__ X,
                                                              long ldx, const float* __restrict__ cb, float* __restrict__ counts,
                                                              float* __restrict__ dw, float* __restrict__ sqerr,
                                                              int* __restrict__ cnt, int* __restrict__ fill, long N, int C, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += 256) { cnt[c] = 0; fill[c] = 0; }
  const long e0 = ((long)blockIdx.x * 4 + wave) * 64;
  if (e0 >= N) return;
  int tl = 0, cl = -1;
  if (e0 + lane < N) {
    const i32x2 e = sorted[e0 + lane];
    tl = e[0];
    cl = e[1];
  }
  float acc[EK], cbr[EK], err = 0.f;
  int run = 0, cur = -1;
#pragma unroll
  for (int k = 0; k < EK; ++k) { acc[k] = 0.f; cbr[k] = 0.f; }
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int k = 0; k < EK; ++k) {
      const int e = lane + 64 * k;
      if (dw != nullptr && e < E) atomicAdd(dw + (long)cur * E + e, acc[k]);
    }
    if (sqerr != nullptr) {
      float t = err;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (lane == 0) atomicAdd(sqerr + cur, t);
    }
    if (counts != nullptr && lane == 0) atomicAdd(counts + cur, (float)run);
  };
  for (int i0 = 0; i0 < 64; i0 += 16) {
    float v[16][EK];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long t = __shfl(tl, i0 + i);                                  // (entries past the end read row 0 and are skipped)
#pragma unroll
      for (int k = 0; k < EK; ++k) {
        const int e = lane + 64 * k;
        v[i][k] = e < E ? X[t * ldx + e] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = __shfl(cl, i0 + i);
      if (c != cur) {
        flush();
        cur = c;
        run = 0;
        err = 0.f;
#pragma unroll
        for (int k = 0; k < EK; ++k) {
          acc[k] = 0.f;
          const int e = lane + 64 * k;
          cbr[k] = (c >= 0 && e < E) ? cb[(long)c * E + e] : 0.f;
        }
      }
      if (c >= 0) {
        ++run;
#pragma unroll
        for (int k = 0; k < EK; ++k) {
          acc[k] += v[i][k];
          const float d = cbr[k] - v[i][k];
          if (lane + 64 * k < E) err = fmaf(d, d, err);
        }
      }
    }
  }
  flush();
}

// single workgroup: cluster-size EMA, Laplace smoothing, codebook update (vq.py:44, :53-65)
__global__ __launch_bounds__(1024) void vq_ema_update_kernel(float* __restrict__ emb, float* __restrict__ cs,
                                                             float* __restrict__ act, const float* __restrict__ counts,
                                                             const float* __restrict__ dw, int C, int E, float decay,
                                                             float one_minus, float eps) {
  __shared__ float red[1024];
  __shared__ float n_s;
  const int tid = threadIdx.x;
  float part = 0.f;
  for (int c = tid; c < C; c += 1024) {
    const float v = cs[c] * decay + one_minus * counts[c];
    cs[c] = v;
    if (act) act[c] += counts[c];
    part += v;
  }
  red[tid] = part;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) n_s = red[0];
  __syncthreads();
  const float n = n_s;
  const float denom = n + (float)C * eps;
  for (int i = tid; i < C * E; i += 1024) {
    const int c = i / E;
    const float smooth = (cs[c] + eps) / denom * n;
    emb[i] = emb[i] * decay + one_minus * (dw[i] / smooth);
  }
}

template <int E>
int launch_argmin(const float* x, long ldx, const float* cb, int64_t* idx, float* dmin, int N, int C, int Ert,
                  hipStream_t st) {
  const int Ed = E > 0 ? E : Ert;
  // LDS: a tile of CT codes + (run-time E only) the workgroup's 64 rows, transposed + the waves' minima.  Wide rows (E > ~100 on
  // the run-time path) need more than the default 64 KB of dynamic LDS: the codebook tile shrinks first, then the limit is raised
  // (160 KB per workgroup on gfx950; E < 512 -- the caller's bound -- fits with CT >= 4)
  const size_t fixed = (E > 0 ? 0 : (size_t)Ed * VQ_ROWS * 4) + VQ_WAVES * VQ_ROWS * 8;
  int CT = 128;
  while (CT > 4 && (size_t)CT * Ed * 4 > 48 * 1024) CT >>= 1;
  while (CT > 4 && (size_t)CT * Ed * 4 + fixed > 152 * 1024) CT >>= 1;
  const size_t smem = (size_t)CT * Ed * 4 + fixed;
  if (smem > 160 * 1024) { wmz_set_error("wmz_vq_argmin: embedding_dim %d too large", Ed); return WMZ_ERR_UNSUPPORTED; }
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(vq_argmin_kernel<E>, dim3(wmz_cdiv(N, VQ_ROWS)), dim3(VQ_NT), smem, st, x, ldx, cb, idx, dmin, N,
                     C, Ert, CT);
  WMZ_LAUNCH_CHECK("wmz_vq_argmin");
  return WMZ_OK;
}

}  // namespace

extern "C" int wmz_vq_argmin(const float* x, long ldx, const float* codebook, int64_t* idx, float* dist_min, int N,
                             int C, int E, void* stream) {
  WMZ_REQUIRE(x && codebook && idx, "wmz_vq_argmin: null tensor");
  WMZ_REQUIRE(N >= 0 && C > 0 && E > 0, "wmz_vq_argmin: bad shape N=%d C=%d E=%d", N, C, E);
  WMZ_REQUIRE(E < 512, "wmz_vq_argmin: embedding_dim >= 512 needs ATen's cascade levels (not built)");
  if (N == 0) return WMZ_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (E) {
    case 64: return launch_argmin<64>(x, ldx, codebook, idx, dist_min, N, C, E, st);
    case 32: return launch_argmin<32>(x, ldx, codebook, idx, dist_min, N, C, E, st);
    case 16: return launch_argmin<16>(x, ldx, codebook, idx, dist_min, N, C, E, st);
    case 8: return launch_argmin<8>(x, ldx, codebook, idx, dist_min, N, C, E, st);
    default: return launch_argmin<0>(x, ldx, codebook, idx, dist_min, N, C, E, st);
  }
}

extern "C" int wmz_vq_gather(const int64_t* idx, const float* codebook, void* out, long ldo, int N, int C, int E,
                             int dtype, void* stream) {
  WMZ_REQUIRE(idx && codebook && out, "wmz_vq_gather: null tensor");
  WMZ_REQUIRE(N >= 0 && C > 0 && E > 0, "wmz_vq_gather: bad shape");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_vq_gather: bad dtype %d", dtype);
  if (N == 0) return WMZ_OK;
  const long total = (long)N * E;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16)
    hipLaunchKernelGGL(vq_gather_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, idx, codebook, (bf16_t*)out, ldo, (long)N, C, E);
  else
    hipLaunchKernelGGL(vq_gather_kernel<float>, dim3(grid), dim3(256), 0, st, idx, codebook, (float*)out, ldo, (long)N, C, E);
  WMZ_LAUNCH_CHECK("wmz_vq_gather");
  return WMZ_OK;
}

extern "C" int wmz_vq_ema_stats(const float* x, long ldx, const int64_t* idx, const float* codebook, float* counts,
                                float* dw, float* sqerr, int N, int C, int E, void* stream) {
  WMZ_REQUIRE(x && idx && codebook, "wmz_vq_ema_stats: null tensor");
  WMZ_REQUIRE(N >= 0 && C > 0 && E > 0, "wmz_vq_ema_stats: bad shape");
  if (N == 0) return WMZ_OK;
  const long total = (long)N * E;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(vq_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, codebook, counts, dw,
                     sqerr, (long)N, C, E);
  WMZ_LAUNCH_CHECK("wmz_vq_ema_stats");
  return WMZ_OK;
}

extern "C" long wmz_vq_ema_stats_workspace_ints(int N, int C) {
  (void)C;
  return 2L * CS_MAXC + 2L * N;
}

extern "C" int wmz_vq_ema_stats_sorted(const float* x, long ldx, const int64_t* idx, const float* codebook, float* counts,
                                       float* dw, float* sqerr, int N, int C, int E, int* workspace, long workspace_ints,
                                       void* stream) {
  WMZ_REQUIRE(x && idx && codebook && workspace && N > 0 && C > 0 && E > 0, "wmz_vq_ema_stats_sorted: bad arguments");
  if (C > CS_MAXC || E > 256) {
    wmz_set_error("wmz_vq_ema_stats_sorted: built for <= %d codes of <= 256 features (got %d, %d); use wmz_vq_ema_stats", CS_MAXC, C, E);
    return WMZ_ERR_UNSUPPORTED;
  }
  WMZ_REQUIRE(workspace_ints >= wmz_vq_ema_stats_workspace_ints(N, C), "wmz_vq_ema_stats_sorted: workspace too small (%ld ints needed)",
              wmz_vq_ema_stats_workspace_ints(N, C));
  int* cnt = workspace;
  int* fill = workspace + CS_MAXC;
  i32x2* sorted = reinterpret_cast<i32x2*>(workspace + 2 * CS_MAXC);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = (unsigned)wmz_cdiv(N, 256);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(class_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(class_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL(class_hist_kernel, dim3(nb), dim3(256), (size_t)C * 4, st, idx, cnt, (long)N, C);
  hipLaunchKernelGGL(class_fill_kernel, dim3(nb), dim3(256), (size_t)C * 8, st, idx, cnt, fill, sorted, (long)N, C);
#define WMZ_VQG(EK) hipLaunchKernelGGL(vq_stats_gather_kernel<EK>, dim3(nb), dim3(256), 0, st, sorted, x, ldx, codebook, counts, dw, sqerr, cnt, fill, (long)N, C, E)
  if (E <= 64) WMZ_VQG(1); else if (E <= 128) WMZ_VQG(2); else WMZ_VQG(4);
#undef WMZ_VQG
  WMZ_LAUNCH_CHECK("wmz_vq_ema_stats_sorted");
  return WMZ_OK;
}

extern "C" int wmz_vq_ema_update(float* embedding, float* cluster_size, float* activation_count, const float* counts,
                                 const float* dw, int C, int E, double decay, double eps, void* stream) {
  WMZ_REQUIRE(embedding && cluster_size && counts && dw, "wmz_vq_ema_update: null tensor");
  WMZ_REQUIRE(C > 0 && E > 0, "wmz_vq_ema_update: bad shape");
  // the reference multiplies fp32 tensors by the Python doubles `decay` and `1 - decay` (vq.py:53, :65)
  const float one_minus = (float)(1.0 - decay);
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, embedding, cluster_size,
                     activation_count, counts, dw, C, E, (float)decay, one_minus, (float)eps);
  WMZ_LAUNCH_CHECK("wmz_vq_ema_update");
  return WMZ_OK;
}
