// Counting sort of N items by class (shared by the embedding backward, embed_bwd.hip, and the VQ statistics, vq.hip): global
// histogram -> class offsets -> class-sorted list of (item, class) pairs, with integer atomics only and at most one GLOBAL
// atomic per (workgroup, class): a class that holds half of a workgroup's items costs one, not 128 (same-address atomics
// complete ~15 ns apart: a 1 000-deep chain is 15 us whatever it carries).
#pragma once
#include "wmz_common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int CS_MAXC = 12288;            // classes: the LDS images of the histogram / offsets are 48 KB each

__device__ __forceinline__ int clamp_class(long tk, int num_classes) {
  return (int)(tk < 0 ? 0 : (tk >= num_classes ? num_classes - 1 : tk));
}

// cnt[c] += #{t in this workgroup's 256 items: key[t] == c}   (dynamic LDS: num_classes ints)
__global__ __launch_bounds__(256) void class_hist_kernel(const int64_t* __restrict__ key, int* __restrict__ cnt, long n,
                                                         int num_classes) {
  extern __shared__ int cs_lh[];
  int* lh = cs_lh;
  for (int c = threadIdx.x; c < num_classes; c += 256) lh[c] = 0;
  __syncthreads();
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) atomicAdd(lh + clamp_class(key[t], num_classes), 1);
  __syncthreads();
  for (int c = threadIdx.x; c < num_classes; c += 256) {
    const int m = lh[c];
    if (m > 0) atomicAdd(cnt + c, m);
  }
}

// exclusive scan of one value per thread over the 256 threads of a workgroup
__device__ __forceinline__ int block_exclusive_scan(int v, int* wsum /* [4] in LDS */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) if (w < wave) base += wsum[w];
  return base + inc - v;
}

// B: every workgroup scans the global histogram into class offsets (LDS), ranks its 256 tokens within their classes (LDS
// atomics), reserves room for each class it holds by ONE returning global atomic (issued by the token ranked first), and drops
// the tokens into the list.
__global__ __launch_bounds__(256) void class_fill_kernel(const int64_t* __restrict__ z, const int* __restrict__ cnt,
                                                             int* __restrict__ fill, i32x2* __restrict__ sorted, long ntok,
                                                             int num_classes) {
  extern __shared__ int cs_sm[];
  int* lh = cs_sm;                                                           // [C] local count, then the reserved base
  int* lo = cs_sm + num_classes;                                             // [C] class offset in the list
  __shared__ int wsum[4];
  const int tid = threadIdx.x;
  for (int c = tid; c < num_classes; c += 256) lh[c] = 0;
  __syncthreads();
  const long t = (long)blockIdx.x * 256 + tid;
  const bool ok = t < ntok;
  const int ct = ok ? clamp_class(z[t], num_classes) : 0;
  const int rank = ok ? atomicAdd(lh + ct, 1) : -1;
  const int seg = (num_classes + 255) / 256;
  const int c0 = tid * seg, c1 = min(num_classes, c0 + seg);
  int sum = 0;
  for (int c = c0; c < c1; ++c) sum += cnt[c];
  int run = block_exclusive_scan(sum, wsum);                              // (its barrier also closes the ranking pass)
  for (int c = c0; c < c1; ++c) { lo[c] = run; run += cnt[c]; }
  int base = 0;
  if (rank == 0) base = atomicAdd(fill + ct, lh[ct]);
  __syncthreads();
  if (rank == 0) lh[ct] = base;
  __syncthreads();
  if (ok) {
    i32x2 e;
    e[0] = (int)t;
    e[1] = ct;
    sorted[lo[ct] + lh[ct] + rank] = e;
  }
}

}  // namespace
