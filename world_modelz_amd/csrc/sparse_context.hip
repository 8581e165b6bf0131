// The host-side prologue of config 5's training step as ONE launch (minecraft/sparse_diffusion.py:train, :398-449):
//   sample_time_dependent (:44-72)   per clip, a frame window whose width grows with the noise level t, placed uniformly, and
//                                    context_length distinct positions uniform inside it (`randperm(window)[:n] + offset`);
//   gather (:437)                    the clip's tokens at those positions (the targets);
//   perturbation & masking (:440-449) the corruption law of wmz_corrupt_tokens on the gathered tokens.
// On the op-by-op route this is ~45 device ops on 6-element tensors (window arithmetic), one top-k over [B, S H W] keys, a sort,
// a gather and the corruption launch: in the captured step every one of them is a graph node the host enqueues in ~10 us -- 0.45 ms
// of a 3.0 ms step in which the GPU mostly waits for the next 3-us kernel.
//
// One workgroup of 1 024 threads per clip:
//   * the window from r[b] and a uniform o (given, or Philox) exactly as the reference computes it (fp32 floor / clamp);
//   * a 32-bit Philox key per position of the window; the n smallest keys are n distinct positions, uniform without replacement,
//     and their key order is a uniform random order (ties -- probability 2^-32 a pair -- fall to the lower position:
//     deterministic).  Selection instead of a full sort (a bitonic sort of 16 384 packed keys in LDS measured 145 us): a histogram
//     of the keys' top 11 bits (LDS atomics) and a scan find the bin that holds the n-th smallest key; every key up to that bin
//     (n plus a bin's worth: <= 1 024) is a candidate, regenerated and compacted into LDS, and only the candidates are sorted
//     (bitonic, <= 1 024 packed (key, position) pairs);
//   * indices, gathered tokens (targets) and their corrupted copies are written from the sorted prefix.
// Deterministic in (seed, stream id, device counter): a hipGraph replay with the counter advanced draws a fresh context.
#include "wmz_common.h"
#include "wmz_internal.h"

namespace {

__device__ __forceinline__ void sc_philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// Philox4x32-10 (Salmon et al. 2011), as loss.hip: counter = (index, stream), key = seed
__device__ __forceinline__ void sc_philox4(unsigned long long idx, unsigned long long seed, unsigned long long stream, unsigned (&c)[4]) {
  c[0] = (unsigned)idx; c[1] = (unsigned)(idx >> 32); c[2] = (unsigned)stream; c[3] = (unsigned)(stream >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    sc_philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float sc_unit(unsigned v) { return (float)(v >> 8) * (1.0f / 16777216.0f); }   // [0, 1)

constexpr int SC_THREADS = 1024;
constexpr unsigned long long SC_KEY_DOMAIN = 1ull << 63;      // stream-id bits that separate the three uses of the generator
constexpr unsigned long long SC_WIN_DOMAIN = 1ull << 62;      // (rank sits in bits 40.., the per-call counter in bits 0..39)

struct ScParams {
  const int64_t* z; long clip_stride;
  const float* r; const float* o;
  int64_t* indices; int64_t* tokens; int64_t* target;
  int B, S, HW, n, C, NP;
  float p_uniform;   // redraw probability per unit of r (training: 0.1, sparse_diffusion.py:447 p_max_uniform; the sampler: 0)
  unsigned long long seed, stream;
  const unsigned long long* counter;
};

constexpr int SC_BINS = 2048, SC_CAND = 1024;

__global__ __launch_bounds__(SC_THREADS) void sparse_context_kernel(ScParams P) {
  __shared__ unsigned hist[SC_BINS];
  __shared__ unsigned long long cand[SC_CAND];                 // packed (key << 32 | position inside the window)
  __shared__ unsigned wave_tot[SC_THREADS / 64];
  __shared__ int win[2];
  __shared__ unsigned sel[2];                                  // the bin of the n-th smallest key; candidates taken so far
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long stream = P.stream;
  if (P.counter != nullptr) stream |= *P.counter & ((1ull << 40) - 1);
  if (tid == 0) {
    // the window (sparse_diffusion.py:53-64), in the reference's fp32 arithmetic
    const float t = fminf(fmaxf(P.r[b], 0.f), 1.f);
    const int need = (P.n + P.HW - 1) / P.HW;
    float frames = floorf((float)need + t * (float)(P.S - need + 1));
    frames = fminf(frames, (float)(P.S - need));
    float o;
    if (P.o != nullptr) o = fminf(fmaxf(P.o[b], 0.f), 1.f - 1e-5f);
    else {
      unsigned c[4];
      sc_philox4((unsigned long long)b, P.seed, stream | SC_WIN_DOMAIN, c);
      o = sc_unit(c[0]);
    }
    const float first = floorf(o * ((float)P.S - frames + 1.f));
    win[0] = (int)first;
    win[1] = (int)frames;
    sel[0] = SC_BINS - 1;                                      // (overwritten by the scan: the window holds >= n positions)
    sel[1] = 0;
  }
  for (int i = tid; i < SC_BINS; i += SC_THREADS) hist[i] = 0;
  cand[tid] = ~0ull;                                           // (SC_CAND == SC_THREADS: the sort's padding)
  __syncthreads();
  const int first = win[0], Wn = win[1] * P.HW;
  const int nq = (Wn + 3) / 4;                                 // Philox calls: four positions each
  const unsigned long long kbase = (unsigned long long)b * (unsigned long long)(P.NP / 4);
  // ---- pass 1: histogram of the keys' top 11 bits
  for (int q = tid; q < nq; q += SC_THREADS) {
    unsigned c[4];
    sc_philox4(kbase + q, P.seed, stream | SC_KEY_DOMAIN, c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < Wn) atomicAdd(&hist[c[e] >> 21], 1u);
  }
  __syncthreads();
  // ---- the bin of the n-th smallest key: exclusive prefix sums over the bins (two bins a thread)
  {
    const unsigned h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    unsigned incl = h0 + h1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned before = incl - (h0 + h1);
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const unsigned want = (unsigned)P.n - 1;                   // rank of the n-th smallest key
    if (before <= want && want < before + h0) sel[0] = 2 * tid;
    else if (before + h0 <= want && want < before + h0 + h1) sel[0] = 2 * tid + 1;
  }
  __syncthreads();
  const unsigned bstar = sel[0];
  // ---- pass 2: the candidates (every key up to that bin: >= n of them, <= n + one bin's worth), regenerated and compacted
  for (int q = tid; q < nq; q += SC_THREADS) {
    unsigned c[4];
    sc_philox4(kbase + q, P.seed, stream | SC_KEY_DOMAIN, c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * q + e;
      if (i < Wn && (c[e] >> 21) <= bstar) {
        const unsigned slot = atomicAdd(&sel[1], 1u);
        if (slot < SC_CAND) cand[slot] = ((unsigned long long)c[e] << 32) | (unsigned)i;
      }
    }
  }
  __syncthreads();
  // ---- bitonic sort of the candidates, ascending (the order of the LDS atomics above does not matter: the sort decides)
  for (int k = 2; k <= SC_CAND; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (tid < SC_CAND / 2) {
        const int lo = 2 * tid - (tid & (j - 1));              // the element of the pair with bit j clear
        const int hi = lo | j;
        const unsigned long long a = cand[lo], c = cand[hi];
        const bool up = (lo & k) == 0;
        if ((a > c) == up) { cand[lo] = c; cand[hi] = a; }
      }
      __syncthreads();
    }
  }
  // ---- the n smallest: positions, targets, corrupted tokens (the law and the stream layout of corrupt_kernel, loss.hip)
  const float rb = P.r[b];
  for (int i = tid; i < P.n; i += SC_THREADS) {
    unsigned wpos = (unsigned)cand[i];
    if (wpos >= (unsigned)Wn) wpos = 0;                        // (never: a sort padding entry would mean fewer than n candidates)
    const long pos = (long)wpos + (long)first * P.HW;
    const int64_t tok = P.z[b * P.clip_stride + pos];
    unsigned c[4];
    sc_philox4((unsigned long long)b * P.n + i, P.seed, stream, c);
    int64_t d = tok;
    if (sc_unit(c[0]) < rb * P.p_uniform) { const int kk = (int)(sc_unit(c[1]) * (float)P.C); d = kk < P.C ? kk : P.C - 1; }
    if (sc_unit(c[2]) < rb) d = P.C;
    const long at = (long)b * P.n + i;
    P.indices[at] = pos;
    P.target[at] = tok;
    P.tokens[at] = d;
  }
}

// The other end of config 5's sampler step (sparse_diffusion.py:190-198): p = softmax(logits), one multinomial draw per row,
// scattered back into the clip at the row's position.  One wave per row of C fp32 logits (C = 8192 at config 5: too wide for
// registers, so three passes over the row, which sits in L2): the maximum, the sum of exp(l - max), then the inverse CDF in class
// order -- the first class whose cumulative weight exceeds u * total (lane l owns the contiguous classes [l C / 64, (l + 1) C / 64):
// lane sums -> a wave prefix -> the owning lane walks its classes).
__global__ __launch_bounds__(256) void categorical_scatter_kernel(const float* __restrict__ logits, long ld, long R, int C,
                                                                  const int64_t* __restrict__ indices, int64_t* __restrict__ z,
                                                                  long clip_stride, long rows_per_clip,
                                                                  int64_t* __restrict__ samples, unsigned long long seed,
                                                                  unsigned long long stream,
                                                                  const unsigned long long* __restrict__ counter) {
  if (counter != nullptr) stream |= *counter & ((1ull << 40) - 1);
  const int lane = threadIdx.x & 63;
  const int per = (C + 63) / 64, c0 = lane * per, c1 = min(c0 + per, C);
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += (long)gridDim.x * 4) {
    const float* x = logits + row * ld;
    float m = -INFINITY;
    for (int c = c0; c < c1; ++c) m = fmaxf(m, x[c]);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
    float s = 0.f;
    for (int c = c0; c < c1; ++c) s += __expf(x[c] - m);
    float incl = s;                                             // inclusive prefix of the lane sums
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    const float total = __shfl(incl, 63);
    unsigned cc[4];
    sc_philox4((unsigned long long)row, seed, stream, cc);
    const float want = sc_unit(cc[0]) * total;
    // the owning lane: the first whose inclusive prefix exceeds `want` (the last lane if rounding leaves none)
    const unsigned long long owners = __ballot(incl > want);
    const int owner = owners ? (int)__builtin_ctzll(owners) : 63;
    int pick = C - 1;
    if (lane == owner) {
      float acc = incl - s;
      pick = c1 - 1 < 0 ? 0 : c1 - 1;
      for (int c = c0; c < c1; ++c) {
        acc += __expf(x[c] - m);
        if (acc > want) { pick = c; break; }
      }
    }
    pick = __shfl(pick, owner);
    if (lane == 0) {
      if (samples != nullptr) samples[row] = pick;
      if (z != nullptr) z[(row / rows_per_clip) * clip_stride + indices[row]] = pick;
    }
  }
}

}  // namespace

// 1 when wmz_sparse_draw_context holds the shape: n <= 512 context positions of a grid of <= 65 536 (the candidates -- n plus one
// histogram bin's worth of keys, S HW / 2048 expected -- must fit the 1 024-entry sort), and the reference's own precondition
extern "C" int wmz_sparse_draw_context_supported(int S, int HW, int n) {
  if (S <= 0 || HW <= 0 || n <= 0) return 0;
  const long G = (long)S * HW;
  const int need = (n + HW - 1) / HW;
  // (2 need <= S: the narrowest window, clamped to S - need frames, still holds n positions -- below that the reference's
  //  randperm(window)[:n] comes back short and its assignment raises)
  return G <= 65536 && n <= 512 && need < S && 2 * need <= S;
}

extern "C" int wmz_sparse_draw_context(const int64_t* z, long clip_stride, const float* r, const float* o, int64_t* indices,
                                       int64_t* tokens, int64_t* target, int B, int S, int HW, int n, int C, float p_uniform,
                                       unsigned long long seed, unsigned long long stream_id, const unsigned long long* counter,
                                       void* stream) {
  WMZ_REQUIRE(z && r && indices && tokens && target && B > 0 && C > 0, "wmz_sparse_draw_context: bad arguments");
  WMZ_REQUIRE(p_uniform >= 0.f && p_uniform <= 1.f, "wmz_sparse_draw_context: p_uniform in [0, 1] expected");
  if (!wmz_sparse_draw_context_supported(S, HW, n)) {
    wmz_set_error("wmz_sparse_draw_context: grid %d x %d with %d context positions not built (<= 65536 positions, <= 512 of "
                  "them drawn, 2 ceil(n / HW) <= S)", S, HW, n);
    return WMZ_ERR_UNSUPPORTED;
  }
  ScParams P;
  P.z = z; P.clip_stride = clip_stride; P.r = r; P.o = o; P.indices = indices; P.tokens = tokens; P.target = target;
  P.B = B; P.S = S; P.HW = HW; P.n = n; P.C = C; P.p_uniform = p_uniform;
  int np = 4;
  while (np < S * HW) np <<= 1;
  P.NP = np;
  P.seed = seed;
  P.stream = counter != nullptr ? (stream_id & ~((1ull << 40) - 1)) : stream_id;
  P.stream &= ~(SC_KEY_DOMAIN | SC_WIN_DOMAIN);
  P.counter = counter;
  hipLaunchKernelGGL(sparse_context_kernel, dim3(B), dim3(SC_THREADS), 0, (hipStream_t)stream, P);
  WMZ_LAUNCH_CHECK("wmz_sparse_draw_context");
  return WMZ_OK;
}

// One multinomial draw per row of fp32 logits [R, C] (row stride ld) from softmax(logits) -- sparse_diffusion.py:190-194 -- and,
// with z != NULL, its scatter into the clips (:197): z[row / rows_per_clip][indices[row]] = the drawn class.  samples (optional):
// the draws [R].  Philox keyed as wmz_sparse_draw_context.
extern "C" int wmz_categorical_scatter(const float* logits, long ld, long R, int C, const int64_t* indices, int64_t* z,
                                       long clip_stride, long rows_per_clip, int64_t* samples, unsigned long long seed,
                                       unsigned long long stream_id, const unsigned long long* counter, void* stream) {
  WMZ_REQUIRE(logits && R > 0 && C > 0 && ld >= C, "wmz_categorical_scatter: bad arguments");
  WMZ_REQUIRE(z == nullptr || (indices != nullptr && rows_per_clip > 0), "wmz_categorical_scatter: the scatter needs indices and rows_per_clip");
  WMZ_REQUIRE(z != nullptr || samples != nullptr, "wmz_categorical_scatter: nothing to write");
  const int grid = (int)((R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048);
  const unsigned long long sid = (counter != nullptr ? (stream_id & ~((1ull << 40) - 1)) : stream_id) & ~(SC_KEY_DOMAIN | SC_WIN_DOMAIN);
  hipLaunchKernelGGL(categorical_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ld, R, C, indices, z, clip_stride,
                     rows_per_clip, samples, seed, sid | (SC_KEY_DOMAIN | SC_WIN_DOMAIN), counter);
  WMZ_LAUNCH_CHECK("wmz_categorical_scatter");
  return WMZ_OK;
}
