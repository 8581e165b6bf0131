// Training-mode BatchNorm bookkeeping (autoencoder.py:21-25: nn.BatchNorm2d behind every convolution) done by the CONSUMER of the
// normalised tensor: the kernel that applies the normalisation (wmz_affine_act_nhwc_bn, the input prologue of
// wmz_conv_point_fwd_bn) turns the producer's raw statistics into (scale, shift) itself while it fills its per-channel table --
// 16 L2 reads per channel and workgroup -- and ONE of its workgroups moves the running statistics and publishes scale / shift / mean
// / rstd for the backward pass.  That removes the wmz_bn_finalize launch between every convolution and its consumer (ten per
// frame-encoder call: ~5 us each of pure launch latency on the critical path).  The arithmetic is wmz_bn_finalize's (one function).
// (The other way round -- the producing convolution's last workgroup finalises -- was built and measured first: every workgroup has
// to wait for its statistics atomics to return before it may draw its ticket, ~3 us at the end of each of thousands of workgroups:
// slower than the launch it removed.  profiles/r05/bn_tail_experiment.patch.)
#pragma once
#include "wmz_common.h"

namespace {

struct BnStats {                            // kernel-argument copy of wmz_bn_stats (sum == nullptr: not used)
  const float* sum; const float* sq; const float* gamma; const float* beta;
  float* running_mean; float* running_var; long long* nbt;
  float* scale; float* shift; float* mean; float* rstd;
  float count, momentum, eps;
};

inline BnStats bn_stats_from(const wmz_bn_stats* t) {
  BnStats b{};
  if (t == nullptr) return b;
  b.sum = t->sum; b.sq = t->sq; b.gamma = t->gamma; b.beta = t->beta;
  b.running_mean = t->running_mean; b.running_var = t->running_var; b.nbt = (long long*)t->num_batches_tracked;
  b.scale = t->scale; b.shift = t->shift; b.mean = t->mean; b.rstd = t->rstd;
  b.count = (float)t->count; b.momentum = (float)t->momentum; b.eps = (float)t->eps;
  return b;
}

inline bool bn_stats_ok(const wmz_bn_stats* t) {
  return t == nullptr || (t->sum && t->sq && t->count > 0 && (t->running_mean == nullptr) == (t->running_var == nullptr) &&
                          (t->mean == nullptr) == (t->rstd == nullptr) && (t->scale == nullptr) == (t->shift == nullptr));
}

// channel c: (scale, shift) from the batch statistics [WMZ_STAT_REPLICAS][C]; publish: also move the running statistics (momentum,
// unbiased variance), count the step and write the optional outputs -- exactly once per launch and channel
__device__ __forceinline__ void bn_channel(const BnStats& b, int C, int c, bool publish, float& sc, float& sh) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int r = 0; r < WMZ_STAT_REPLICAS; ++r) { s1 += b.sum[r * C + c]; s2 += b.sq[r * C + c]; }
  const float mean = s1 / b.count;
  const float var = fmaxf(s2 / b.count - mean * mean, 0.f);
  const float rs = rsqrtf(var + b.eps);
  const float g = b.gamma ? b.gamma[c] : 1.f, be = b.beta ? b.beta[c] : 0.f;
  sc = g * rs;
  sh = be - mean * g * rs;
  if (publish) {
    if (b.running_mean) {
      const float unbiased = b.count > 1.f ? var * b.count / (b.count - 1.f) : var;
      b.running_mean[c] = (1.f - b.momentum) * b.running_mean[c] + b.momentum * mean;
      b.running_var[c] = (1.f - b.momentum) * b.running_var[c] + b.momentum * unbiased;
    }
    if (c == 0 && b.nbt != nullptr) *b.nbt += 1;              // nn.BatchNorm2d's step counter
    if (b.scale) { b.scale[c] = sc; b.shift[c] = sh; }
    if (b.mean) { b.mean[c] = mean; b.rstd[c] = rs; }
  }
}

}  // namespace
