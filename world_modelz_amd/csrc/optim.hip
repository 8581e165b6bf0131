// Flat-arena optimizer kernels for the denoiser training step (reference: grad_norm main.py:188-193, which costs
// one .item() sync per parameter tensor, and torch.optim.AdamW as configured at main.py:433).  All parameters,
// gradients and moments live in contiguous fp32 arenas, so each of these is ONE launch over n elements.
#include "wmz_common.h"

namespace {

// Grids are capped at OPT_WGS workgroups of 512 threads walking 16-byte elements: every workgroup ends with ONE atomic on
// the norm accumulator, and same-address atomics complete at ~15 ns apiece whatever their type -- 2 048 workgroups made the
// 9 us AdamW pass a 33 us kernel (1 024 made the norm of a 6 MB arena 15 us).
constexpr int OPT_WGS = 256, OPT_THREADS = 512;

__device__ __forceinline__ void block_sum_atomic(float acc, float* out, float mul) {
  __shared__ float red[OPT_THREADS / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
    atomicAdd(out, t * mul);
  }
}

// out[0] += scale^2 * sum g^2   (caller zeroes out; sqrt on the host or in the consumer)
__global__ __launch_bounds__(OPT_THREADS) void sqnorm_kernel(const float* __restrict__ g, long n, float scale, float* __restrict__ out) {
  float acc = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; acc += v * v; }
  block_sum_atomic(acc, out, scale * scale);
}

// torch.optim.AdamW (amsgrad=False, maximize=False), decoupled weight decay, bias corrections passed in:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// g is read as grad_scale * g (1/world_size after a SUM all-reduce).  Returns (grad_scale g)^2 for the norm.
__device__ __forceinline__ float adamw_one(float& p, float g, float& m, float& v, float lr, float wd, float step, float b1,
                                           float b2, float eps, float sqrt_bc2, float grad_scale) {
  const float gi = g * grad_scale;
  float pi = p * (1.f - lr * wd);
  const float mi = b1 * m + (1.f - b1) * gi;
  const float vi = b2 * v + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / sqrt_bc2 + eps;
  pi -= step * (mi / denom);
  p = pi; m = mi; v = vi;
  return gi * gi;
}
__device__ __forceinline__ float adamw_sweep(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, long n, float lr, float wd, float step, float b1, float b2,
                                             float eps, float sqrt_bc2, float grad_scale) {
  float acc = 0.f;
  const long n4 = n >> 2;
#pragma unroll 2
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pv[k], mk = mv[k], vk = vv[k];
      acc += adamw_one(pk, gv[k], mk, vk, lr, wd, step, b1, b2, eps, sqrt_bc2, grad_scale);
      pv[k] = pk; mv[k] = mk; vv[k] = vk;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = n4 * 4 + threadIdx.x;
    acc += adamw_one(p[i], g[i], m[i], v[i], lr, wd, step, b1, b2, eps, sqrt_bc2, grad_scale);
  }
  return acc;
}
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                            float wd, float bc1, float sqrt_bc2, float grad_scale) {
  (void)adamw_sweep(p, g, m, v, n, lr, wd, lr / bc1, b1, b2, eps, sqrt_bc2, grad_scale);
}

// The same with the per-step scalars in DEVICE memory (hyper = [lr, bc1, sqrt(bc2)]): a hipGraph of the training step is
// captured once and replayed with a new learning rate / bias correction every step.
// sqnorm (optional): += sum (grad_scale g)^2 -- the gradient norm the step body reports (main.py:188-193) rides along in
// the same pass over g instead of a launch of its own.
__global__ __launch_bounds__(OPT_THREADS) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, long n,
                                                                const float* __restrict__ hyper, float b1, float b2, float eps,
                                                                float wd, float grad_scale, float* __restrict__ sqnorm) {
  const float lr = hyper[0], bc1 = hyper[1], sqrt_bc2 = hyper[2];
  const float acc = adamw_sweep(p, g, m, v, n, lr, wd, lr / bc1, b1, b2, eps, sqrt_bc2, grad_scale);
  if (sqnorm != nullptr) block_sum_atomic(acc, sqnorm, 1.f);
}

// workgroups for n floats walked as 16-byte elements, at most `cap`
static long opt_grid(long n, long cap) {
  const long b = (n / 4 + OPT_THREADS - 1) / OPT_THREADS;
  return b < 1 ? 1 : (b < cap ? b : cap);
}
}  // namespace

extern "C" int wmz_adamw_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, double beta1,
                                  double beta2, double eps, double weight_decay, double grad_scale, float* sqnorm_out,
                                  void* stream) {
  WMZ_REQUIRE(p && g && m && v && hyper && n > 0, "wmz_adamw_step_dev: bad arguments");
  WMZ_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "wmz_adamw_step_dev: arrays must be 16-byte aligned");
  hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)opt_grid(n, sqnorm_out ? OPT_WGS : 2048)), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, g,
                     m, v, n, hyper, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)grad_scale, sqnorm_out);
  WMZ_LAUNCH_CHECK("wmz_adamw_step_dev");
  return WMZ_OK;
}

extern "C" int wmz_grad_sqnorm(const float* g, long n, float scale, float* out, void* stream) {
  WMZ_REQUIRE(g && out && n > 0, "wmz_grad_sqnorm: bad arguments");
  WMZ_REQUIRE(((uintptr_t)g & 15) == 0, "wmz_grad_sqnorm: arena must be 16-byte aligned");
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)opt_grid(n, OPT_WGS)), dim3(OPT_THREADS), 0, (hipStream_t)stream, g, n, scale, out);
  WMZ_LAUNCH_CHECK("wmz_grad_sqnorm");
  return WMZ_OK;
}

extern "C" int wmz_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                              double eps, double weight_decay, long step, double grad_scale, void* stream) {
  WMZ_REQUIRE(p && g && m && v && n > 0 && step > 0, "wmz_adamw_step: bad arguments");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  WMZ_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "wmz_adamw_step: arrays must be 16-byte aligned");
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)opt_grid(n, 2048)), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, g,
                     m, v, n, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)bc1,
                     (float)sqrt(bc2), (float)grad_scale);
  WMZ_LAUNCH_CHECK("wmz_adamw_step");
  return WMZ_OK;
}
