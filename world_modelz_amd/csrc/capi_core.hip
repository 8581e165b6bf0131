// Error plumbing and version of libwmz_hip.so, plus the token/position embedding kernel.
#include "wmz_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void wmz_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int wmz_version(void) { return WMZ_VERSION; }
extern "C" const char* wmz_last_error(void) { return g_err; }

namespace {

// x[b,s,h,w,:] = emb[z] + ((pos_s[s] + pos_h[h]) + pos_w[w])   (local_3d_attention.py:140-157)
// one 16-byte-wide lane per 4 channels; tables are fp32 and tiny (L2-resident), x is written once.
template <typename T>
__global__ __launch_bounds__(256) void embed_pos3d_kernel(const int64_t* __restrict__ z, const float* __restrict__ emb,
                                                          const float* __restrict__ ps, const float* __restrict__ ph,
                                                          const float* __restrict__ pw, T* __restrict__ x, long ntok,
                                                          int S, int H, int W, int D, int num_classes) {
  const int d4 = D >> 2;
  const long total = ntok * d4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i / d4;
    const int c = (int)(i - t * d4) * 4;
    const int w = (int)(t % W);
    const int h = (int)((t / W) % H);
    const int s = (int)((t / ((long)W * H)) % S);
    long tok = z[t];
    tok = tok < 0 ? 0 : (tok >= num_classes ? num_classes - 1 : tok);   // reference would raise; stay in bounds
    const f32x4 e = *reinterpret_cast<const f32x4*>(emb + tok * D + c);
    const f32x4 a = *reinterpret_cast<const f32x4*>(ps + (long)s * D + c);
    const f32x4 bq = *reinterpret_cast<const f32x4*>(ph + (long)h * D + c);
    const f32x4 cw = *reinterpret_cast<const f32x4*>(pw + (long)w * D + c);
    const f32x4 v = e + ((a + bq) + cw);
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(x + t * D + c) = v;
    } else {
      s16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (short)f32_to_bf16_bits(v[k]);
      *reinterpret_cast<s16x4*>(x + t * D + c) = o;
    }
  }
}

}  // namespace

extern "C" int wmz_embed_pos3d_fwd(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                   const float* pos_w, void* x, int B, int S, int H, int W, int D, int num_classes,
                                   int dtype, void* stream) {
  WMZ_REQUIRE(z && emb && pos_s && pos_h && pos_w && x, "wmz_embed_pos3d_fwd: null tensor");
  WMZ_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && D > 0 && num_classes > 0, "wmz_embed_pos3d_fwd: bad shape");
  WMZ_REQUIRE(D % 4 == 0, "wmz_embed_pos3d_fwd: D must be a multiple of 4 (got %d)", D);
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_embed_pos3d_fwd: bad dtype %d", dtype);
  const long ntok = (long)B * S * H * W;
  const long total = ntok * (D / 4);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == WMZ_BF16)
    hipLaunchKernelGGL(embed_pos3d_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, z, emb, pos_s, pos_h, pos_w,
                       (bf16_t*)x, ntok, S, H, W, D, num_classes);
  else
    hipLaunchKernelGGL(embed_pos3d_kernel<float>, dim3(grid), dim3(256), 0, st, z, emb, pos_s, pos_h, pos_w,
                       (float*)x, ntok, S, H, W, D, num_classes);
  WMZ_LAUNCH_CHECK("wmz_embed_pos3d_fwd");
  return WMZ_OK;
}
