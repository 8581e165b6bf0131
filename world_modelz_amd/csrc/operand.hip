// MFMA-operand copies of the fp32 parameters, all of them in ONE launch.
// The training step rewrites every weight (wmz_adamw_step), after which the bf16 casts, the transposes the dgrad GEMMs
// read and the k|v concatenations are stale.  Rebuilding them with tensor ops was ~60 small launches per step; this is a
// table-driven kernel: entry i turns the logical matrix [rows0 + rows1, cols] = (src0 ; src1) (src1 optional: row
// concatenation; src0 NULL: zeros) into dst, row-major or transposed, bf16 or fp32.
#include "wmz_common.h"

namespace {

struct OpDesc { const float* s0; const float* s1; int r0, r1, cols, flags; void* dst; long start; };   // start: first 64x64 tile
struct OpTable { OpDesc d[64]; int n; long total; };

// One workgroup per 64 x 64 tile of a logical matrix: rows are read coalesced, a transposed destination is written
// coalesced too (through LDS) -- the element-per-thread version wrote transposes 2 bytes at a stride of a whole row.
__global__ __launch_bounds__(256) void operands_refresh_kernel(OpTable T) {
  __shared__ float tile[64][65];
  const long tb = blockIdx.x;
  int lo = 0, hi = T.n - 1;                                     // last entry with start <= tb (uniform per workgroup)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.d[mid].start <= tb) lo = mid; else hi = mid - 1;
  }
  const OpDesc D = T.d[lo];
  const int R = D.r0 + D.r1;
  const int tcols = (D.cols + 63) >> 6;
  const int t = (int)(tb - D.start);
  const int r0 = (t / tcols) * 64, c0 = (t % tcols) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const bool tr = (D.flags & WMZ_OPERAND_TRANSPOSE) != 0, f32 = (D.flags & WMZ_OPERAND_F32) != 0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < D.cols) {
      if (r < D.r0) { if (D.s0) v = D.s0[(long)r * D.cols + c]; }
      else if (D.s1) v = D.s1[(long)(r - D.r0) * D.cols + c];
    }
    if (tr) tile[ty + 4 * i][tx] = v;
    else if (r < R && c < D.cols) {
      const long o = (long)r * D.cols + c;
      if (f32) reinterpret_cast<float*>(D.dst)[o] = v; else reinterpret_cast<bf16_t*>(D.dst)[o] = __float2bfloat16(v);
    }
  }
  if (!tr) return;                                              // (uniform)
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;                 // destination row = source column
    if (c < D.cols && r < R) {
      const float v = tile[tx][ty + 4 * i];
      const long o = (long)c * R + r;
      if (f32) reinterpret_cast<float*>(D.dst)[o] = v; else reinterpret_cast<bf16_t*>(D.dst)[o] = __float2bfloat16(v);
    }
  }
}

// The conv encoder / decoder's GEMM operands (autoencoder.py:_w_op / _wT_op), every layer in ONE launch:
//   mode 0  forward / weight-gradient layout  [Co8, KH*KW*Ci8]: element (co, tap, c) = w[co, c, tap]         (c >= Ci, co >= Co: 0)
//   mode 1  data-gradient layout              [Ci8, KH*KW*Co8]: element (ci, tap', co) = w[co, ci, KK-1-tap'] (taps flipped)
// from nn.Conv2d's fp32 weight [Co, Ci, KH, KW].  Built with tensor ops these were ~6 launches per convolution and step.
// pack != 0 (bf16 only): the SAME logical matrix [rows, K] (rows = Co | Ci8, K = kk * Ci8 | kk * Co8) written in the fragment order
// of the direct kernels instead of row-major: 1 = csrc/conv_direct.hip (wmz_conv3x3_direct_pack: [pass][tap][k-step][block][lane][8]),
// 2 = csrc/conv_point.hip (wmz_conv_point_pack: [k-step][block][lane][8], K zero-padded to a multiple of 64)
struct ConvOpDesc { const float* w; void* dst; int co, ci, kk, mode, f32, pack; long start, count; };
struct ConvOpTable { ConvOpDesc d[48]; int n; long total; };

__device__ __forceinline__ float conv_op_value(const ConvOpDesc& D, int row, int k) {
  const int ci8 = (D.ci + 7) & ~7, co8 = (D.co + 7) & ~7;
  if (D.mode == 0) {
    const int tap = k / ci8, c = k - tap * ci8;
    return (row < D.co && tap < D.kk && c < D.ci) ? D.w[((long)row * D.ci + c) * D.kk + tap] : 0.f;
  }
  const int tap = k / co8, co = k - tap * co8;
  return (row < D.ci && tap < D.kk && co < D.co) ? D.w[((long)co * D.ci + row) * D.kk + (D.kk - 1 - tap)] : 0.f;
}

__global__ __launch_bounds__(256) void conv_operands_refresh_kernel(ConvOpTable T) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T.total; i += (long)gridDim.x * 256) {
    int lo = 0, hi = T.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (T.d[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const ConvOpDesc D = T.d[lo];
    const long e = i - D.start;
    const int ci8 = (D.ci + 7) & ~7, co8 = (D.co + 7) & ~7;
    if (D.pack != 0) {
      const int rows = D.mode == 0 ? co8 : ci8;                    // the packed convolution's Cout
      const int kin = D.mode == 0 ? ci8 : co8;                     // ... and Cin
      const int ncb = rows <= 64 ? 2 : 4;
      const int e8 = (int)(e & 7), lane = (int)((e >> 3) & 63);
      long r = e >> 9;
      const int j = (int)(r % ncb); r /= ncb;
      int k;
      if (D.pack == 1) {
        const int ks = (int)(r & 3); r >>= 2;
        const int tap = (int)(r % 9), pass = (int)(r / 9);
        k = tap * kin + 64 * pass + 16 * ks + 8 * (lane >> 5) + e8;
      } else {
        k = 16 * (int)r + 8 * (lane >> 5) + e8;
        if (k >= D.kk * kin) k = -1;
      }
      const float pv = k < 0 ? 0.f : conv_op_value(D, 32 * j + (lane & 31), k);
      reinterpret_cast<bf16_t*>(D.dst)[e] = __float2bfloat16(pv);
      continue;
    }
    float v = 0.f;
    if (D.mode == 0) {
      const int rowlen = D.kk * ci8;
      const int n = (int)(e / rowlen), rem = (int)(e - (long)n * rowlen);
      const int tap = rem / ci8, c = rem - tap * ci8;
      if (c < D.ci && n < D.co) v = D.w[((long)n * D.ci + c) * D.kk + tap];           // (rows Co .. Co8 - 1: zero padding)
    } else {
      const int rowlen = D.kk * co8;
      const int r = (int)(e / rowlen), rem = (int)(e - (long)r * rowlen);
      const int tap = rem / co8, co = rem - tap * co8;
      if (r < D.ci && co < D.co) v = D.w[((long)co * D.ci + r) * D.kk + (D.kk - 1 - tap)];
    }
    if (D.f32) reinterpret_cast<float*>(D.dst)[e] = v; else reinterpret_cast<bf16_t*>(D.dst)[e] = __float2bfloat16(v);
  }
}

}  // namespace

extern "C" int wmz_conv_operands_refresh(const void* const* weight, void* const* dst, const int* co, const int* ci, const int* kk,
                                         const int* mode, int n, int dtype, void* stream) {
  return wmz_conv_operands_refresh_packed(weight, dst, co, ci, kk, mode, nullptr, n, dtype, stream);
}

extern "C" int wmz_conv_operands_refresh_packed(const void* const* weight, void* const* dst, const int* co, const int* ci, const int* kk,
                                                const int* mode, const int* pack, int n, int dtype, void* stream) {
  WMZ_REQUIRE(n >= 0 && n <= 48, "wmz_conv_operands_refresh: at most 48 operands per call (got %d)", n);
  if (n == 0) return WMZ_OK;
  WMZ_REQUIRE(weight && dst && co && ci && kk && mode, "wmz_conv_operands_refresh: null table");
  WMZ_REQUIRE(dtype == WMZ_F32 || dtype == WMZ_BF16, "wmz_conv_operands_refresh: bad dtype %d", dtype);
  ConvOpTable T;
  long off = 0;
  for (int i = 0; i < n; ++i) {
    WMZ_REQUIRE(weight[i] && dst[i] && co[i] > 0 && ci[i] > 0 && kk[i] > 0 && (mode[i] == 0 || mode[i] == 1), "wmz_conv_operands_refresh: bad entry %d", i);
    const int ci8 = (ci[i] + 7) & ~7, co8 = (co[i] + 7) & ~7;
    T.d[i].w = (const float*)weight[i]; T.d[i].dst = dst[i]; T.d[i].co = co[i]; T.d[i].ci = ci[i]; T.d[i].kk = kk[i];
    T.d[i].mode = mode[i]; T.d[i].f32 = dtype == WMZ_F32; T.d[i].start = off;
    T.d[i].pack = pack ? pack[i] : 0;
    WMZ_REQUIRE(T.d[i].pack >= 0 && T.d[i].pack <= 2 && (T.d[i].pack == 0 || dtype == WMZ_BF16), "wmz_conv_operands_refresh: bad pack kind in entry %d", i);
    const int prows = mode[i] == 0 ? co8 : ci8, pkin = mode[i] == 0 ? ci8 : co8;
    WMZ_REQUIRE(T.d[i].pack == 0 || prows <= 128, "wmz_conv_operands_refresh: packed operands have at most 128 rows (entry %d)", i);
    WMZ_REQUIRE(T.d[i].pack != 1 || (kk[i] == 9 && (pkin == 64 || pkin == 128)), "wmz_conv_operands_refresh: entry %d is not a direct 3x3 shape", i);
    if (T.d[i].pack == 1) T.d[i].count = (long)9 * pkin * (prows <= 64 ? 2 : 4) * 32;
    else if (T.d[i].pack == 2) T.d[i].count = (long)((kk[i] * pkin + 63) / 64) * 64 * (prows <= 64 ? 2 : 4) * 32;
    else
    T.d[i].count = mode[i] == 0 ? (long)co8 * kk[i] * ci8 : (long)ci8 * kk[i] * co8;
    off += T.d[i].count;
  }
  T.n = n; T.total = off;
  const long blocks = (off + 255) / 256;
  hipLaunchKernelGGL(conv_operands_refresh_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, T);
  WMZ_LAUNCH_CHECK("wmz_conv_operands_refresh");
  return WMZ_OK;
}

extern "C" int wmz_operands_refresh(const void* const* src0, const void* const* src1, const int* rows0, const int* rows1,
                                    const int* cols, void* const* dst, const int* flags, int n, void* stream) {
  WMZ_REQUIRE(n >= 0 && n <= 64, "wmz_operands_refresh: at most 64 operands per call (got %d)", n);
  if (n == 0) return WMZ_OK;
  WMZ_REQUIRE(src0 && src1 && rows0 && rows1 && cols && dst && flags, "wmz_operands_refresh: null table");
  OpTable T;
  long off = 0;
  for (int i = 0; i < n; ++i) {
    WMZ_REQUIRE(dst[i] && rows0[i] >= 0 && rows1[i] >= 0 && rows0[i] + rows1[i] > 0 && cols[i] > 0, "wmz_operands_refresh: bad entry %d", i);
    WMZ_REQUIRE(rows1[i] == 0 || src1[i], "wmz_operands_refresh: entry %d has rows1 without src1", i);
    T.d[i].s0 = (const float*)src0[i]; T.d[i].s1 = (const float*)src1[i]; T.d[i].r0 = rows0[i]; T.d[i].r1 = rows1[i];
    T.d[i].cols = cols[i]; T.d[i].flags = flags[i]; T.d[i].dst = dst[i]; T.d[i].start = off;
    off += (long)wmz_cdiv(rows0[i] + rows1[i], 64) * wmz_cdiv(cols[i], 64);
  }
  T.n = n; T.total = off;
  hipLaunchKernelGGL(operands_refresh_kernel, dim3((unsigned)off), dim3(256), 0, (hipStream_t)stream, T);
  WMZ_LAUNCH_CHECK("wmz_operands_refresh");
  return WMZ_OK;
}
