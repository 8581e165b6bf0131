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

}  // namespace

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
