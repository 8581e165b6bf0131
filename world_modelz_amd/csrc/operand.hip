// MFMA-operand copies of the fp32 parameters, all of them in ONE launch.
// The training step rewrites every weight (wmz_adamw_step), after which the bf16 casts, the transposes the dgrad GEMMs
// read and the k|v concatenations are stale.  Rebuilding them with tensor ops was ~60 small launches per step; this is a
// table-driven kernel: entry i turns the logical matrix [rows0 + rows1, cols] = (src0 ; src1) (src1 optional: row
// concatenation; src0 NULL: zeros) into dst, row-major or transposed, bf16 or fp32.
#include "wmz_common.h"

namespace {

struct OpDesc { const float* s0; const float* s1; int r0, r1, cols, flags; void* dst; long start; };
struct OpTable { OpDesc d[64]; int n; long total; };

__global__ __launch_bounds__(256) void operands_refresh_kernel(OpTable T) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;          // one source element per thread, source-major (coalesced reads)
  if (e >= T.total) return;
  int i = 0;
#pragma unroll 1
  for (int k = 1; k < T.n; ++k) if (e >= T.d[k].start) i = k;
  const OpDesc D = T.d[i];
  const long idx = e - D.start;
  const int r = (int)(idx / D.cols), c = (int)(idx - (long)r * D.cols);
  const int R = D.r0 + D.r1;
  float v = 0.f;
  if (r < D.r0) { if (D.s0) v = D.s0[(long)r * D.cols + c]; }
  else if (D.s1) v = D.s1[(long)(r - D.r0) * D.cols + c];
  const long o = (D.flags & WMZ_OPERAND_TRANSPOSE) ? (long)c * R + r : idx;
  if (D.flags & WMZ_OPERAND_F32) reinterpret_cast<float*>(D.dst)[o] = v;
  else reinterpret_cast<bf16_t*>(D.dst)[o] = __float2bfloat16(v);
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
    off += (long)(rows0[i] + rows1[i]) * cols[i];
  }
  T.n = n; T.total = off;
  hipLaunchKernelGGL(operands_refresh_kernel, dim3((unsigned)wmz_cdiv(off, 256)), dim3(256), 0, (hipStream_t)stream, T);
  WMZ_LAUNCH_CHECK("wmz_operands_refresh");
  return WMZ_OK;
}
