// VectorQuantizerEMA nearest code (vq-video-diffusion/vq.py:30-33, :77-87) by SCREENING + EXACT RE-CHECK, embedding_dim 64.
// Compiled with -ffp-contract=off like vq.hip.  Results are bit-identical to wmz_vq_argmin's (indices and minimum distances
// in the reference's fp32 summation order): the screening only decides WHICH code is evaluated exactly.
//
// Why: the pinned distance is three un-fused fp32 lane operations per (row, code, element) -- 12.9 G lane-ops at N = 65 536,
// C = 1 024: 164 us at the vector ALU's peak however it is scheduled (vq.hip runs at 0.55-0.62 of it).  But the ARGMIN needs
// that arithmetic only where two codes are closer than what cheaper arithmetic can tell apart:
//   d(n, c) = |x_n|^2 + |e_c|^2 - 2 x_n.e_c   =>   argmin_c d = argmax_c a(n, c),   a = x_n.e_c - |e_c|^2 / 2.
// a~ is computed on the matrix cores with x and e split into bf16 head + tail (x.e ~= xh.eh + xh.el + xl.eh: three bf16 MFMAs
// per 16 elements, fp32 accumulate, -|e_c|^2 / 2 as the accumulator's initial value), i.e. 3/16 of the cost of an fp32 MFMA
// product, with the error bound (u = 2^-8, the bf16 unit round-off; |.| Euclidean norms; emax = max_c |e_c|)
//   |a~ - a_pinned| <= eps_n = 8e-5 |x_n| emax + 1e-5 (|x_n| + emax)^2
//     (dropped tail products and double-rounded tails <= 3.1 u^2 |x||e|; fp32 accumulation of 193 terms in any order
//      <= 1.2e-5 (|x||e| + |e|^2 / 2); |e|^2 in fp32 <= 4e-6 |e|^2; the pinned distance itself is within 2e-6 (|x| + |e|)^2
//      of the real-number distance -- the constants above carry > 30 % margin over the sum).
// Every lane keeps the largest and second-largest a~ of the codes it sees.  If the row's best beats its runner-up by more
// than 2 eps_n, NO other code can be the pinned argmin (strictly: ties are impossible then), and the row costs one exact
// evaluation (its minimum distance).  Otherwise (equal codes, near ties: ~0.4 % of rows on Gaussian data) the row goes to a
// list, and a second kernel scans the whole codebook for it in the pinned arithmetic, one wave per row.
#include "wmz_common.h"
#include <limits.h>

namespace {

constexpr int E = 64;
constexpr int ROWS_W = 32;            // rows per wave (the MFMA's N)
constexpr int NWAVE = 4;
constexpr int ROWS_WG = ROWS_W * NWAVE;
constexpr int CT = 64;                // codes per LDS tile (two 32-code MFMA groups)
constexpr int CROW = 2 * E + 16;      // LDS pitch of a bf16 code row: 32 codes on the lanes of ds_read_b128 -> conflict-free
constexpr int TILE_B = 2 * CT * CROW + CT * 4;      // eh rows, el rows, -|e|^2/2

// sum_e (x[e]-c[e])^2 in ATen's AVX order (see vq.hip::dist_exact; E = 64: 8 lanes x 4 accumulators x 2 rounds)
template <typename XF, typename CF>
__device__ __forceinline__ float dist_pinned64(XF xf, CF cf) {
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xf(8 * k + j) - cf(8 * k + j); acc[k][j] = d * d; }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = 8 * (4 + k) + j;
      const float d = xf(e) - cf(e);
      acc[k][j] = acc[k][j] + d * d;
    }
  float fin = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) fin = fin + (((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j]);
  return fin;
}

// codebook -> bf16 head / tail rows, -|e|^2 / 2, and the largest |e|^2 of every block of 64 codes (the screening kernel takes
// the maximum of those: no same-address atomic chain); block 0 also puts the flagged-row count back to zero.
// One wave per 16 codes: lane = (code, quarter row), 16 elements each.
__global__ __launch_bounds__(256) void vq_prep_kernel(const float* __restrict__ CB, unsigned short* __restrict__ EH,
                                                      unsigned short* __restrict__ EL, float* __restrict__ NH,
                                                      float* __restrict__ emax2_blk, int* __restrict__ nflag, int C) {
  __shared__ float wmax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x * 64 + (tid >> 2), q = tid & 3;            // C is a multiple of 64
  const float* row = CB + (long)c * E + q * 16;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    s16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned short hb = f32_to_bf16_bits(v[j]);
      h[j] = (short)hb;
      l[j] = (short)f32_to_bf16_bits(v[j] - bf16_bits_to_f32(hb));
      s = fmaf(v[j], v[j], s);
    }
    *reinterpret_cast<s16x4*>(EH + (long)c * E + q * 16 + i) = h;
    *reinterpret_cast<s16x4*>(EL + (long)c * E + q * 16 + i) = l;
  }
  s += __shfl_xor(s, 1);                                    // the code's four lanes are neighbours: lanes 4k .. 4k + 3
  s += __shfl_xor(s, 2);
  if (q == 0) NH[c] = -0.5f * s;
  float m = s;
#pragma unroll
  for (int d = 4; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
  if (lane == 0) wmax[wave] = m;
  __syncthreads();
  if (tid == 0) {
    emax2_blk[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (blockIdx.x == 0) *nflag = 0;
  }
}

__global__ __launch_bounds__(ROWS_WG * 2) void vq_screen_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ CB,
                                                                const unsigned short* __restrict__ EH,
                                                                const unsigned short* __restrict__ EL, const float* __restrict__ NH,
                                                                const float* __restrict__ emax2_blk, int64_t* __restrict__ IDX,
                                                                float* __restrict__ DMIN, int* __restrict__ nflag,
                                                                int* __restrict__ flagged, int N, int C) {
  __shared__ __attribute__((aligned(16))) char sm[2 * TILE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n32 = lane & 31, hi = lane >> 5;
  const long row = (long)blockIdx.x * ROWS_WG + wave * ROWS_W + n32;
  const bool rok = row < N;
  const float* xrow = X + (rok ? row : (long)N - 1) * ldx;

  // B operands: this lane's 8 k-elements of every 16-wide k-step, split into bf16 head and tail
  s16x8 xh[4], xl[4];
  float x2 = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = xrow[ks * 16 + 8 * hi + j];
      const unsigned short h = f32_to_bf16_bits(v);
      xh[ks][j] = (short)h;
      xl[ks][j] = (short)f32_to_bf16_bits(v - bf16_bits_to_f32(h));
      x2 = x2 + v * v;
    }
  }
  x2 = wave_halves_sum(x2);                                // |x_n|^2 (both lanes of the row)

  // staging: 256 threads move one tile = 64 codes x (128 B head + 128 B tail) + 64 floats; chunk q = tid + 256 i
  auto stage_load = [&](int c0, i32x4 (&rh)[2], i32x4 (&rl)[2], f32x4& rn) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i, code = q >> 3, ch = q & 7;
      rh[i] = *reinterpret_cast<const i32x4*>(EH + (long)(c0 + code) * E + ch * 8);
      rl[i] = *reinterpret_cast<const i32x4*>(EL + (long)(c0 + code) * E + ch * 8);
    }
    if (tid < 16) rn = *reinterpret_cast<const f32x4*>(NH + c0 + tid * 4);
  };
  auto stage_store = [&](char* buf, const i32x4 (&rh)[2], const i32x4 (&rl)[2], const f32x4& rn) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i, code = q >> 3, ch = q & 7;
      *reinterpret_cast<i32x4*>(buf + code * CROW + ch * 16) = rh[i];
      *reinterpret_cast<i32x4*>(buf + CT * CROW + code * CROW + ch * 16) = rl[i];
    }
    if (tid < 16) *reinterpret_cast<f32x4*>(buf + 2 * CT * CROW + tid * 16) = rn;
  };

  // best, runner-up, and WHERE the best was seen: accumulator register (rsel) and 32-code group (gsel) -- the code index is put
  // together once, at the end (a per-element index costs as much vector ALU time as the comparison itself)
  float M1 = -INFINITY, M2 = -INFINITY;
  int rsel = 0, gsel = -1;
  i32x4 rh[2], rl[2];
  f32x4 rn = (f32x4)(0.f);
  stage_load(0, rh, rl, rn);
  stage_store(sm, rh, rl, rn);
  __syncthreads();
  const int ntile = C / CT;
  for (int t = 0; t < ntile; ++t) {
    const char* buf = sm + (t & 1) * TILE_B;
    if (t + 1 < ntile) stage_load((t + 1) * CT, rh, rl, rn);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // initial accumulator: -|e_c|^2 / 2 of the 16 codes this lane's registers stand for: c = 32 g + (r & 3) + 8 (r >> 2) + 4 hi
      f32x16 acc;
      const float* nh = reinterpret_cast<const float*>(buf + 2 * CT * CROW) + 32 * g + 4 * hi;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(nh + 8 * q4);
        acc[4 * q4] = v[0]; acc[4 * q4 + 1] = v[1]; acc[4 * q4 + 2] = v[2]; acc[4 * q4 + 3] = v[3];
      }
      const char* ah = buf + (32 * g + n32) * CROW + hi * 16;
      const char* al = ah + CT * CROW;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const s16x8 eh = *reinterpret_cast<const s16x8*>(ah + ks * 32);
        const s16x8 el = *reinterpret_cast<const s16x8*>(al + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(el, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, xl[ks], acc, 0, 0, 0);
      }
      const float m_before = M1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        M2 = __builtin_amdgcn_fmed3f(M1, v, M2);           // the runner-up: M1 <= .. is kept, v in between replaces it
        rsel = v > M1 ? r : rsel;
        M1 = __builtin_amdgcn_fmed3f(M1, v, INFINITY);     // max(M1, v) as one instruction (fmaxf adds two canonicalising ops)
      }
      gsel = M1 != m_before ? 2 * t + g : gsel;
    }
    if (t + 1 < ntile) stage_store(sm + ((t + 1) & 1) * TILE_B, rh, rl, rn);
    __syncthreads();
  }
  int c1 = gsel < 0 ? INT_MAX : 32 * gsel + 4 * hi + (rsel & 3) + 8 * (rsel >> 2);
  // the row's two lanes (hi = 0 / 1 saw disjoint codes) merge: best, runner-up, index (lower index on equal values)
  {
    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(M1), __float_as_uint(M1), false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(M2), __float_as_uint(M2), false, false);
    const auto s3 = __builtin_amdgcn_permlane32_swap((unsigned)c1, (unsigned)c1, false, false);
    const float o1 = __uint_as_float(hi ? s1[0] : s1[1]), o2 = __uint_as_float(hi ? s2[0] : s2[1]);
    const int oc = (int)(hi ? s3[0] : s3[1]);
    const float best = fmaxf(M1, o1);
    const float second = fmaxf(fminf(M1, o1), fmaxf(M2, o2));
    c1 = (M1 > o1 || (M1 == o1 && c1 < oc)) ? c1 : oc;
    M1 = best; M2 = second;
  }
  if (hi == 0 && rok) {
    float e2 = 0.f;
    for (int i = 0; i < C / 64; ++i) e2 = fmaxf(e2, emax2_blk[i]);   // (wave-uniform addresses: scalar loads)
    const float xn = sqrtf(x2) * 1.0001f, em = sqrtf(e2) * 1.0001f;
    const float eps = 8e-5f * xn * em + 1e-5f * (xn + em) * (xn + em);
    const bool sure = (M1 - M2) > 2.f * eps && M1 < INFINITY && x2 < INFINITY;   // (false for NaN / inf anywhere)
    if (sure) {
      IDX[row] = c1;
      if (DMIN != nullptr) {
        const float* crow = CB + (long)c1 * E;
        DMIN[row] = dist_pinned64([&](int e) { return xrow[e]; }, [&](int e) { return crow[e]; });
      }
    } else {
      flagged[atomicAdd(nflag, 1)] = (int)row;
    }
  }
}

// rows the screening could not decide: the whole codebook in the pinned arithmetic -- exactly what vq_argmin_kernel computes for
// the row.  A block takes ONE flagged row, its four waves a quarter of the codebook each: tiles of 64 codes go through a
// WAVE-PRIVATE LDS image (coalesced 16-byte loads, the next tile's in flight while this one is evaluated; rows pitched 65 floats:
// lane = code reads its row conflict-free; no block barrier per tile); lexicographic (distance, index) minimum over the wave, then
// over the four waves.  (Round 5: one wave per row walked all C / 64 tiles behind two block barriers each -- a ~30-us critical
// path for ~260 rows of work; now C / 256 tile steps per wave, on a grid that gives every flagged row its own workgroup.)
constexpr int RC_TILE = 64 * 65;                 // floats of one wave's tile image
constexpr int RC_LDS = 4 * RC_TILE * 4 + 64;     // bytes: four images + the waves' minima

__global__ __launch_bounds__(256) void vq_recheck_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ CB,
                                                         int64_t* __restrict__ IDX, float* __restrict__ DMIN,
                                                         const int* __restrict__ nflag, const int* __restrict__ flagged, int C) {
  extern __shared__ __attribute__((aligned(16))) float rc_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const tile = rc_lds + wave * RC_TILE;
  float* const sd = rc_lds + 4 * RC_TILE;
  int* const sc = reinterpret_cast<int*>(sd + 4);
  const int n = *nflag;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {                       // block-uniform trip count: barriers inside are safe
    const int row = __builtin_amdgcn_readfirstlane(flagged[i]);
    const float* xrow = X + (long)row * ldx;                              // wave-uniform: scalar loads
    float xr[E];
#pragma unroll
    for (int e = 0; e < E; ++e) xr[e] = xrow[e];
    float bd = INFINITY;
    int bc = INT_MAX;
    // a tile = 64 codes x 16 chunks of 16 bytes: chunk q = lane + 64 k belongs to code q >> 4, columns 4 (q & 15) ..
    f32x4 nxt[16];
    auto fetch = [&](int c0) {
      const float* src = CB + (long)min(c0, C - 64) * E + lane * 4;       // (past the codebook: a valid tile, never evaluated)
#pragma unroll
      for (int k = 0; k < 16; ++k) nxt[k] = *reinterpret_cast<const f32x4*>(src + 256 * k);
    };
    fetch(64 * wave);
    for (int c0 = 64 * wave; c0 < C; c0 += 256) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int q = lane + 64 * k;
        float* d = tile + (q >> 4) * 65 + (q & 15) * 4;
        d[0] = nxt[k][0]; d[1] = nxt[k][1]; d[2] = nxt[k][2]; d[3] = nxt[k][3];
      }
      fetch(c0 + 256);
      __builtin_amdgcn_sched_barrier(0);                                  // (the requests stay above the evaluation: hipcc would sink them to their use)
      __builtin_amdgcn_wave_barrier();
      const float* cr = tile + lane * 65;
      const float d = dist_pinned64([&](int e) { return xr[e]; }, [&](int e) { return cr[e]; });
      if (d < bd) { bd = d; bc = c0 + lane; }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float od = __shfl_xor(bd, m);
      const int oc = __shfl_xor(bc, m);
      if (od < bd || (od == bd && oc < bc)) { bd = od; bc = oc; }
    }
    if (lane == 0) { sd[wave] = bd; sc[wave] = bc; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float od = sd[w];
        const int oc = sc[w];
        if (od < bd || (od == bd && oc < bc)) { bd = od; bc = oc; }
      }
      IDX[row] = bc == INT_MAX ? 0 : bc;
      if (DMIN != nullptr) DMIN[row] = bd;
    }
    __syncthreads();
  }
}

}  // namespace

// Workspace layout: [16 B: flag count at +4] [C x 64 bf16 heads] [C x 64 bf16 tails] [C floats -|e|^2/2] [C/64 floats: block maxima of
// |e|^2] [N ints flagged rows]
extern "C" long wmz_vq_argmin_screened_workspace_bytes(int N, int C, int E) {
  if (E != 64 || C < 64 || C % 64 != 0 || N <= 0) return 0;            // 0: shape not built, use wmz_vq_argmin
  return 16 + (long)C * E * 2 * 2 + (long)C * 4 + (long)(C / 64) * 4 + (long)N * 4;
}

extern "C" int wmz_vq_argmin_screened(const float* x, long ldx, const float* codebook, int64_t* idx, float* dist_min, int N, int C,
                                      int E, void* workspace, long workspace_bytes, void* stream) {
  WMZ_REQUIRE(x && codebook && idx && workspace, "wmz_vq_argmin_screened: null tensor");
  const long need = wmz_vq_argmin_screened_workspace_bytes(N, C, E);
  if (need == 0) { wmz_set_error("wmz_vq_argmin_screened: built for embedding_dim 64 and a multiple of 64 codes (got E=%d C=%d)", E, C); return WMZ_ERR_UNSUPPORTED; }
  WMZ_REQUIRE(workspace_bytes >= need, "wmz_vq_argmin_screened: workspace %ld < %ld bytes", workspace_bytes, need);
  WMZ_REQUIRE(ldx % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0, "wmz_vq_argmin_screened: x rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int* nflag = (int*)(ws + 4);
  unsigned short* EH = (unsigned short*)(ws + 16);
  unsigned short* EL = EH + (long)C * E;
  float* NH = (float*)(EL + (long)C * E);
  float* emax2_blk = NH + C;
  int* flagged = (int*)(emax2_blk + C / 64);
  // (no hipMemsetAsync here: the call is captured into hipGraphs, where a memset node in front of kernels reading its target
  //  proved unreliable on replay -- the prep kernel zeroes the flag count)
  hipLaunchKernelGGL(vq_prep_kernel, dim3(C / 64), dim3(256), 0, st, codebook, EH, EL, NH, emax2_blk, nflag, C);
  hipLaunchKernelGGL(vq_screen_kernel, dim3(wmz_cdiv(N, ROWS_WG)), dim3(ROWS_WG * 2), 0, st, x, ldx, codebook, EH, EL, NH, emax2_blk,
                     idx, dist_min, nflag, flagged, N, C);
  static std::atomic<uint64_t> attr_devs{0};                       // (> 64 KB of dynamic LDS has to be asked for once per device)
  if (wmz_first_use_on_device(attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_recheck_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
  }
  // (the flagged-row count lives on the device: a grid that gives ~0.4 % of the rows a workgroup each, the rest leave at once)
  const int rgrid = N / 128 < 256 ? 256 : (N / 128 > 2048 ? 2048 : N / 128);
  hipLaunchKernelGGL(vq_recheck_kernel, dim3(rgrid), dim3(256), RC_LDS, st, x, ldx, codebook, idx, dist_min, nflag, flagged, C);
  WMZ_LAUNCH_CHECK("wmz_vq_argmin_screened");
  return WMZ_OK;
}
