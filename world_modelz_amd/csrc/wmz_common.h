// Shared device/host helpers for libwmz_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <utility>
#include <atomic>

#include "../../include/wmz.h"

// ---------------------------------------------------------------- error plumbing (host)
void wmz_set_error(const char* fmt, ...);

#define WMZ_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) {                                   \
      wmz_set_error(__VA_ARGS__);                    \
      return WMZ_ERR_ARG;                            \
    }                                                \
  } while (0)

// Function attributes (the > 64 KB dynamic-LDS opt-in) are per DEVICE: a process that drives several GPUs has to set them on each.
// True the first time the calling thread's current device is seen through `mask` (thread-safe; setting an attribute twice is harmless).
static inline bool wmz_first_use_on_device(std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
}

#define WMZ_LAUNCH_CHECK(name)                                               \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      wmz_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
      return WMZ_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

static inline int wmz_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- element types
typedef __hip_bfloat16 bf16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;   // 8 bf16 = one 16x16x32 / 32x32x16 operand
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// ---- the 16-bit MFMA operand format of a translation unit.  The kernels move 16-bit elements as raw bits (s16x8 fragments, the
// `bf16_t` storage tag) and touch their VALUE only through the few primitives below, so the number format is decided here, once
// per translation unit: bfloat16 (the speed mode BASELINE.json names) by default; IEEE half when the unit is compiled with
// WMZ_OP16_F16 -- the "precise" fused mode (same MFMA rate, 11 significand bits instead of 8: end-to-end logits 8x closer to the
// reference's fp32, range +-65504).  A precise unit is a second compilation of the same source (csrc/*_f16.hip) whose entry
// points carry the suffix _f16 (WMZ_FN below).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#ifdef WMZ_OP16_F16
#define WMZ_FN(name) name##_f16
constexpr int kOp16Dtype = WMZ_F16;
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
// round-to-nearest-even (v_cvt_f16_f32); beyond +-65504 the result is an infinity, like the hardware conversion
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ s16x4 cvt_pk4_bf16(float a, float b, float c, float d) {
  const f16x2_t lo = __builtin_convertvector((f32x2){a, b}, f16x2_t), hi = __builtin_convertvector((f32x2){c, d}, f16x2_t);
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const u32x2 w = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, w);
}
// acc + value(bits): as ONE v_fma_mix_f32 (the half operand converted inside the instruction, hi / lo half selected by op_sel) --
// no conversion temporaries where a whole operand row is added to the fp32 stream (fused_common.h add_bop)
__device__ __forceinline__ float op16_add_to(float acc, unsigned short b) { return __builtin_fmaf((float)__builtin_bit_cast(_Float16, b), 1.0f, acc); }
__device__ __forceinline__ f32x4 op16_mfma_16x16x32(const s16x8& a, const s16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 op16_mfma_32x32x16(const s16x8& a, const s16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 op16_mfma_16x16x16(const s16x4& a, const s16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
}
#else
#define WMZ_FN(name) name
constexpr int kOp16Dtype = WMZ_BF16;
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
// round-to-nearest-even; the plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  bf16_t h = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&h);
}

// four floats -> four bf16 (round-to-nearest-even) in two registers: two v_cvt_pk_bf16_f32, nothing else
__device__ __forceinline__ s16x4 cvt_pk4_bf16(float a, float b, float c, float d) {
  const bf16x2_t lo = __builtin_convertvector((f32x2){a, b}, bf16x2_t), hi = __builtin_convertvector((f32x2){c, d}, bf16x2_t);
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const u32x2 w = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, w);
}
__device__ __forceinline__ float op16_add_to(float acc, unsigned short b) { return acc + bf16_bits_to_f32(b); }
__device__ __forceinline__ f32x4 op16_mfma_16x16x32(const s16x8& a, const s16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 op16_mfma_32x32x16(const s16x8& a, const s16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 op16_mfma_16x16x16(const s16x4& a, const s16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
#endif

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = WMZ_F32;
  static constexpr int kPerChunk = 4;                  // elements per 16-byte chunk
  __device__ static __forceinline__ float to_f32(float v) { return v; }
  __device__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = WMZ_BF16;
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float to_f32(bf16_t v) { return bf16_bits_to_f32(__builtin_bit_cast(unsigned short, v)); }
  __device__ static __forceinline__ bf16_t from_f32(float v) { return __builtin_bit_cast(bf16_t, f32_to_bf16_bits(v)); }
};

// An MFMA operand fragment of 8 consecutive k-elements (lane-local).
template <typename T> struct Frag8;
template <> struct Frag8<float> { float v[8]; };
template <> struct Frag8<bf16_t> { s16x8 v; };

__device__ __forceinline__ void frag_zero(Frag8<float>& f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = 0.f;
}
__device__ __forceinline__ void frag_zero(Frag8<bf16_t>& f) { f.v = (s16x8)(0); }

// 16-byte chunk loads into a fragment half/whole.  bf16: one chunk = whole fragment; f32: two chunks.
__device__ __forceinline__ void frag_load(Frag8<bf16_t>& f, const bf16_t* p) {
  f.v = *reinterpret_cast<const s16x8*>(p);
}
__device__ __forceinline__ void frag_load(Frag8<float>& f, const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { f.v[i] = a[i]; f.v[4 + i] = b[i]; }
}

// D(16x16) += A(16x32) * B(32x16).  Lane l holds A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15],
// j = 0..7; D: col = l&15, row = 4*(l>>4)+reg.   f32: eight exact-f32 16x16x4 MFMAs (k = 8g+j summed over g).
__device__ __forceinline__ void mma16(f32x4& acc, const Frag8<bf16_t>& a, const Frag8<bf16_t>& b) {
  acc = op16_mfma_16x16x32(a.v, b.v, acc);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag8<float>& a, const Frag8<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// D(32x32) += A(32x16) * B(16x32).  Lane l holds A[row l&31][k = 8*(l>>5)+j], B[k = 8*(l>>5)+j][col l&31];
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
__device__ __forceinline__ void mma32(f32x16& acc, const Frag8<bf16_t>& a, const Frag8<bf16_t>& b) {
  acc = op16_mfma_32x32x16(a.v, b.v, acc);
}
__device__ __forceinline__ void mma32(f32x16& acc, const Frag8<float>& a, const Frag8<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// exact-erf GELU (nn.GELU default) with erf from Abramowitz-Stegun 7.1.26: |erf error| <= 1.5e-7 (fp32 round-off level,
// 4 orders below bf16 resolution) at about a third of libm erff's instruction count.  Shared by every kernel that
// applies GELU or its derivative.
__device__ __forceinline__ float wmz_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return copysignf(1.f - p * t * __expf(-ax * ax), x);
}
__device__ __forceinline__ float wmz_gelu(float v) { return 0.5f * v * (1.f + wmz_erf(v * 0.70710678118654752440f)); }
// d/dv GELU(v) = Phi(v) + v * phi(v)
__device__ __forceinline__ float wmz_dgelu(float v) {
  return 0.5f * (1.f + wmz_erf(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

// GELU for kernels whose output is rounded to bf16 anyway: v * sigmoid(v * (c1 + c3 v^2 + c5 v^4)), coefficients fitted
// (minimax over [-9, 9], tools in DESIGN.md) to the exact-erf GELU: max |error| 2.6e-5 absolute, two orders below the
// bf16 resolution of the result, at 9 instructions (the erf form above costs ~20).  v^2 is clamped at 50: beyond |v| = 7
// the sigmoid argument stays at 3.5 |v| (the polynomial would turn over at |v| = 11).
__device__ __forceinline__ float wmz_gelu_fast(float v) {
  const float w = fminf(v * v, 50.f);
  float p = fmaf(-1.0148166e-3f, w, 0.10677913f);          // -(c5, c3, c1) * log2(e): exp2 of the negated argument
  p = fmaf(p, w, 2.3011176f);
  const float e = __builtin_amdgcn_exp2f(-p * v);
  return v * __builtin_amdgcn_rcpf(1.f + e);
}

// wmz_gelu_fast(v) and its derivative in one go (the backward of the fused feed-forward): with s = sigmoid(u(v)),
// u = v (c1 + c3 v^2 + c5 v^4):  d/dv [v s] = s (1 + v (1 - s) u'(v)).  Max |error| against the exact-erf derivative
// Phi(v) + v phi(v): 1.1e-4 (checked over [-12, 12]); beyond the clamp u' is the clamped polynomial itself.
__device__ __forceinline__ void wmz_gelu_fast_both(float v, float& g, float& d) {
  constexpr float LN2 = 0.6931471805599453f;
  const float vv = v * v;
  const float w = fminf(vv, 50.f);
  float p = fmaf(-1.0148166e-3f, w, 0.10677913f);
  p = fmaf(p, w, 2.3011176f);
  const float e = __builtin_amdgcn_exp2f(-p * v);
  const float s = __builtin_amdgcn_rcpf(1.f + e);
  g = v * s;
  float q = fmaf(-5.f * 1.0148166e-3f * LN2, w, 3.f * 0.10677913f * LN2);
  q = fmaf(q, w, 2.3011176f * LN2);
  q = vv < 50.f ? q : p * LN2;
  d = s * fmaf(v * (1.f - s), q, 1.f);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>) -- for bodies that need the index as a constant
// expression (instruction immediates in inline asm)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// LDS byte address of a pointer into __shared__ memory
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// ds_read_b64_tr_b16 as inline asm.  The builtin form is an "unknown memory access" to hipcc: with an LDS-DMA
// (global_load_lds) in flight it puts s_waitcnt vmcnt(0) in front of every such read, i.e. it drains the slab being
// prefetched before the current one may be used (measured: the prefetch then overlaps nothing).  The asm form is invisible
// to that bookkeeping -- and to lgkmcnt tracking as well: the caller waits with ds_tr_wait() before using the results.
template <int OFF>
__device__ __forceinline__ s16x4 ds_read_tr16_asm(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ void ds_tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// ds_read_b128 as inline asm + a counted wait that names the value it releases (the consumer cannot be scheduled above it): a
// hand-ordered queue of LDS reads, retired one by one while the younger ones are still in flight.
template <int OFF>
__device__ __forceinline__ s16x8 ds_read_b128_asm(unsigned addr) {
  s16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lgkm_wait_for(s16x8& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
// the same queue discipline for the other shapes a hand-ordered LDS queue carries: four floats (an accumulator's initial
// value), one dword, and a wait that releases two / four values at once
template <int OFF>
__device__ __forceinline__ f32x4 ds_read_f32x4_asm(unsigned addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ unsigned ds_read_u32_asm(unsigned addr) {
  unsigned r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N, typename A, typename B>
__device__ __forceinline__ void lgkm_wait_for2(A& a, B& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N, typename A>
__device__ __forceinline__ void lgkm_wait_for4(A& a, A& b, A& c, A& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
// lane id (0..63) that the optimiser can neither hoist out of a loop nor keep alive across it
__device__ __forceinline__ unsigned lane_id_volatile() {
  unsigned r;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(r));
  return r;
}

// Reductions over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48 -> the same result in all four) with the
// gfx950 row / half swaps: two VALU instructions per level, no LDS round trip (__shfl_xor is a ds_bpermute: ~100+ cycles
// of latency each, and with an LDS-DMA in flight hipcc drains vmcnt in front of it).
//   v_permlane16_swap d, s: rows 1, 3 of d <-> rows 0, 2 of s;   v_permlane32_swap d, s: upper half of d <-> lower half of s
__device__ __forceinline__ float wave_groups_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned m = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave_groups_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned m = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_halves_sum(float v) {          // lanes l, l^32
  const unsigned u = __float_as_uint(v);
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Sum over all 64 lanes, result in every lane: four DPP row rotations (within the 16-lane rows) and the two swaps above
// (across the rows) -- VALU only.  The __shfl_xor butterfly is six ds_bpermute round trips per reduction.
template <int N> __device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 | N, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_row_ror<8>(v);
  v += dpp_row_ror<4>(v);
  v += dpp_row_ror<2>(v);
  v += dpp_row_ror<1>(v);
  return wave_groups_sum(v);
}

// wave64 butterfly helpers
__device__ __forceinline__ float wave_xor_max(float v, int mask) { return fmaxf(v, __shfl_xor(v, mask)); }
__device__ __forceinline__ float wave_xor_add(float v, int mask) { return v + __shfl_xor(v, mask); }

// XCD-aware block remap: consecutive logical ids land on one XCD (blocks b, b+8, .. share an XCD's L2).
// Bijective for any grid size (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
