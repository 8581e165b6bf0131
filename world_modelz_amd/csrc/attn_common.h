// Shared geometry helpers of the local-3D-attention kernels (forward and backward).
//
// Token grid [B,S,H,W]; one (b, s) plane is flattened to p = h*W + w and cut into TILES of 16 consecutive
// positions (the last tile of a plane may be ragged).  A workgroup owns NWAVES*QPW consecutive query tiles
// of one plane and one head; key/value tiles of plane s+ds are staged through LDS KC tiles at a time.
// Window membership of (query, key) is decided from packed (h<<16 | w) coordinates with packed-u16 math.
#pragma once
#include "wmz_common.h"

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

struct AttnGeom {
  int B, S, H, W, heads, dh;
  int eS, eH, eW;
  long ldq, ldk, ldv, ldo;
  int HW;        // H*W
  int tiles;     // ceil(HW/16)
  int qgroups;   // workgroups per (b, head, s) plane
  float scale;   // dh^-0.5
  int qs0, Sq;   // query planes [qs0, qs0+Sq) only; out / lse are compact [B, Sq, H, W, ..] (forward; full grid: 0, S)
  int dbg;       // ablation switches for timing experiments: 1 = skip the per-tile compute, 2 = skip the K/V staging
  int variant;   // development A/B switch between kernel instantiations (wmz_debug_attn_knobs); 0 = product default
  int w8;        // row kernels: the plane is 8 wide and handed over as H / 2 tile rows of 16 (attn_fwd_row16.hip)
};

struct TileInfo { int hlo, hhi, wlo, whi; };

#define WMZ_ATTN_INVALID_COORD 0x7FFF7FFF

// coords[p] = (h<<16 | w) for p < HW, INVALID beyond; tinfo[t] = row/col bounding box of tile t.
__device__ __forceinline__ void attn_build_tables(int* coords, TileInfo* tinfo, int HW, int W, int tiles,
                                                  int tid, int nthreads) {
  for (int p = tid; p < tiles * 16; p += nthreads) {
    int c = WMZ_ATTN_INVALID_COORD;
    if (p < HW) { const int h = p / W; c = (h << 16) | (p - h * W); }
    coords[p] = c;
  }
  for (int t = tid; t < tiles; t += nthreads) {
    const int p0 = t * 16, p1 = min(p0 + 15, HW - 1);
    TileInfo ti;
    ti.hlo = p0 / W; ti.hhi = p1 / W;
    if (ti.hlo == ti.hhi) { ti.wlo = p0 - ti.hlo * W; ti.whi = p1 - ti.hhi * W; }
    else { ti.wlo = 0; ti.whi = W - 1; }
    tinfo[t] = ti;
  }
}

// Packed window test.  cmin = ((hq-eH)&0xFFFF)<<16 | ((wq-eW)&0xFFFF), lim = (2eH<<16)|2eW.
// Returns t = ck - cmin per 16-bit field; in-window iff both fields <= lim's.
__device__ __forceinline__ unsigned win_delta(int ck, unsigned cmin) {
  const u16x2_t t = __builtin_bit_cast(u16x2_t, (unsigned)ck) - __builtin_bit_cast(u16x2_t, cmin);
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ bool win_inside(unsigned t, unsigned lim) {
  const u16x2_t m = __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, t), __builtin_bit_cast(u16x2_t, lim));
  return __builtin_bit_cast(unsigned, m) == t;
}
__device__ __forceinline__ unsigned win_cmin(int cq, int eH, int eW) {
  const int hq = cq >> 16, wq = cq & 0xFFFF;
  return (((unsigned)(hq - eH) & 0xFFFFu) << 16) | ((unsigned)(wq - eW) & 0xFFFFu);
}

// LDS tile images.  Row r of a tile block holds ROWB bytes; 16-byte chunk c of row r lives at
//   r*ROWB + ((c*16) ^ swz(r)).
// kswz makes the row-fragment read (ds_read_b128: lanes = 16 rows x 4 chunks) conflict-free,
// vswz makes the transposed read (ds_read_b64_tr_b16: 8 rows x 32 B per half-wave) conflict-free.
template <int ROWB> __device__ __forceinline__ int kswz(int r) {
  if constexpr (ROWB >= 256) return (r & 15) << 4;
  else return ((r / (256 / ROWB)) << 4) & (ROWB - 16);
}
template <int ROWB> __device__ __forceinline__ int vswz(int r) {
  if constexpr (ROWB >= 256) return (r & 7) << 5;
  else if constexpr (ROWB >= 64) return ((r / (256 / ROWB)) << 5) & (ROWB - 32);
  else return 0;
}

// Stage `ntiles` (<= KC) 16-row tiles of a [HW, ld] plane (rows c0*16 ...) into an LDS image; rows >= HW and
// columns >= dh are zero-filled.  VS selects the V swizzle.  Fixed trip count: all global loads of the slab are
// issued before the first LDS store, so their latencies overlap.
template <typename T, int DH, bool VS, int KC, int NTHREADS>
__device__ __forceinline__ void attn_stage_load(i32x4 (&regs)[KC * 16 * (DH * (int)sizeof(T) / 16) / NTHREADS],
                                                const T* plane, long ld, int c0, int ntiles, int HW, int dh, int tid) {
  constexpr int ROWB = DH * (int)sizeof(T);
  constexpr int CPR = ROWB / 16;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int ITERS = KC * 16 * CPR / NTHREADS;
  static_assert(KC * 16 * CPR % NTHREADS == 0, "slab must divide evenly over the workgroup");
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * NTHREADS;
    const int r = idx / CPR, c = idx - r * CPR;
    const int p = c0 * 16 + r;
    regs[it] = (i32x4)(0);
    if (r < ntiles * 16 && p < HW && c * EPC < dh) regs[it] = *reinterpret_cast<const i32x4*>(plane + (long)p * ld + c * EPC);
  }
}
template <typename T, int DH, bool VS, int KC, int NTHREADS>
__device__ __forceinline__ void attn_stage_store(char* dst, const i32x4 (&regs)[KC * 16 * (DH * (int)sizeof(T) / 16) / NTHREADS],
                                                 int ntiles, int tid) {
  constexpr int ROWB = DH * (int)sizeof(T);
  constexpr int CPR = ROWB / 16;
  constexpr int ITERS = KC * 16 * CPR / NTHREADS;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * NTHREADS;
    const int r = idx / CPR, c = idx - r * CPR;
    if (r < ntiles * 16) {
      const int sw = VS ? vswz<ROWB>(r) : kswz<ROWB>(r);
      *reinterpret_cast<i32x4*>(dst + r * ROWB + ((c << 4) ^ sw)) = regs[it];
    }
  }
}

// LDS-DMA staging of `KC` tiles (rows c0*16 ...) of a [HW, ld] plane into a swizzled LDS image.  global_load_lds writes
// lane-linear (1 KB per wave instruction), so the swizzle is applied on the SOURCE side: the lane that lands on physical
// chunk pc of row r fetches logical chunk pc ^ (swz(r) >> 4).  Rows past the staged range / the plane and columns past dh
// are redirected to valid addresses (finite data; the window mask or Q's zero padding nullifies them).  Each of the NW
// waves issues IMG/1024/NW pieces; completion is the issuing wave's vmcnt.
template <typename T, int DH, bool VS, int KC, int NW>
__device__ __forceinline__ void attn_stage_dma(char* dst, const T* plane, long ld, int c0, int ntiles, int HW, int dh,
                                               int wave, int lane) {
  constexpr int ROWB = DH * (int)sizeof(T);
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int PIECES = KC * 16 * ROWB / 1024;
  const int last_row = min(ntiles * 16, HW - c0 * 16) - 1;
  const int cmax = dh / EPC - 1;
#pragma unroll
  for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
    const int piece = wave + NW * i;
    if (PIECES % NW != 0 && piece >= PIECES) break;      // wave-uniform
    const int off = piece * 1024 + lane * 16;
    const int r = off / ROWB, pc = (off - r * ROWB) >> 4;
    const int sw = VS ? vswz<ROWB>(r) : kswz<ROWB>(r);
    const int c = min(pc ^ (sw >> 4), cmax);
    const int rr = min(r, last_row);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(plane + (long)(c0 * 16 + rr) * ld + c * EPC),
                                     (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
  }
}

// Row fragment (8 consecutive elements starting at element e0, e0 % 8 == 0) of image row r.
template <typename T, int DH, bool VS>
__device__ __forceinline__ void lds_row_frag(Frag8<T>& f, const char* img, int r, int e0) {
  constexpr int ROWB = DH * (int)sizeof(T);
  const int sw = VS ? vswz<ROWB>(r) : kswz<ROWB>(r);
  const char* row = img + r * ROWB;
  if constexpr (sizeof(T) == 2) {
    f.v = *reinterpret_cast<const s16x8*>(row + ((e0 * 2) ^ sw));
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + ((e0 * 4) ^ sw));
    const f32x4 b = *reinterpret_cast<const f32x4*>(row + ((e0 * 4 + 16) ^ sw));
#pragma unroll
    for (int i = 0; i < 4; ++i) { f.v[i] = a[i]; f.v[4 + i] = b[i]; }
  }
}

// Transposed fragment: element j of the result = img[row rbase[j>>2] + 4g + (j&3)][col], i.e. 8 rows of ONE
// column.  bf16 uses the hardware transpose read (all 64 lanes must execute it: EXEC all ones).
//   ra / rb: first row of the two 16-row tiles, g = lane>>4, li = lane&15, col0 = first column of the
//   16-column block; the lane's column is col0 + li.
template <typename T, int DH, bool VS>
__device__ __forceinline__ void lds_col_frag(Frag8<T>& f, const char* img, int ra, int rb, int g, int li, int col0) {
  constexpr int ROWB = DH * (int)sizeof(T);
  if constexpr (sizeof(T) == 2) {
    // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the 4x16 block
    const int q = li >> 2, p = li & 3;
    const int r0 = ra + 4 * g + q, r1 = rb + 4 * g + q;
    const int cb = (col0 + 4 * p) * 2;
    const int sw0 = VS ? vswz<ROWB>(r0) : kswz<ROWB>(r0);
    const int sw1 = VS ? vswz<ROWB>(r1) : kswz<ROWB>(r1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r0 * ROWB + (cb ^ sw0)));
    const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r1 * ROWB + (cb ^ sw1)));
    f.v = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ((j >> 2) ? rb : ra) + 4 * g + (j & 3);
      const int sw = VS ? vswz<ROWB>(r) : kswz<ROWB>(r);
      f.v[j] = *reinterpret_cast<const float*>(img + r * ROWB + (((col0 + li) * 4) ^ sw));
    }
  }
}

template <typename T> __device__ __forceinline__ void frag_from_f32(Frag8<T>& f, const float (&p)[8]);
template <> __device__ __forceinline__ void frag_from_f32<float>(Frag8<float>& f, const float (&p)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = p[i];
}
template <> __device__ __forceinline__ void frag_from_f32<bf16_t>(Frag8<bf16_t>& f, const float (&p)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = (short)f32_to_bf16_bits(p[i]);
}
