// Shared device code of the fused per-token kernels (layer_fused.hip: forward; layer_fused_bwd.hip: backward): the LDS-DMA
// weight ring, the register-chained GEMM stage, operand packing, LDS-staged row I/O, lane-local LayerNorm.
#pragma once
#include "wmz_common.h"
#include <stdlib.h>

// Compiler + scheduler fence: neither IR passes (memory clobber) nor the machine scheduler (sched_barrier) may move
// loads across it.  Used to cap how many operand loads are in flight: the register file is the scarce resource here.
#define WMZ_FENCE()                        \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

namespace {

constexpr int TW = 32;          // tokens per wave
#ifndef WMZ_FUSED_FW
#define WMZ_FUSED_FW 8
#endif
constexpr int FW = WMZ_FUSED_FW; // waves per workgroup.  8: one workgroup of 256 tokens per CU (two waves per SIMD, one's epilogue
                                // overlaps the other's MFMAs).  4: two independent workgroups of 128 tokens per CU, out of phase
constexpr int NTHR = FW * 64;
constexpr int HALF = 16384;     // granule of the weight stream: every GEMM stage is a whole number of these
constexpr int PIECES = HALF / 1024;   // MFMA A operands per granule
#ifndef WMZ_FUSED_HPS
#define WMZ_FUSED_HPS 2
#endif
constexpr int HPS = WMZ_FUSED_HPS;    // granules per slab = per workgroup barrier (1: 16 KB slabs, ring of 4; 2: 32 KB, ring of 2)
constexpr int SLAB = HPS * HALF;      // bytes per weight slab (one LDS-DMA burst, one barrier)
#ifndef WMZ_FUSED_RING
#define WMZ_FUSED_RING (WMZ_FUSED_HPS == 1 ? 4 : 2)
#endif
constexpr int RING = WMZ_FUSED_RING;  // LDS ring slots; RING-1 slabs of the weight stream stay in flight
constexpr int WPP = SLAB / 1024 / FW; // LDS-DMA pieces per wave per slab
constexpr int VECB = 8192;      // the layer's bias / LayerNorm vectors (2048 fp32), staged once per workgroup
constexpr int MC = 32;          // feed-forward hidden chunk: W1 rows / W2 columns streamed MC at a time

// ---- weight stream: RING-slot LDS ring filled by LDS-DMA (global_load_lds), RING-1 slabs in flight.
struct WStream {
  int dbg;
  const char* src;     // global address of the next slab to ISSUE (this lane's 16 bytes of piece 0 of its wave)
  char* ring;          // LDS ring base + this wave's eighth of a slab
  int issue_slot;      // ring slot the next issued slab goes to
  int cur;             // ring slot of the slab being multiplied
  int half;            // granule of that slab the next stage starts at
  int wave;
  int probe;           // timing probe slot base for the next slab (0 = off)
  long long* ts;
  // vmcnt bookkeeping: tot = every other VMEM op (row loads / stores) this wave has issued so far; t1..t3 = tot at the
  // moment the last three slabs were issued, oldest first
  int tot, t1, t2, t3;
  int all;             // every VMEM op this wave has issued (ring pieces included): vm_wait_since() waits by sequence number
};

__device__ __forceinline__ void ws_issue(WStream& ws) {
  ws.t1 = ws.t2; ws.t2 = ws.t3; ws.t3 = ws.tot;
  ws.all += WPP;
  if (ws.dbg & 2) return;
  char* dst = ws.ring + ws.issue_slot * SLAB;
#pragma unroll
  for (int i = 0; i < WPP; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws.src + i * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  ws.src += SLAB;
  ws.issue_slot = ws.issue_slot == RING - 1 ? 0 : ws.issue_slot + 1;
}

// Before multiplying a slab: this wave's eighth of it has landed, then one barrier: every piece landed, and every wave is
// done with the previous slab, whose slot the caller refills (ws_issue) once its first fragment reads are out (spreading
// the requests over the stage instead measured slower).  vmcnt counts loads, stores and LDS-DMA together in issue
// order: the oldest slab in flight has landed <=> at most [the WPP*(RING-2) pieces of the younger slabs + every other op
// issued after its pieces (tot - its t)] is outstanding.
__device__ __forceinline__ void ws_wait(WStream& ws) {
  constexpr int YB = WPP * (RING - 2);
  static_assert(YB == 4 || YB == 0, "literals below");
  const int e = ws.tot - (RING == 4 ? ws.t1 : (RING == 3 ? ws.t2 : ws.t3));
  if (!(ws.dbg & 2)) {
#define WMZ_VMC(n) case n: if (YB == 4) asm volatile("s_waitcnt vmcnt(" #n " + 4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    switch (e) {
      WMZ_VMC(0) WMZ_VMC(1) WMZ_VMC(2) WMZ_VMC(3) WMZ_VMC(4) WMZ_VMC(5) WMZ_VMC(6) WMZ_VMC(7) WMZ_VMC(8) WMZ_VMC(9)
      WMZ_VMC(10) WMZ_VMC(11) WMZ_VMC(12) WMZ_VMC(13) WMZ_VMC(14) WMZ_VMC(15) WMZ_VMC(16) WMZ_VMC(17) WMZ_VMC(18)
      WMZ_VMC(19) WMZ_VMC(20) WMZ_VMC(21) WMZ_VMC(22) WMZ_VMC(23) WMZ_VMC(24) WMZ_VMC(25) WMZ_VMC(26) WMZ_VMC(27)
      WMZ_VMC(28) WMZ_VMC(29) WMZ_VMC(30) WMZ_VMC(31) WMZ_VMC(32)
      default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;      // more than 32: waits for the surplus (safe)
    }
#undef WMZ_VMC
  }
  if (ws.probe && ws.ts) ws.ts[ws.probe] = __builtin_readcyclecounter();
  if (!(ws.dbg & 4)) __builtin_amdgcn_s_barrier();     // (dbg 4: timing experiment without the per-slab rendezvous; garbage results)
  if (ws.probe && ws.ts) ws.ts[ws.probe + 1] = __builtin_readcyclecounter();
}
__device__ __forceinline__ void ws_release(WStream& ws) { ws.cur = ws.cur == RING - 1 ? 0 : ws.cur + 1; }
__device__ __forceinline__ void ws_extra(WStream& ws, int n) { ws.tot += n; ws.all += n; }
// s_waitcnt vmcnt(n) for a run-time n (vmcnt completes in issue order: "at most n outstanding" = everything but the n
// youngest VMEM ops of this wave has landed).  More than 48: waits for the surplus (safe).
__device__ __forceinline__ void vm_wait(int n) {
#define WMZ_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    WMZ_VMW(0) WMZ_VMW(1) WMZ_VMW(2) WMZ_VMW(3) WMZ_VMW(4) WMZ_VMW(5) WMZ_VMW(6) WMZ_VMW(7) WMZ_VMW(8) WMZ_VMW(9) WMZ_VMW(10)
    WMZ_VMW(11) WMZ_VMW(12) WMZ_VMW(13) WMZ_VMW(14) WMZ_VMW(15) WMZ_VMW(16) WMZ_VMW(17) WMZ_VMW(18) WMZ_VMW(19) WMZ_VMW(20)
    WMZ_VMW(21) WMZ_VMW(22) WMZ_VMW(23) WMZ_VMW(24) WMZ_VMW(25) WMZ_VMW(26) WMZ_VMW(27) WMZ_VMW(28) WMZ_VMW(29) WMZ_VMW(30)
    WMZ_VMW(31) WMZ_VMW(32) WMZ_VMW(33) WMZ_VMW(34) WMZ_VMW(35) WMZ_VMW(36) WMZ_VMW(37) WMZ_VMW(38) WMZ_VMW(39) WMZ_VMW(40)
    WMZ_VMW(41) WMZ_VMW(42) WMZ_VMW(43) WMZ_VMW(44) WMZ_VMW(45) WMZ_VMW(46) WMZ_VMW(47)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef WMZ_VMW
}
// every VMEM op up to sequence number `mark` (= ws.all right after it was issued and accounted) has landed
__device__ __forceinline__ void vm_wait_since(const WStream& ws, int mark) { vm_wait(ws.all - mark); }

// Side work of a GEMM stage: called once per group of AG MFMAs; kValuPerMfma tells the stage how many of its VALU
// instructions the scheduler should place behind EACH MFMA (measured on MI355X at two waves per SIMD: up to ~2 VALU per
// MFMA and wave hide completely, 4-8 cost about half their time, a block of VALU behind a block of MFMAs hides nothing).
struct NoSide { static constexpr int kValuPerMfma = 0; __device__ __forceinline__ void operator()(int) const {} };
template <int VPM, typename F> struct SideWork {
  static constexpr int kValuPerMfma = VPM;
  F f;
  __device__ __forceinline__ void operator()(int g) const { f(g); }
};
template <int VPM, typename F> __device__ __forceinline__ SideWork<VPM, F> side_work(F f) { return SideWork<VPM, F>{f}; }

// acc[NB blocks of 32 features x 32 tokens] += W . act^T over KS 16-deep k-steps; the stream holds the pieces in
// (k-step, block) order, so a k-step's operand is used by NB independent accumulators.  bget(s) yields the B operand
// of k-step s (a register array, or an LDS read issued one k-step ahead).
template <int NB, int KS, typename BGet, typename Side = NoSide>
__device__ __forceinline__ void gemm_stage_b(f32x16 (&acc)[NB], BGet bget, const char* ring0, WStream& ws, int lane,
                                             Side side = Side()) {
  constexpr int NP = NB * KS;
  static_assert(NP % PIECES == 0, "a stage is a whole number of slabs");
  constexpr int AG = 4, GPS = PIECES / AG;                 // fragments per group, groups per slab
  Frag8<bf16_t> bcur, bnext = bget(0);
  bcur = bnext;
#pragma unroll
  for (int sl = 0; sl < NP / PIECES; ++sl) {
    if (ws.half == 0) ws_wait(ws);
    const char* slab = ring0 + ws.cur * SLAB + ws.half * HALF + lane * 16;
    // A operands: groups of AG, the next group's ds_reads in flight under this group's MFMAs (8 fragments live, no more:
    // the scheduler is fenced so that it cannot hoist the whole slab's reads into registers the chain needs)
    Frag8<bf16_t> af[2][AG];
    if (!(ws.dbg & 1)) {
#pragma unroll
      for (int j = 0; j < AG; ++j) af[0][j].v = *reinterpret_cast<const s16x8*>(slab + j * 1024);
    }
    if (ws.half == 0) ws_issue(ws);                        // refill the retired slot while the first fragments arrive
    if (ws.probe && ws.ts) ws.ts[ws.probe + 2] = __builtin_readcyclecounter();
    if (!(ws.dbg & 1)) {
#pragma unroll
      for (int gq = 0; gq < GPS; ++gq) {
        if (gq == 1 && ws.probe && ws.ts) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ws.ts[ws.probe + 3] = __builtin_readcyclecounter(); }
        if (gq + 1 < GPS) {
#pragma unroll
          for (int j = 0; j < AG; ++j)
            af[(gq + 1) & 1][j].v = *reinterpret_cast<const s16x8*>(slab + ((gq + 1) * AG + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < AG; ++j) {
          const int idx = sl * PIECES + gq * AG + j;
          if (idx % NB == 0) {
            bcur = bnext;
            if (idx / NB + 1 < KS) bnext = bget(idx / NB + 1);
          }
          mma32(acc[idx % NB], af[gq & 1][j], bcur);
        }
        side(sl * GPS + gq);                               // VALU work that rides under this group's MFMAs,
        // Order inside the group: the NEXT group's fragment reads go out first (left to itself the scheduler sinks them
        // behind the MFMAs and the wave then waits a full LDS round trip per group), then the MFMAs, each followed by
        // its share of the side work.
        if (gq + 1 < GPS) __builtin_amdgcn_sched_group_barrier(0x100, AG, 0);
        if constexpr (Side::kValuPerMfma > 0) {
#pragma unroll
          for (int j = 0; j < AG; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, Side::kValuPerMfma, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, AG, 0);
        }
        WMZ_FENCE();
      }
    }
    if (ws.half == HPS - 1) { ws_release(ws); ws.half = 0; }
    else ++ws.half;
  }
}
template <int NB, int KS, typename Side = NoSide>
__device__ __forceinline__ void gemm_stage(f32x16 (&acc)[NB], const Frag8<bf16_t> (&bop)[KS], const char* ring0, WStream& ws,
                                           int lane, Side side = Side()) {
  gemm_stage_b<NB, KS>(acc, [&](int s) { return bop[s]; }, ring0, ws, lane, side);
}

__device__ __forceinline__ f32x16 lds_vec16(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  const f32x4 c = *reinterpret_cast<const f32x4*>(p + 8), d = *reinterpret_cast<const f32x4*>(p + 12);
  f32x16 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; v[8 + i] = c[i]; v[12 + i] = d[i]; }
  return v;
}
template <int NB>
__device__ __forceinline__ void add_vec(f32x16 (&acc)[NB], const float* vec /* + h*16*NB */) {
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] += lds_vec16(vec + 16 * b);
}
// accumulators that start at the bias: the add rides in the MFMA chain
template <int NB>
__device__ __forceinline__ void init_vec(f32x16 (&acc)[NB], const float* vec) {
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = lds_vec16(vec + 16 * b);
}

// The empty asm pins the packed operand HERE: without it the compiler sinks the conversion arithmetic down to the MFMA
// that consumes it and keeps the fp32 sources (and every gamma / beta / bias fetched for them) alive until then.
__device__ __forceinline__ void pack8(Frag8<bf16_t>& f, const float (&y)[8]) {
#ifdef WMZ_OP16_F16
  // (half operands: written as four explicit two-wide conversions -- element by element hipcc pairs most but not all of them
  //  into v_cvt_pk_f16_f32 and assembles the rest with v_cvt_f16_f32 + v_alignbit + moves)
  const s16x4 lo = cvt_pk4_bf16(y[0], y[1], y[2], y[3]), hi = cvt_pk4_bf16(y[4], y[5], y[6], y[7]);
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#else
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)f32_to_bf16_bits(y[j]);
#endif
  asm volatile("" : "+v"(v));
  f.v = v;
}

// operand k-step s of a K-feature activation = features h*K/2 + 8*s .. +7 of the lane's token: the lane's half row.
// Unconditional loads (rows past ntok are clamped by the caller to a valid row; their results are never stored): a
// predicated load makes hipcc branch around it and wait for each one in turn -- one L2 round trip per 16 bytes.
template <int KS>
__device__ __forceinline__ void load_bop(Frag8<bf16_t> (&bop)[KS], const bf16_t* half_row) {
#pragma unroll
  for (int s = 0; s < KS; ++s) bop[s].v = *reinterpret_cast<const s16x8*>(half_row + 8 * s);
}

// A load the compiler does not know about: no automatic s_waitcnt (which, with LDS-DMA in flight, is always vmcnt(0)).
// Its result may only be used behind wait_untracked().
__device__ __forceinline__ s16x8 gload_untracked(const bf16_t* p) {
  s16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// vmcnt is in issue order: once at most the ring's WPP*(RING-1) youngest pieces are outstanding, every older load is done.
template <int KS>
__device__ __forceinline__ void wait_untracked(Frag8<bf16_t> (&b)[KS]) {
  if (WPP * (RING - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
  for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(b[s].v));       // uses stay behind the wait
}
// the 8 KB vector block by LDS-DMA: wave w moves KB w
__device__ __forceinline__ void vec_dma(float* vecs, const float* src, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 8 / FW; ++i) {
    const int kb = wave * (8 / FW) + i;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kb * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(vecs) + kb * 1024), 16, 0, 0);
  }
}
template <int NB>
__device__ __forceinline__ void add_bop(f32x16 (&acc)[NB], const Frag8<bf16_t> (&bop)[2 * NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = op16_add_to(acc[b][i], (unsigned short)bop[2 * b + (i >> 3)].v[i & 7]);
}

template <int KS>
__device__ __forceinline__ void load_bop_tiled(Frag8<bf16_t> (&bop)[KS], const bf16_t* tile, int lane) {
#pragma unroll
  for (int s = 0; s < KS; ++s) bop[s].v = *reinterpret_cast<const s16x8*>(tile + (s * 64 + lane) * 8);   // (2s+h)*32+t = 64s+lane
}
template <int KS>
__device__ __forceinline__ void store_bop_tiled(bf16_t* tile, const Frag8<bf16_t> (&bop)[KS], int lane) {
#pragma unroll
  for (int s = 0; s < KS; ++s) *reinterpret_cast<s16x8*>(tile + (s * 64 + lane) * 8) = bop[s].v;
}

// Row stores through the wave's private 8 KB LDS buffer: a lane owns HALF A ROW of its token, so direct 16-byte stores
// would scatter 64 pieces per instruction over 32 rows (measured: ~570 cycles per store instruction, and the LDS-DMA
// weight stream queues behind them).  Instead: the lanes write their pieces into a [32 rows x 256 B] image (16-byte
// chunk c of row r at chunk c ^ (r & 15): conflict-free both ways), then the wave copies the image out 1 KB per
// instruction, whole 256-byte runs per row.
// LDS-DMA a [32 tokens x 128 features] tile (rows of ROWF features, 256 B of each from column col0) into the wave's image,
// same chunk swizzle as the store path (applied on the source address: the DMA writes lane-linear).
__device__ __forceinline__ void stage_dma128(char* stg, const bf16_t* src, int rowf, long tok0, int ntok, int lane) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = p * 4 + (lane >> 4), pc = lane & 15;
    const long row = tok0 + r < ntok ? tok0 + r : ntok - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + row * rowf + ((pc ^ (r & 15)) << 3)),
                                     (__attribute__((address_space(3))) void*)(stg + p * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ Frag8<bf16_t> stage_get(const char* stg, int t, int c) {
  Frag8<bf16_t> f;
  f.v = *reinterpret_cast<const s16x8*>(stg + t * 256 + ((c ^ (t & 15)) << 4));
  return f;
}
__device__ __forceinline__ void stage_put(char* stg, const s16x8& v, int t, int c) {
  *reinterpret_cast<s16x8*>(stg + t * 256 + ((c ^ (t & 15)) << 4)) = v;
}
// copy the 8 KB image out: row r of the image -> dst + (tok0 + r) * ROWF + col0, 128 features (256 B) per row
template <int ROWF>
__device__ __forceinline__ void stage_flush(const char* stg, bf16_t* dst, long tok0, int ntok, int col0, int lane) {
  asm volatile("" : "+s"(tok0), "+v"(lane));   // compute the store addresses HERE (hoisted / shared with the prologue's
                                               // index math they only get spilled)
  // all eight LDS reads first, unconditionally, then the predicated stores: with the read inside the predicate hipcc
  // emits branch / read / wait / store per KB, eight LDS round trips in a row
  s16x8 v[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const s16x8*>(stg + p * 1024 + lane * 16);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = p * 4 + (lane >> 4), pc = lane & 15;
    const int c = pc ^ (r & 15);
    if (tok0 + r < ntok) *reinterpret_cast<s16x8*>(dst + (tok0 + r) * ROWF + col0 + c * 8) = v[p];
  }
}
// operand fragments parked in the wave's LDS image, lane-linear (each lane reads back what it wrote)
__device__ __forceinline__ void frag_park(char* stg, int slot, const Frag8<bf16_t>& f, int lane) {
  *reinterpret_cast<s16x8*>(stg + slot * 1024 + lane * 16) = f.v;
}
__device__ __forceinline__ Frag8<bf16_t> frag_unpark(const char* stg, int slot, int lane) {
  Frag8<bf16_t> f;
  f.v = *reinterpret_cast<const s16x8*>(stg + slot * 1024 + lane * 16);
  return f;
}

// an I-feature tile (q, k, v: 128 features, every lane holds 8 chunks of its row): one pass
__device__ __forceinline__ void store_tile128(char* stg, bf16_t* dst, int rowf, long tok0, int ntok, int col0,
                                              const Frag8<bf16_t> (&b)[8], int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 8; ++s) stage_put(stg, b[s].v, t, h * 8 + s);
  if (rowf == 128) stage_flush<128>(stg, dst, tok0, ntok, col0, lane);
  else stage_flush<256>(stg, dst, tok0, ntok, col0, lane);
}
// the D-feature stream (256 features, a lane holds 16 chunks = its whole 256-byte half row): one pass per lane half
__device__ __forceinline__ void store_tile256(char* stg, bf16_t* dst, long tok0, int ntok, const Frag8<bf16_t> (&b)[16],
                                              int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    if (h == hh) {
#pragma unroll
      for (int s = 0; s < 16; ++s) stage_put(stg, b[s].v, t, s);
    }
    stage_flush<256>(stg, dst, tok0, ntok, hh * 128, lane);
  }
}

template <int NB>
__device__ __forceinline__ void acc_from_bop(f32x16 (&acc)[NB], const Frag8<bf16_t> (&bop)[2 * NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = bf16_bits_to_f32((unsigned short)bop[2 * b + (i >> 3)].v[i & 7]);
#ifdef WMZ_OP16_F16
  // (half operands: without this pin hipcc folds the conversion into the LayerNorm's multiply-add as v_fma_mix_f32 reading the
  //  PACKED source -- and keeps the packed row alive beside its fp32 copy: +30 registers, a spill in the head kernels)
#pragma unroll
  for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(acc[b]));
#endif
}
template <int NB>
__device__ __forceinline__ void bop_from_acc(Frag8<bf16_t> (&bop)[2 * NB], const f32x16 (&acc)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = acc[b][8 * m + j];
      pack8(bop[2 * b + m], y);
    }
}

// LayerNorm statistics of the lane pair's token (features split over the two lane halves): rstd and -mean*rstd
template <int NB>
__device__ __forceinline__ void ln_stats(const f32x16 (&acc)[NB], float eps, float& rstd, float& mr) {
  constexpr int NF = NB * 32;
  f32x16 sv = acc[0];
#pragma unroll
  for (int b = 1; b < NB; ++b) sv += acc[b];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += sv[i];
  s = wave_halves_sum(s);
  const float mean = s / (float)NF;
  f32x16 qv = (f32x16)(0.f);
#pragma unroll
  for (int b = 0; b < NB; ++b) { const f32x16 d = acc[b] - mean; qv += d * d; }
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) q += qv[i];
  q = wave_halves_sum(q);
  rstd = rsqrtf(q / (float)NF + eps);
  mr = -mean * rstd;
}
// one block: x*rstd - mean*rstd -> two bf16 operands (the LayerNorm affine lives in the weights that consume it).
// Not (x - mean)*rstd: the variance pass used x - mean, and reusing it would keep a second copy of the row alive.
__device__ __forceinline__ void ln_block(Frag8<bf16_t>& lo, Frag8<bf16_t>& hi, const f32x16& x, float rstd, float mr) {
  const f32x16 y = x * rstd + mr;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = y[j]; b[j] = y[8 + j]; }
  pack8(lo, a);
  pack8(hi, b);
}
// training: mean / rstd of the lane pair's token leave through the lower lane half (two wave-level stores)
__device__ __forceinline__ void put_stats(float* st, long tok, int ntok, float rstd, float mr, int lane) {
  if (lane < 32 && tok < ntok) {
    st[tok] = -mr / rstd;
    st[(long)ntok + tok] = rstd;
  }
}
template <int NB>
__device__ __forceinline__ void ln_to_bop(Frag8<bf16_t> (&bop)[2 * NB], const f32x16 (&acc)[NB], float eps, float* st = nullptr,
                                          long tok = 0, int ntok = 0, int lane = 0) {
  float rstd, mr;
  ln_stats<NB>(acc, eps, rstd, mr);
  if (st != nullptr) put_stats(st, tok, ntok, rstd, mr, lane);
#pragma unroll
  for (int b = 0; b < NB; ++b) ln_block(bop[2 * b], bop[2 * b + 1], acc[b], rstd, mr);
}
// x2 (fp32) -> normalised operand and x2 itself as bf16, block by block (each block of xr dies as its operands appear)
template <int NB>
__device__ __forceinline__ void ln_and_pack(Frag8<bf16_t> (&lnb)[2 * NB], Frag8<bf16_t> (&xb)[2 * NB], const f32x16 (&acc)[NB],
                                            float eps, float* st = nullptr, long tok = 0, int ntok = 0, int lane = 0) {
  float rstd, mr;
  ln_stats<NB>(acc, eps, rstd, mr);
  if (st != nullptr) put_stats(st, tok, ntok, rstd, mr, lane);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    ln_block(lnb[2 * b], lnb[2 * b + 1], acc[b], rstd, mr);
    float a[8], c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = acc[b][j]; c[j] = acc[b][8 + j]; }
    pack8(xb[2 * b], a);
    pack8(xb[2 * b + 1], c);
  }
}

template <int N> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) acc[b] = (f32x16)(0.f);
}
}  // namespace
