// Micro-benchmark: how fast can a workgroup stage L2-resident data into LDS on gfx950 -- by LDS-DMA (global_load_lds, 16 B per
// lane) or through registers (global_load_dwordx4 + ds_write_b128)?  One workgroup of NW waves per CU, every workgroup sweeps
// its own REGION bytes (L2-resident after the first sweep) SWEEPS times in slabs of SLAB bytes into a double-buffered LDS image,
// with the attention kernel's per-slab rendezvous (vmcnt(0) + s_barrier) or without it.
//
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_bw.hip -o lds_dma_bw && ./lds_dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SLAB = 73728;     // bytes per slab (the attention kernel's K + V slab)
constexpr int REGION = 16 * SLAB;

template <int NW, int MODE, bool BARRIER>   // MODE 0: LDS-DMA, 1: registers
__global__ __launch_bounds__(NW * 64, 1) void stage_kernel(const char* __restrict__ src, int sweeps, int* __restrict__ sink, int nregions) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * SLAB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // nregions = grid: private regions (302 MB: HBM-bound); nregions = 8: the workgroups of an XCD (block index mod 8) share one
  // L2-resident region, as the attention kernel's neighbouring planes share key planes; each starts at its own slab
  const char* base = src + (size_t)(blockIdx.x % nregions) * REGION;
  const int phase = (blockIdx.x / nregions) % (REGION / SLAB);
  constexpr int PIECES = SLAB / 1024;                       // 1 KB pieces per slab
  constexpr int NP = (PIECES + NW - 1) / NW;
  int acc = 0;
  const int nslab = sweeps * (REGION / SLAB);
  auto issue = [&](int j) {
    const char* s = base + (size_t)((j + phase) % (REGION / SLAB)) * SLAB;
    char* d = smem + (j & 1) * SLAB;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = wave + NW * i;
      if (piece < PIECES) {
        if constexpr (MODE == 0) {
          __builtin_amdgcn_global_load_lds((gptr_t)(s + piece * 1024 + lane * 16), (lptr_t)(d + piece * 1024), 16, 0, 0);
        } else {
          const i32x4 v = *reinterpret_cast<const i32x4*>(s + piece * 1024 + lane * 16);
          *reinterpret_cast<i32x4*>(d + piece * 1024 + lane * 16) = v;
        }
      }
    }
  };
  issue(0);
  for (int j = 0; j < nslab; ++j) {
    if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
    if (j + 1 < nslab) issue(j + 1);
    acc += *reinterpret_cast<const int*>(smem + (j & 1) * SLAB + tid * 4);     // touch the slab
  }
  if (acc == 0x12345678) sink[0] = acc;
}

template <int NW, int MODE, bool BARRIER>
static void run(const char* name, const char* src, int* sink, int ncu, int nregions) {
  const int sweeps = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((stage_kernel<NW, MODE, BARRIER>), dim3(ncu), dim3(NW * 64), 0, 0, src, sweeps, sink, nregions);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((stage_kernel<NW, MODE, BARRIER>), dim3(ncu), dim3(NW * 64), 0, 0, src, sweeps, sink, nregions);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 10.0 * ncu * sweeps * (double)REGION;
  const double per_slab_us = ms * 1e3 / (10.0 * sweeps * (REGION / SLAB));
  printf("%-44s %7.2f TB/s aggregate, %6.1f GB/s per CU, %5.2f us per %d KB slab\n", name, bytes / ms / 1e9, bytes / ms / 1e6 / ncu,
         per_slab_us, SLAB / 1024);
}

int main() {
  int ncu = 256;
  char* src;
  int* sink;
  hipMalloc(&src, (size_t)ncu * REGION);
  hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)ncu * REGION);
  for (int nreg : {256, 8}) {
    printf("---- %s\n", nreg == 256 ? "private regions (302 MB footprint)" : "one L2-resident region per XCD (1.2 MB each)");
    run<16, 0, true>("LDS-DMA, 16 waves, barrier per slab", src, sink, ncu, nreg);
    run<16, 0, false>("LDS-DMA, 16 waves, no barrier", src, sink, ncu, nreg);
    run<8, 0, true>("LDS-DMA, 8 waves, barrier per slab", src, sink, ncu, nreg);
    run<16, 1, true>("registers, 16 waves, barrier per slab", src, sink, ncu, nreg);
    run<16, 1, false>("registers, 16 waves, no barrier", src, sink, ncu, nreg);
    run<8, 1, true>("registers, 8 waves, barrier per slab", src, sink, ncu, nreg);
  }
  // every workgroup of an XCD sweeping the SAME region (the attention kernel's neighbouring planes share key planes)
  return 0;
}
