// Development probes of libwmz_hip.so: process-wide switches for kernel timing experiments (tools/), NOT part of the product
// interface (include/wmz.h) -- production callers never touch them, and with all of them at their defaults (NULL / 0) the library
// has no global state.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-development probe: workgroup 0 of the fused layer kernel writes the shader clock at its stage boundaries into
 * buf (device, 8 waves x 64 int64); NULL (default) switches the probe off. */
int wmz_debug_fused_timestamps(void* buf);
/* Ablation switches of the fused per-token kernel (timing experiments only, results are garbage): 1 = skip the MFMA loops,
 * 2 = skip the weight DMA and its waits, 4 = skip the per-slab workgroup barrier (bits combine); 0 = product behaviour. */
int wmz_debug_fused_knobs(int dbg);
/* Same for the 16-wide-plane attention forward kernel (16 waves x 64 int64). */
int wmz_debug_attn_timestamps(void* buf);
/* development knobs of the attention forward: dbg = ablation switches (1 skip the per-tile compute, 2 skip the K/V
 * staging: timing experiments only, results are garbage), variant = reserved (the library carries one instantiation per
 * shape class; other schedules are separate builds, tools/build_variant.py); (0, 0) is the product behaviour. */
int wmz_debug_attn_knobs(int dbg, int variant);

/* development knobs of the direct convolution (csrc/conv_direct.hip): skew = start delay (units of s_sleep 127) of the second
 * workgroup per CU; dbg = ablation switches (1 skip the epilogue, 2 one weight slab only: timing experiments, results are garbage). */
int wmz_debug_conv_knobs(int skew, int dbg);

/* development knob of the nn.Linear kernel (csrc/linear_fwd.hip): dma = 0 runs the small-M GEMMs on the register-staged K loop
 * instead of the LDS-DMA ring (A/B timing; same results). */
int wmz_debug_linear_knobs(int dma);

/* Timeline probe for captured steps: one thread writes the device's constant-rate wall clock (wall_clock64: 100 MHz) into
 * buf[slot] (device int64) when the launch executes on `stream` -- a marker between the phases of a hipGraph replay, where a
 * profiler's own per-node cost (rocprofv3 --kernel-trace: ~9 us a node) would distort the very overlap under study. */
int wmz_debug_stamp(void* buf, int slot, void* stream);

#ifdef __cplusplus
}
#endif
