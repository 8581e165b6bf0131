// Library-internal entry points shared between translation units (not part of include/wmz.h).
#pragma once
#include <hip/hip_runtime.h>

// linear_bwd.hip: the second stage of the deterministic weight-gradient reduction -- sums `nsplit` partial results laid out as
// workspace[block of 256 floats of the flat [N, K] result][split][256] (+ [split][N] bias partials behind them) into dW / dbias;
// taps > 0: dW is nn.Conv2d's own [co, ci, taps] layout (k = tap * cin_p + c), channel padding cropped.
int wmz_wgrad_reduce_launch(const float* workspace, float* dW, float* dbias, long NK, int nsplit, int N, int overwrite, int taps,
                            int cin_p, int co, int ci, hipStream_t stream);
// conv_wgrad.hip: direct 3x3 weight gradient (bf16, stride 1, pad 1, Cout = 128, Cin in {64, 128}, H % 8 == 0, W % 16 == 0)
int wmz_convw_supported(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype);
long wmz_convw_workspace_floats(int B, int H, int W, int Cin, int Cout);
int wmz_convw_launch(const void* x, const void* dy, float* dW, float* dbias, int B, int H, int W, int Cin, int Cout, int overwrite,
                     int conv_layout_co, int conv_layout_ci, float* workspace, long workspace_floats, hipStream_t stream);
