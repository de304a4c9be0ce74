// Argument structs and host launchers of every kernel in libfr_hip.so.
#pragma once
#include "common.h"

// Activation storage formats of a plan's NHWC buffers (the `fmt` / `in_fmt` / `out_fmt` arguments below).
// FMT_F24: common.h, struct f24 -- hi plane where the bf16 tensor would be, lo plane given separately (`*_lo`).
enum { FMT_BF16 = 0, FMT_F32 = 1, FMT_F24 = 2 };

// ---- pointwise (1x1) conv / linear as MFMA GEMM:  Y[m][co] = act(sum_k X[m][k] W[co][k] + b) (+R)
struct PwArgs {
    const void* x;        // [M, K] activations (bf16 or fp32; f24: the hi plane), row stride K
    const unsigned char* x_lo;   // f24: lo plane [M][f24_lo_pitch(K)]
    const bf16* w;        // packed fragments [NT][KS][NS][64][8]  (NS = 2 (hi,lo) for fp32 input)
    const float* se;      // [M/HW, K] per-image channel scale applied to X on load, or NULL
    EpiArgs e;
    long M;
    int K, KS, NT, HW;
    int chunk_ks;         // k-steps of W staged in LDS at a time
    int stage_off;        // set by the launcher: LDS offset of the output staging tiles
    int ntg;              // set by the launcher: n-tiles per column group (large-K kernel)
    // fc form only: x is [M][x_tiles][K] fp32 partial sums (squeeze-excite pooling written by the depthwise
    // kernel); the row fed to the GEMM is their sum in tile order times x_scale.  0: x is [M][K] as usual.
    int x_tiles;
    float x_scale;
};
// in_fmt / out_fmt: FMT_*; f24 goes in and out together (FMT_F24, FMT_F24)
int launch_pw(int in_fmt, int out_fmt, const PwArgs& a, hipStream_t st);
const char* pw_kernel_name(int in_fmt, int out_fmt, int K, int cout, bool se, int HW);
// Small-M form (M = batch): SE fc layers, Linear(+BN1d).  fp32 output, no SE scale, no residual.  in_f32: fp32 rows (never f24)
int launch_fc(bool in_f32, const PwArgs& a, hipStream_t st);
// long-K Linear: S = fc_splitk_slices(K) > 0 slices of K, partial sums [S][M][cout] fp32 in `partial`, then a reduce launch
int fc_splitk_slices(int K);
int launch_fc_splitk(bool in_f32, const PwArgs& a, float* partial, int S, hipStream_t st);
// LDS bytes per (n-tile, k-step) = 1024 * NS; used by the planner to pick chunk_ks.  wide: fp32-class input (FMT_F32 / FMT_F24)
void pw_pick_tiling(bool wide, int NT, int KS, int* ntb, int* chunk_ks);

// ---- depthwise KxK conv, NHWC, LDS-staged
struct DwArgs {
    const void* x;        // [N,H,W,C]
    const unsigned char* x_lo;   // f24: lo planes of x and y
    unsigned char* y_lo;
    const float* w;       // [K*K][C] folded
    const float* bias;    // [C]
    const float* slope;   // [C] or NULL
    void* y;              // [N,Ho,Wo,C]
    int N, H, W, C, Ho, Wo, pad, act;
    int TH, TW, CBV;      // output tile (TW multiple of 4) and channel-vectors (8 ch) per block
    int tiles_x, tiles_y, cchunks;
    const void* zeros;    // >= 16 B of device zeros: LDS-DMA source for out-of-image taps and absent slopes
    float* pool;          // optional [N][pool_tiles][C] per-tile channel sums of the OUTPUT (squeeze-excite pooling), or NULL
    int pool_tiles;       // tiles per image the caller sized `pool` for (must equal dw_tiles_per_image)
};
// Depthwise + the following 1x1 conv in one kernel (bf16 or fp32 plans; all channels of a tile in one block).
// Optionally the expand 1x1 conv in front of the depthwise conv is computed in the same kernel too.
struct DwpExpand {
    const void* xin;      // block input [N,H,W,Cin] (H,W = the depthwise input size), bf16 or fp32 as the plan stores it
    const unsigned char* xin_lo;   // f24: its lo plane
    const bf16* we;       // expand weights packed [NT][KS][NS][64][8]
    const float* bias;    // padded to a multiple of 32
    const float* slope;   // PReLU slopes or NULL
    int act, Cin;
    // stem form: the conv in front of the depthwise conv is the network's 3x3 stem (Cin = 3, stride 1, pad 1, weights packed
    // as for stem_kernel) computed from the fp32 NCHW network input `xs`; xin is unused.  res_lds: the 1x1 conv's residual is
    // the stem output (kept in LDS; EpiArgs::residual must be NULL)
    int stem, res_lds;
    const float* xs;
};
// cin: 0 = no expand conv; DWP_STEM / DWP_STEM_RES = the stem form (bf16 / fp32 plans, C <= 32), without / with the LDS residual
enum { DWP_STEM = -1, DWP_STEM_RES = -2 };
bool dwp_supported(int fmt, int k, int stride, int C, int Ho, int Wo, int cout, int cin);
bool dwp_is_chunked(int fmt, int k, int stride, int C, int Ho, int Wo, int cout, int cin);   // the launch will take dwc_kernel (expanded channels in chunks of 32)
int launch_dwp(int fmt, int k, int stride, DwArgs a, const bf16* w_pw_packed, const EpiArgs& pw_epilogue, const DwpExpand* expand, hipStream_t st);
// Spatial tiles per image launch_dw will use for this layer shape (sizes DwArgs::pool).
int dw_tiles_per_image(int fmt, int k, int stride, int C, int Ho, int Wo);
int launch_dw(int fmt, int k, int stride, DwArgs a, hipStream_t st);

// ---- 3x3 stem conv, Cin = 3, fp32 NCHW in -> NHWC out, MFMA with K = 27 (padded to 32)
struct StemArgs {
    const float* x;       // [N,3,H,W]
    const bf16* w;        // packed [NT][2][NS][64][8]
    EpiArgs e;
    int N, H, W, Ho, Wo, stride, NT;
    const void* zeros;    // >= 16 B of device zeros (LDS-DMA source outside the image)
    int stage_off;        // set by the launcher: LDS offset of the output staging tiles
    int pb_off;           // set by the launcher: LDS offset of the bias, then the PReLU slopes ([NT*32] floats each)
    int TW;               // set by the launcher: output columns per block (8-row strips)
    // uint8 HWC input (fr_embed_u8): x8 != nullptr replaces x; lut = 256 floats ((b/255)-0.5)/0.5
    const unsigned char* x8;
    const float* lut;
    int raw_off, raw_pitch, lut_off;   // set by the launcher
};
int launch_stem(int out_fmt, const StemArgs& a, hipStream_t st);

// ---- dense KxK conv (K = 3) / large-K linear (K = 1) as MFMA implicit GEMM
struct ConvArgs {
    const void* x;        // [N,H,W,Cin] activations
    const bf16* w;        // packed [cout tile][ks][NS][BN][32], ks = chunk * taps + tap (chunk = bk input channels)
    EpiArgs e;
    long M;               // N*Ho*Wo
    int H, W, Ho, Wo, Cin, ksize, stride, pad;
    int bk;               // channels per K slice the weights were packed with (32; 64 for the bf16 LDS-DMA path)
    const void* zeros;    // >= 64 B of device zeros (source of padded taps in the LDS-DMA path), or NULL
    int spb;              // conv_halo16_kernel: K slices per barrier (1..3; 0 = default 2), from FR_CONV_SPB when the plan was created
    int halo;             // 1: take conv_halo_kernel where conv_halo_rows() allows it (the plan packed with bk 32 for that); 2: conv_halo16_kernel (weights in its row order)
};
int launch_conv(bool in_f32, bool out_f32, const ConvArgs& a, hipStream_t st);
int conv_block_couts(int cout, int out_hw, bool halo);   // couts per block tile (64 or 128) of a layer with out_hw output pixels per image (halo: it runs the shared-pixel kernel): the packing granularity
int conv_halo_rows(int ksize, int stride, int pad, int W, int bk);   // > 0: shape the 3x3 stride-1 shared-pixel kernel takes (bk 32); rows of its staged range at 128 pixels

// ---- weighted spatial reduction: out[n][c] = sum_p x[n][p][c] * w[p][c] (+ b[c]); w NULL -> 1/P
struct ReduceArgs {
    const void* x;        // [N,P,C]
    const unsigned char* x_lo;   // f24: lo plane
    const float* w;       // [P][C] or NULL (mean)
    const float* bias;    // [C] or NULL
    float* out;           // [N,C] fp32
    int N, P, C;
    float scale;          // w == NULL: multiply the sum by this instead of 1/P when non-zero
};
int launch_reduce(int fmt, const ReduceArgs& a, hipStream_t st);

// ---- layout conversion (plan input for non-stem first layers, taps, final non-linear outputs)
int launch_nchw_to_nhwc(const float* x, void* y, void* y_lo, int fmt, int N, int C, int H, int W, hipStream_t st);
int launch_nhwc_to_nchw(const void* x, const void* x_lo, int fmt, float* y, int N, int C, int H, int W, hipStream_t st);
