// 3x3 stem convolution with Cin = 3 (stride 1 for the MobileNets, stride 2 for SphereFace layer1):
// fp32 NCHW network input -> NHWC activations, as an MFMA implicit GEMM with K = 27 (padded to 32),
// fused folded-BN bias + PReLU / Hardswish.  Replaces the first Conv2dNormActivation of every
// backbone (models/mobilenetv1.py:18-24, mobilenetv2.py:112-118, mobilenetv3.py:133-141,
// sphereface.py:107-115 for layer1).
//
// A block owns an 8x16 output tile: the 3 x (8S+2) x (16S+2) fp32 input patch is staged in LDS
// (zero padded), each wave builds the im2col "B" fragments of a 4x8-pixel sub-tile from LDS
// (k = c*9 + ky*3 + kx, the natural order of weight.reshape(cout, 27)) and multiplies by the
// register-resident weight fragments.  Output layout/epilogue are shared with the pointwise GEMM.
//
// U8 form (fr_embed_u8): the network input is the uint8 HWC image itself.  The block stages the raw
// bytes of its patch rows (4-byte LDS-DMA of the word-aligned superset of each row span) and a
// 256-entry table of ((b / 255) - 0.5) / 0.5 -- ToTensor + Normalize(0.5, 0.5) of inference.py:102-106,
// computed on the host in the same fp32 operations -- and converts bytes to the fp32 patch in LDS
// (0 outside the image: the conv pads the NORMALISED tensor).  From there on both forms are the same
// code, so the results are bit-identical to feeding the transformed fp32 tensor.
#include "kernels.h"

// ACT: the activation at compile time (FR_ACT_PRELU, FR_ACT_HSWISH), or -1: a.e.act at run time.  With a constant the epilogue's
// ladder of uniform branches folds away, and the PReLU slopes are read from LDS as LDS (a pointer that is either the caller's
// array or the staged copy is a generic pointer: flat loads with a full vmcnt + lgkmcnt wait per 8 channels).
template <typename TOut, bool SPLIT, bool U8, int ACT>
__global__ __launch_bounds__(256, 2) void stem_kernel(StemArgs a) {
    constexpr int NS = SPLIT ? 2 : 1;
    constexpr int TH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* patch = reinterpret_cast<float*>(smem);
    const int S = a.stride, TW = a.TW;              // TW: multiple of 8, up to the whole output row
    const int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int n = b;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int gy0 = oy0 * S - 1, gx0 = ox0 * S - 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // bias and PReLU slopes go to LDS with the patch (waited for with it): read from global memory inside the epilogue they
    // were two to four dependent memory round trips per 32-pixel sub-tile, seven sub-tiles per wave
    {
        const int npb = a.NT * 8;                       // 16-byte units per array (<= 64 couts: 16)
        if (wave == 3 && lane < 2 * npb) {
            const int which = lane >= npb, u = lane - which * npb;
            const float* pb = which ? a.e.slope : a.e.bias;
            const void* src = pb ? (const void*)(pb + u * 4) : a.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + a.pb_off), 16, 0, 0);
        }
    }
    EpiArgs e2 = a.e;
    e2.bias = reinterpret_cast<const float*>(smem + a.pb_off);
    if constexpr (ACT >= 0) e2.act = ACT;
    if constexpr (ACT == FR_ACT_PRELU) e2.slope = e2.bias + a.NT * 32;
    else if (a.e.slope) e2.slope = e2.bias + a.NT * 32;

    // fp32 patch by LDS-DMA, 4 bytes per lane (lane after lane = the row-major patch); every copy is
    // issued before the single wait; pixels outside the image read the zero page.  A block owns an
    // 8-row strip of (up to) the whole image width: 100 k tiny blocks of 8x16 pixels were bound by
    // workgroup turnover, not by memory.
    if constexpr (U8) {
        unsigned char* raw = smem + a.raw_off;
        const float* lut = reinterpret_cast<const float*>(smem + a.raw_off + a.lut_off);
        const int RP = a.raw_pitch, WPR = RP >> 2;                   // bytes / words per staged row
        const int gxs = gx0 < 0 ? 0 : gx0;
        const int gxe = min(gx0 + IW, a.W);
        const int span = (gxe - gxs) * 3;                            // bytes of one row inside the image
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.lut + threadIdx.x),
                                         (__attribute__((address_space(3))) void*)(smem + a.raw_off + a.lut_off + wave * 256), 4, 0, 0);
        const FastDiv d_wpr(WPR);
        const int units = IH * WPR;
        for (int i0 = wave * 64; i0 < units; i0 += 256) {
            const int i = i0 + lane;
            if (i < units) {
                unsigned iy, j;
                d_wpr.divmod(i, iy, j);
                const int gy = gy0 + (int)iy;
                const size_t b0 = (((size_t)n * a.H + (gy < 0 ? 0 : gy)) * a.W + gxs) * 3;
                const int mis = (int)(b0 & 3);
                const bool ok = gy >= 0 && gy < a.H && (int)j * 4 < mis + span;
                const unsigned char* src = ok ? a.x8 + (b0 - mis) + j * 4 : reinterpret_cast<const unsigned char*>(a.zeros);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(raw + i0 * 4), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int total = 3 * IH * IW;
        const FastDiv d_hw(IH * IW), d_w(IW);
        for (int i = threadIdx.x; i < total; i += 256) {
            unsigned c, rem, iy, ix;
            d_hw.divmod(i, c, rem);
            d_w.divmod(rem, iy, ix);
            const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const unsigned mis = (unsigned)((((size_t)n * a.H + gy) * a.W + gxs) * 3) & 3u;
                v = lut[raw[iy * RP + mis + (gx - gxs) * 3 + c]];
            }
            patch[i] = v;
        }
    } else {
        const int total = 3 * IH * IW;
        const FastDiv d_hw(IH * IW), d_w(IW);
        for (int i0 = wave * 64; i0 < total; i0 += 256) {
            const int i = i0 + lane;
            if (i < total) {
                unsigned c, rem, iy, ix;
                d_hw.divmod(i, c, rem);
                d_w.divmod(rem, iy, ix);
                const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
                const float* src = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                                       ? a.x + (((long)n * 3 + c) * a.H + gy) * a.W + gx : reinterpret_cast<const float*>(a.zeros);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(patch + i0), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int p = lane & 31, h = lane >> 5;
    // per-lane LDS offsets of the 16 im2col taps this lane feeds (k = 16 s + 8 h + j = c*9 + ky*3 + kx)
    int koff[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            const int c = k / 9, r9 = k - 9 * c;
            const int ky = r9 / 3, kx = r9 - 3 * ky;
            koff[s][j] = k < 27 ? (c * IH * IW + ky * IW + kx) * 4 : 0;     // bytes; k >= 27 (s = 1, h = 1, j >= 3): loaded anyway, replaced by 0
        }
    const bf16x8* W = reinterpret_cast<const bf16x8*>(a.w);
    const int nsub_x = TW >> 3, nsub = 2 * nsub_x;              // 4x8-pixel sub-tiles of the strip
    for (int st = wave; st < nsub; st += 4) {
        const int sy = st / nsub_x, sx = st - sy * nsub_x;
        const int py = 4 * sy + (p >> 3), px = 8 * sx + (p & 7);
        const unsigned char* base = smem + (py * S * IW + px * S) * 4;      // (byte offsets: one add per tap)
        bf16x8 xh[2], xl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Vec8f v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v.v[j] = *reinterpret_cast<const float*>(base + koff[s][j]);     // unconditional (a conditional load is a branch + a dependent LDS round trip per element)
            if (s == 1) {
#pragma unroll
                for (int j = 3; j < 8; ++j) v.v[j] = h ? 0.f : v.v[j];
            }
            if constexpr (SPLIT) split8(v, xh[s], xl[s]);
            else xh[s] = to_frag(v);
        }
        const int gy = oy0 + py, gx = ox0 + px;
        const bool ok = gy < a.Ho && gx < a.Wo;
        const long m = ((long)n * a.Ho + gy) * a.Wo + gx;
        for (int t = 0; t < a.NT; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8* wf = W + ((t * 2 + s) * NS) * 64 + lane;
                bf16x8 wh = wf[0];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
                if constexpr (SPLIT) {
                    bf16x8 wl = wf[64];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], acc, 0, 0, 0);
                }
            }
            if constexpr (std::is_same<TOut, bf16>::value) {
                // wave-private [32 pixels][32 channels] tile in LDS, written back as contiguous runs of 8 pixels
                constexpr int PITCH = 64 + 16;
                unsigned char* stage = smem + a.stage_off + wave * 32 * PITCH;
                epilogue_stage(acc, e2, m, 32 * t + 16 * h, true, stage, PITCH, p, 16 * h);
                const int ncol = min(32, a.e.cout - 32 * t), cr = ncol >> 3;       // 16-byte chunks per pixel: 1..4
                __builtin_amdgcn_wave_barrier();
                for (int c = lane; c < 32 * cr; c += 64) {
                    const int q = c / cr, ch = c - q * cr;
                    const int y = oy0 + 4 * sy + (q >> 3), xx = ox0 + 8 * sx + (q & 7);
                    if (y < a.Ho && xx < a.Wo) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + q * PITCH + ch * 16);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(a.e.y) + (((long)n * a.Ho + y) * a.Wo + xx) * a.e.cout + 32 * t + ch * 8) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else if constexpr (kIsF24<TOut>) {
                // f24: the hi halves take the bf16 route; the lo bytes (32 per pixel and n-tile) a second tile
                constexpr int PITCH = 64 + 16, PITCH_LO = stage_pitch_lo(32);
                unsigned char* stage = smem + a.stage_off + wave * 32 * (PITCH + PITCH_LO);
                unsigned char* stage_lo = stage + 32 * PITCH;
                epilogue_stage_f24(acc, e2, m, 32 * t + 16 * h, true, stage, PITCH, stage_lo, PITCH_LO, p, 16 * h);
                const int ncol = min(32, a.e.cout - 32 * t), cr = ncol >> 3, crl = (ncol + 15) >> 4;
                const int LP = f24_lo_pitch(a.e.cout);
                __builtin_amdgcn_wave_barrier();
                for (int c = lane; c < 32 * cr; c += 64) {
                    const int q = c / cr, ch = c - q * cr;
                    const int y = oy0 + 4 * sy + (q >> 3), xx = ox0 + 8 * sx + (q & 7);
                    if (y < a.Ho && xx < a.Wo) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + q * PITCH + ch * 16);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(a.e.y) + (((long)n * a.Ho + y) * a.Wo + xx) * a.e.cout + 32 * t + ch * 8) = v;
                    }
                }
                for (int c = lane; c < 32 * crl; c += 64) {
                    const int q = c / crl, ch = c - q * crl;
                    const int y = oy0 + 4 * sy + (q >> 3), xx = ox0 + 8 * sx + (q & 7);
                    if (y < a.Ho && xx < a.Wo) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stage_lo + q * PITCH_LO + ch * 16);
                        *reinterpret_cast<u32x4*>(a.e.y_lo + (((long)n * a.Ho + y) * a.Wo + xx) * LP + 32 * t + ch * 16) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else if (a.e.cout < 16) {
                // (8 channels: 32 contiguous bytes per pixel either way; measured mobilenetv1 x0.25 f32 -2.2 % through the tile)
                epilogue_store<TOut>(acc, e2, m, 32 * t + 16 * h, ok);
            } else {
                // fp32: the same route with a [32 pixels][32 channels] fp32 tile (a lane's 16 channels are 64 B that a direct store
                // writes as 16-byte pieces cout * 4 bytes away from the neighbour lane's: the isolated-piece pattern of common.h);
                // mobilenetv3_small f32 +1.7 % (same-box A/B)
                constexpr int PITCH = stage_pitch(32 * 4);
                unsigned char* stage = smem + a.stage_off + wave * 32 * PITCH;
                epilogue_stage_f32(acc, e2, m, 32 * t + 16 * h, true, stage, PITCH, p, 16 * h);
                const int ncol = min(32, a.e.cout - 32 * t), cr = ncol >> 2;       // 16-byte chunks per pixel: 2..8
                __builtin_amdgcn_wave_barrier();
                for (int c = lane; c < 32 * cr; c += 64) {
                    const int q = c / cr, ch = c - q * cr;
                    const int y = oy0 + 4 * sy + (q >> 3), xx = ox0 + 8 * sx + (q & 7);
                    if (y < a.Ho && xx < a.Wo) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + q * PITCH + ch * 16);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(a.e.y) + (((long)n * a.Ho + y) * a.Wo + xx) * a.e.cout + 32 * t + ch * 4) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                (void)ok;
            }
        }
    }
}

int launch_stem(int out_fmt, const StemArgs& a, hipStream_t st) {
    if (a.e.cout % 8 || a.NT != cdiv(a.e.cout, 32) || (a.stride != 1 && a.stride != 2))
        FR_FAIL(FR_ERR_ARG, "stem: inconsistent shapes");
    if (!a.zeros) FR_FAIL(FR_ERR_ARG, "stem: no zero page");
    if (out_fmt == FMT_F24 && !a.e.y_lo) FR_FAIL(FR_ERR_ARG, "stem: f24 output without its lo plane");
    StemArgs b = a;
    b.TW = std::min(112, cdiv(a.Wo, 8) * 8);
    const int IH = 7 * a.stride + 3, IW = (b.TW - 1) * a.stride + 3;
    size_t lds = align_up((size_t)3 * IH * IW * sizeof(float), 16);
    b.stage_off = (int)lds;
    if (out_fmt == FMT_BF16) lds += 4 * 32 * (64 + 16);
    if (out_fmt == FMT_F24) lds += 4 * 32 * (64 + 16 + stage_pitch_lo(32));
    if (out_fmt == FMT_F32 && a.e.cout >= 16) lds += 4 * 32 * stage_pitch(32 * 4);     // (8 channels: direct stores, no tile)
    if (a.NT > 8) FR_FAIL(FR_ERR_UNSUPPORTED, "stem: more than 256 output channels");
    b.pb_off = (int)align_up(lds, 16);
    lds = (size_t)b.pb_off + (size_t)a.NT * 32 * 4 * 2;
    const long blocks = (long)a.N * cdiv(a.Ho, 8) * cdiv(a.Wo, b.TW);
    if (blocks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "stem: grid too large");
    const dim3 grid((unsigned)blocks);
    if (a.x8) {
        // rows are staged as the word-aligned superset of their bytes: the base must be word aligned and an
        // image must be a whole number of words, so that the superset never leaves the caller's buffer
        if (!a.lut || ((size_t)a.x8 & 3) || ((size_t)a.H * a.W * 3) % 4) FR_FAIL(FR_ERR_UNSUPPORTED, "stem (uint8 input): needs a 4-byte aligned buffer and H*W*3 % 4 == 0");
        b.raw_off = (int)align_up(lds, 16);
        b.raw_pitch = (3 * IW + 3 + 3) & ~3;
        b.lut_off = (int)align_up((size_t)IH * b.raw_pitch, 16);
        lds = (size_t)b.raw_off + b.lut_off + 1024;
    }
    const bool prelu = a.e.act == FR_ACT_PRELU && a.e.slope;
#define FR_STEM_GO(TO, SP, U8V) do { \
        if (prelu) hipLaunchKernelGGL((stem_kernel<TO, SP, U8V, FR_ACT_PRELU>), grid, dim3(256), lds, st, b); \
        else if (a.e.act == FR_ACT_HSWISH) hipLaunchKernelGGL((stem_kernel<TO, SP, U8V, FR_ACT_HSWISH>), grid, dim3(256), lds, st, b); \
        else hipLaunchKernelGGL((stem_kernel<TO, SP, U8V, -1>), grid, dim3(256), lds, st, b); } while (0)
    if (a.x8) {
        if (out_fmt == FMT_F24) FR_STEM_GO(f24, true, true);
        else if (out_fmt == FMT_F32) FR_STEM_GO(float, true, true);
        else FR_STEM_GO(bf16, false, true);
    } else if (out_fmt == FMT_F24) FR_STEM_GO(f24, true, false);
    else if (out_fmt == FMT_F32) FR_STEM_GO(float, true, false);
    else FR_STEM_GO(bf16, false, false);
#undef FR_STEM_GO
    FR_LAUNCH_CHECK("stem_kernel");
    return FR_OK;
}
