// Depthwise KxK convolution (K = 3 or 5, stride 1 or 2), NHWC, LDS-staged, fused folded-BN bias +
// PReLU / Hardswish.  Replaces nn.Conv2d(groups=C)+BatchNorm2d+activation (utils/layers.py:15-84
// as used at :143-149, models/mobilenetv2.py:36-43, models/mobilenetv3.py:70-80).
//
// HBM-bound: a block stages a (TH*S+K-S) x (TW*S+K-S) x (8*CBV channel) input tile into LDS with
// coalesced 16-byte loads (zero-filled outside the image), then every thread produces strips of
// 4 horizontally adjacent outputs for one 8-channel vector, re-using each LDS read for up to K/S
// outputs.  fp32 accumulate, fp32 weights (tap-major [K*K][C], BN folded in), one rounding at the store.
#include "kernels.h"

template <typename T>
struct V8 { T v[8]; };   // 8 channels as stored (16 B for bf16, 32 B for fp32)

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
    constexpr int SW = 4;
    constexpr int IWS = (SW - 1) * S + K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int b = blockIdx.x;
    const int cc = b % a.cchunks; b /= a.cchunks;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int n = b;
    const int CV = a.C >> 3;
    const int cv0 = cc * a.CBV;
    const int ncv = min(a.CBV, CV - cv0);
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int IH = (a.TH - 1) * S + K, IW = (a.TW - 1) * S + K;

    V8<T>* tile = reinterpret_cast<V8<T>*>(smem);                       // [IH*IW][ncv]
    float* wl = reinterpret_cast<float*>(smem + (size_t)IH * IW * a.CBV * sizeof(V8<T>));   // [K*K][2][ncv][4]
    float* bl = wl + K * K * a.CBV * 8;                                 // [2][ncv][4] bias, then slope
    const T* X = reinterpret_cast<const T*>(a.x);

    const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;
    const int nin = IH * IW * ncv;
    for (int i = threadIdx.x; i < nin; i += 256) {
        const int cv = i % ncv, pix = i / ncv;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = gy0 + iy, gx = gx0 + ix;
        V8<T> v;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            v = *reinterpret_cast<const V8<T>*>(X + (((long)n * a.H + gy) * a.W + gx) * a.C + (cv0 + cv) * 8);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v.v[j] = (T)0.f;
        }
        tile[i] = v;
    }
    for (int i = threadIdx.x; i < K * K * ncv * 8; i += 256) {
        const int j = i & 3; int r = i >> 2;
        const int cv = r % ncv; r /= ncv;
        const int half = r & 1, tap = r >> 1;
        wl[i] = a.w[(long)tap * a.C + (cv0 + cv) * 8 + half * 4 + j];
    }
    for (int i = threadIdx.x; i < ncv * 8; i += 256) {
        const int j = i & 3; int r = i >> 2;
        const int cv = r % ncv; const int half = r / ncv;
        const int c = (cv0 + cv) * 8 + half * 4 + j;
        bl[i] = a.bias[c];
        bl[ncv * 8 + i] = a.slope ? a.slope[c] : 0.f;
    }
    __syncthreads();

    const int nstrips = a.TW / SW;
    const int nitems = a.TH * nstrips * ncv;
    T* Y = reinterpret_cast<T*>(a.y);
    for (int it = threadIdx.x; it < nitems; it += 256) {
        const int cv = it % ncv; int r = it / ncv;
        const int sx = r % nstrips, oy = r / nstrips;
        float acc[SW][8];
        {
            f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + (0 * ncv + cv) * 4);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + (1 * ncv + cv) * 4);
#pragma unroll
            for (int o = 0; o < SW; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[o][j] = b0[j]; acc[o][4 + j] = b1[j]; }
        }
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            float wr[K][8];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (((ky * K + kx) * 2 + 0) * ncv + cv) * 4);
                f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + (((ky * K + kx) * 2 + 1) * ncv + cv) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { wr[kx][j] = w0[j]; wr[kx][4 + j] = w1[j]; }
            }
            const V8<T>* row = tile + ((oy * S + ky) * IW + sx * SW * S) * ncv + cv;
#pragma unroll
            for (int ix = 0; ix < IWS; ++ix) {
                Vec8f v = load8(reinterpret_cast<const T*>(row + ix * ncv));
#pragma unroll
                for (int o = 0; o < SW; ++o) {
                    const int kx = ix - o * S;
                    if (kx >= 0 && kx < K) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[o][j] = fmaf(v.v[j], wr[kx][j], acc[o][j]);
                    }
                }
            }
        }
        const int gy = oy0 + oy;
        if (gy < a.Ho) {
            f32x4 s0, s1;
            if (a.act == FR_ACT_PRELU) {
                s0 = *reinterpret_cast<const f32x4*>(bl + ncv * 8 + (0 * ncv + cv) * 4);
                s1 = *reinterpret_cast<const f32x4*>(bl + ncv * 8 + (1 * ncv + cv) * 4);
            }
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                const int gx = ox0 + sx * SW + o;
                if (gx < a.Wo) {
                    Vec8f v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float slope = 0.f;
                        if (a.act == FR_ACT_PRELU) slope = j < 4 ? s0[j & 3] : s1[j & 3];
                        v.v[j] = apply_act(acc[o][j], a.act, slope);
                    }
                    store8(Y + (((long)n * a.Ho + gy) * a.Wo + gx) * a.C + (cv0 + cv) * 8, v);
                }
            }
        }
    }
}

static size_t dw_lds_bytes(bool f32, int k, int s, int TH, int TW, int CBV) {
    const int IH = (TH - 1) * s + k, IW = (TW - 1) * s + k;
    return (size_t)IH * IW * CBV * (f32 ? 32 : 16) + (size_t)k * k * CBV * 32 + (size_t)CBV * 64;
}

// Pick the output tile and channel chunk: keep LDS <= 40 KB (>= 3 blocks per CU), prefer all
// channels in one block (long contiguous NHWC runs), then the least padded-work + halo overhead.
static void dw_pick_tile(bool f32, int k, int s, DwArgs& a) {
    const int CV = a.C / 8;
    double best = 1e30;
    const int ths[] = {2, 4, 7, 8, 14, 16};
    const int tws[] = {4, 8, 12, 16, 28, 32, 56};
    for (int th : ths) for (int tw : tws) {
        if (th > a.Ho && th != ths[0]) { if (th - a.Ho >= 2) continue; }
        if (tw - a.Wo >= 4) continue;
        for (int cbv = CV; cbv >= 1; cbv = (cbv > 1 ? (cbv + 1) / 2 : 0)) {
            if (dw_lds_bytes(f32, k, s, th, tw, cbv) > 40 * 1024) continue;
            const int tyc = cdiv(a.Ho, th), txc = cdiv(a.Wo, tw), ccs = cdiv(CV, cbv);
            const int items = th * (tw / 4) * cbv;
            const double passes = cdiv(items, 256);
            const double IH = (th - 1) * s + k, IW = (tw - 1) * s + k;
            // cost ~ compute passes + staged input vectors / 256 (load loop), per useful output
            double cost = (double)tyc * txc * ccs * (passes * (k * k / 4.0 + 2.0) + IH * IW * cbv / 256.0 * 1.5);
            if (cbv < CV && cbv * 16 < 64) cost *= 1.5;     // short contiguous runs
            if (cost < best) { best = cost; a.TH = th; a.TW = tw; a.CBV = cbv; a.tiles_y = tyc; a.tiles_x = txc; a.cchunks = ccs; }
            break;   // largest fitting chunk for this tile
        }
    }
}

template <typename T>
static int launch_dw_t(int k, int s, const DwArgs& a, size_t lds, hipStream_t st) {
    dim3 grid((unsigned)((long)a.N * a.tiles_y * a.tiles_x * a.cchunks));
#define FR_DW_CASE(KK, SS) if (k == KK && s == SS) { hipLaunchKernelGGL((dw_kernel<T, KK, SS>), grid, dim3(256), lds, st, a); }
    FR_DW_CASE(3, 1) else FR_DW_CASE(3, 2) else FR_DW_CASE(5, 1) else FR_DW_CASE(5, 2)
    else FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: only k in {3,5}, stride in {1,2}");
#undef FR_DW_CASE
    FR_LAUNCH_CHECK("dw_kernel");
    return FR_OK;
}

int launch_dw(bool f32, int k, int stride, DwArgs a, hipStream_t st) {
    if (a.C % 8 || a.N <= 0) FR_FAIL(FR_ERR_ARG, "depthwise: C must be a multiple of 8");
    if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: only k in {3,5}, stride in {1,2}");
    a.TH = 0;
    dw_pick_tile(f32, k, stride, a);
    if (a.TH == 0) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: no tile fits LDS");
    if ((long)a.N * a.tiles_y * a.tiles_x * a.cchunks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "depthwise: grid too large");
    const size_t lds = dw_lds_bytes(f32, k, stride, a.TH, a.TW, a.CBV);
    return f32 ? launch_dw_t<float>(k, stride, a, lds, st) : launch_dw_t<bf16>(k, stride, a, lds, st);
}
