// Depthwise KxK convolution (K = 3 or 5, stride 1 or 2), NHWC, LDS-staged, fused folded-BN bias +
// PReLU / Hardswish.  Replaces nn.Conv2d(groups=C)+BatchNorm2d+activation (utils/layers.py:15-84
// as used at :143-149, models/mobilenetv2.py:36-43, models/mobilenetv3.py:70-80).
//
// HBM-bound: a block stages a (TH*S+K-S) x (TW*S+K-S) x (8*CBV channel) input tile into LDS with
// coalesced 16-byte loads (zero-filled outside the image), then every thread produces strips of
// 4 horizontally adjacent outputs for one 8-channel vector, re-using each LDS read for up to K/S
// outputs.  fp32 accumulate, fp32 weights (tap-major [K*K][C], BN folded in), one rounding at the store.
#include "dw_core.h"

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
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
    const FastDiv d_ncv(ncv), d_iw(IW);
    for (int i = threadIdx.x; i < nin; i += 256) {
        unsigned cv, pix, iy, ix;
        d_ncv.divmod(i, pix, cv);
        d_iw.divmod(pix, iy, ix);
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
    dw_stage_params<T, K, S>(a, wl, bl, cv0, ncv);
    __syncthreads();
    dw_compute<T, K, S>(a, tile, wl, bl, n, oy0, ox0, cv0, ncv, IW);
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
