// Depthwise KxK convolution (K = 3 or 5, stride 1 or 2), NHWC, LDS-staged, fused folded-BN bias +
// PReLU / Hardswish.  Replaces nn.Conv2d(groups=C)+BatchNorm2d+activation (utils/layers.py:15-84
// as used at :143-149, models/mobilenetv2.py:36-43, models/mobilenetv3.py:70-80).
//
// HBM-bound: a block stages a (TH*S+K-S) x (TW*S+K-S) x (8*CBV channel) input tile into LDS with
// coalesced 16-byte loads (zero-filled outside the image), then every thread produces strips of
// 4 horizontally adjacent outputs for one 8-channel vector, re-using each LDS read for up to K/S
// outputs.  fp32 accumulate, fp32 weights (tap-major [K*K][C], BN folded in), one rounding at the store.
#include "dw_core.h"

typedef __attribute__((address_space(3))) void* dw_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* dw_glb_ptr_t;

// Staging is LDS-DMA (global_load_lds, 16 B per lane): every lane of the block issues all its
// copies back to back and the block waits ONCE, so a tile costs one HBM round trip instead of one
// per loop iteration (the register-staged loop compiled to load -> wait -> ds_write per iteration).
// The DMA writes lane after lane, which is exactly the [pixel][channel-vector] order of the tile;
// out-of-image taps read a zero page.
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

    constexpr int UPE = sizeof(V8<T>) / 16;                             // 16-byte units per 8-channel vector
    V8<T>* tile = reinterpret_cast<V8<T>*>(smem);                       // [IH*IW][ncv]
    unsigned char* wl_b = smem + (size_t)IH * IW * a.CBV * sizeof(V8<T>);
    float* wl = reinterpret_cast<float*>(wl_b);                         // [K*K][2][ncv][4]
    float* bl = wl + K * K * a.CBV * 8;                                 // [2][ncv][4] bias, then slope
    const unsigned char* X = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* Z = reinterpret_cast<const unsigned char*>(a.zeros);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;
    const int nunits = IH * IW * ncv * UPE;
    const FastDiv d_ncv(ncv), d_iw(IW);
    for (int u0 = wave * 64; u0 < nunits; u0 += 256) {
        const int u = u0 + lane;
        if (u < nunits) {
            unsigned cv, pix, iy, ix;
            d_ncv.divmod((unsigned)u / UPE, pix, cv);
            d_iw.divmod(pix, iy, ix);
            const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
            const unsigned char* src = Z;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                src = X + ((((long)n * a.H + gy) * a.W + gx) * a.C + (cv0 + cv) * 8) * sizeof(T) + (u % UPE) * 16;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + u0 * 16), 16, 0, 0);
        }
    }
    // tap weights: unit (tap, half, cv) = 4 floats w[tap][(cv0+cv)*8 + 4*half ..]
    const int nw = K * K * 2 * ncv;
    for (int u0 = wave * 64; u0 < nw; u0 += 256) {
        const int u = u0 + lane;
        if (u < nw) {
            unsigned r, cv;
            d_ncv.divmod(u, r, cv);
            const float* src = a.w + (long)(r >> 1) * a.C + (cv0 + cv) * 8 + (r & 1) * 4;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(wl_b + u0 * 16), 16, 0, 0);
        }
    }
    // bias [2][ncv][4], then slopes [2][ncv][4]
    const int nb = 2 * ncv;
    if (wave == 0 && lane < 2 * nb) {                                   // 2*nb <= 64 is checked by the launcher
        const int which = lane >= nb, r = lane - which * nb;
        const int half = r / ncv, cv = r - half * ncv;
        const float* base = which ? a.slope : a.bias;
        const void* src = base ? (const void*)(base + (cv0 + cv) * 8 + half * 4) : (const void*)Z;
        __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(reinterpret_cast<unsigned char*>(bl)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        for (int cbv = std::min(CV, 16); cbv >= 1; cbv = (cbv > 1 ? (cbv + 1) / 2 : 0)) {
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
    if (a.CBV > 16 || !a.zeros) FR_FAIL(FR_ERR_ARG, "depthwise: channel chunk above 128 channels or no zero page");
    if ((long)a.N * a.tiles_y * a.tiles_x * a.cchunks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "depthwise: grid too large");
    const size_t lds = dw_lds_bytes(f32, k, stride, a.TH, a.TW, a.CBV);
    return f32 ? launch_dw_t<float>(k, stride, a, lds, st) : launch_dw_t<bf16>(k, stride, a, lds, st);
}
