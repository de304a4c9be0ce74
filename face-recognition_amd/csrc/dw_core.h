// Shared depthwise compute phase: consumes an LDS-resident input tile [IH*IW][ncv] of 8-channel
// vectors (zero outside the image), LDS-resident tap weights/bias/slopes, and writes the output tile.
#pragma once
#include "kernels.h"

template <typename T>
struct V8 { T v[8]; };   // 8 channels as stored (16 B for bf16, 32 B for fp32)

// wl: [K*K][2][ncv][4] fp32 tap weights; bl: [2][ncv][4] bias then [2][ncv][4] slopes.
template <typename T, int K, int S>
FR_DEV void dw_stage_params(const DwArgs& a, float* wl, float* bl, int cv0, int ncv) {
    const FastDiv d_ncv(ncv);
    for (int i = threadIdx.x; i < K * K * ncv * 8; i += 256) {
        const int j = i & 3; unsigned r, cv;
        d_ncv.divmod(i >> 2, r, cv);
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
}

template <typename T, int K, int S>
FR_DEV void dw_compute(const DwArgs& a, const V8<T>* tile, const float* wl, const float* bl,
                       int n, int oy0, int ox0, int cv0, int ncv, int IW) {
    constexpr int SW = 4;
    constexpr int IWS = (SW - 1) * S + K;
    const int nstrips = a.TW / SW;
    const int nitems = a.TH * nstrips * ncv;
    T* Y = reinterpret_cast<T*>(a.y);
    const FastDiv d_ncv(ncv), d_ns(nstrips);
    for (int it = threadIdx.x; it < nitems; it += 256) {
        unsigned cv, r, sx, oy;
        d_ncv.divmod(it, r, cv);
        d_ns.divmod(r, oy, sx);
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
            float sl[8];
            if (a.act == FR_ACT_PRELU) {
                f32x4 s0 = *reinterpret_cast<const f32x4*>(bl + ncv * 8 + (0 * ncv + cv) * 4);
                f32x4 s1 = *reinterpret_cast<const f32x4*>(bl + ncv * 8 + (1 * ncv + cv) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { sl[j] = s0[j]; sl[4 + j] = s1[j]; }
            }
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                const int gx = ox0 + sx * SW + o;
                if (gx < a.Wo) {
                    Vec8f v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.v[j] = acc[o][j];
                    act8(v, a.act, sl);
                    store8(Y + (((long)n * a.Ho + gy) * a.Wo + gx) * a.C + (cv0 + cv) * 8, v);
                }
            }
        }
    }
}
