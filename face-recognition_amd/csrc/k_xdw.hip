// Fused expand(1x1 conv + BN + act) -> depthwise KxK conv (+BN + act): the expanded tensor of an
// inverted-residual block lives only in LDS.  Replaces the first two Conv2dNormActivation units of
// models/mobilenetv3.py:57-80 and models/mobilenetv2.py:22-43 in one launch and removes the write
// and the read of the largest activations of the network (half of its HBM traffic).
//
// Per block (image, output tile TH x TW, chunk of 32*NTc expanded channels):
//   1. stage the block-INPUT tile [(TH-1)S+K] x [(TW-1)S+K] x Cin into LDS (zero outside the image);
//   2. expansion on MFMA: D[cexp][pixel] = We[cexp][:] . X[pixel][:] for every tile pixel (halo
//      included), weights pre-packed as in the pointwise kernel; epilogue bias + PReLU/Hardswish,
//      forced to ZERO at out-of-image pixels (the depthwise conv pads the *expanded* tensor), rounded
//      to the activation type and written to the LDS tile the depthwise phase reads;
//   3. depthwise phase: identical code to dw_kernel (dw_core.h).
// Numerics are the same as the unfused pair (same MFMA k order, same rounding points).
#include "dw_core.h"

#define XDW_MAX_KS 10      // Cin <= 160

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void xdw_kernel(XdwArgs xa) {
    constexpr bool SPLIT = std::is_same<T, float>::value;
    constexpr int NS = SPLIT ? 2 : 1;
    const DwArgs& a = xa.dw;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int b = blockIdx.x;
    const int cc = b % a.cchunks; b /= a.cchunks;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int n = b;
    const int CV = a.C >> 3;
    const int cv0 = cc * a.CBV;
    const int ncv = min(a.CBV, CV - cv0);
    const int t0 = cv0 >> 2;                       // first 32-channel MFMA tile of this chunk
    const int nt = (ncv + 3) >> 2;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int IH = (a.TH - 1) * S + K, IW = (a.TW - 1) * S + K;
    const int npix = IH * IW;
    const int Cin = xa.Cin, KS = xa.KS;
    const int in_row = Cin * (int)sizeof(T) + 16;  // bytes per staged input pixel (padded)

    // LDS carve (all 16-byte aligned)
    V8<T>* tile = reinterpret_cast<V8<T>*>(smem);                                   // [npix][ncv] expanded
    unsigned char* p1 = smem + (size_t)npix * a.CBV * sizeof(V8<T>);
    float* wl = reinterpret_cast<float*>(p1);                                       // dw weights
    float* bl = wl + K * K * a.CBV * 8;                                             // dw bias/slope
    unsigned char* p2 = reinterpret_cast<unsigned char*>(bl + a.CBV * 16);
    unsigned char* xin = p2;                                                        // [npix][in_row]
    unsigned char* wexp = xin + (size_t)npix * in_row;                              // [nt][KS][NS][64][16 B]
    float* eb = reinterpret_cast<float*>(wexp + (size_t)((a.CBV + 3) >> 2) * KS * NS * 1024);   // [nt*32] bias, then slope

    const T* X = reinterpret_cast<const T*>(a.x);
    const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;

    // 1. stage the input tile: units of 8 channels
    const int cin_v = Cin >> 3;
    const FastDiv d_cinv(cin_v), d_iw(IW);
    for (int i = threadIdx.x; i < npix * cin_v; i += 256) {
        unsigned cv, pix, iy, ix;
        d_cinv.divmod(i, pix, cv);
        d_iw.divmod(pix, iy, ix);
        const int gy = gy0 + iy, gx = gx0 + ix;
        V8<T> v;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            v = *reinterpret_cast<const V8<T>*>(X + (((long)n * a.H + gy) * a.W + gx) * Cin + cv * 8);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v.v[j] = (T)0.f;
        }
        *reinterpret_cast<V8<T>*>(xin + (size_t)pix * in_row + cv * sizeof(V8<T>)) = v;
    }
    // expand weights of this chunk + bias/slope
    {
        const int total = nt * KS * NS * 64;
        const u32x4* src = reinterpret_cast<const u32x4*>(xa.we) + (size_t)t0 * KS * NS * 64;
        for (int i = threadIdx.x; i < total; i += 256) reinterpret_cast<u32x4*>(wexp)[i] = src[i];
        for (int i = threadIdx.x; i < nt * 32; i += 256) {
            eb[i] = xa.bias_e[t0 * 32 + i];
            eb[nt * 32 + i] = xa.slope_e ? xa.slope_e[t0 * 32 + i] : 0.f;
        }
    }
    dw_stage_params<T, K, S>(a, wl, bl, cv0, ncv);
    __syncthreads();

    // 2. expansion
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int p = lane & 31, h = lane >> 5;
        const int ptiles = (npix + 31) >> 5;
        for (int pt = wave; pt < ptiles; pt += 4) {
            const int pi = pt * 32 + p;
            const bool valid = pi < npix;
            const int pic = valid ? pi : npix - 1;
            unsigned iy, ix;
            d_iw.divmod(pic, iy, ix);
            const int gy = gy0 + iy, gx = gx0 + ix;
            const bool in_img = valid && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            bf16x8 xh[XDW_MAX_KS], xl[SPLIT ? XDW_MAX_KS : 1];
#pragma unroll
            for (int s = 0; s < XDW_MAX_KS; ++s) {
                if (s < KS) {
                    const int kk = 16 * s + 8 * h;
                    if (kk < Cin) {
                        const T* src = reinterpret_cast<const T*>(xin + (size_t)pic * in_row) + kk;
                        if constexpr (SPLIT) split8(load8(src), xh[s], xl[s]);
                        else xh[s] = *reinterpret_cast<const bf16x8*>(src);
                    } else {
                        xh[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        if constexpr (SPLIT) xl[s] = xh[s];
                    }
                }
            }
            for (int t = 0; t < nt; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < XDW_MAX_KS; ++s) {
                    if (s < KS) {
                        const bf16x8* wf = reinterpret_cast<const bf16x8*>(wexp) + ((t * KS + s) * NS) * 64 + lane;
                        const bf16x8 wh = wf[0];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
                        if constexpr (SPLIT) {
                            const bf16x8 wlo = wf[64];
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh[s], acc, 0, 0, 0);
                        }
                    }
                }
                if (valid) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int cvl = t * 4 + h * 2 + g;            // local channel vector
                        if (cvl < ncv) {
                            Vec8f v;
                            const float* bb = eb + t * 32 + h * 16 + g * 8;
                            const float* ss = eb + nt * 32 + t * 32 + h * 16 + g * 8;
#pragma unroll
                            for (int j = 0; j < 8; ++j) v.v[j] = acc[8 * g + j] + bb[j];
                            act8(v, xa.act_e, ss);
                            if (!in_img) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) v.v[j] = 0.f;
                            }
                            store8(reinterpret_cast<T*>(tile + (size_t)pi * ncv + cvl), v);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // 3. depthwise
    dw_compute<T, K, S>(a, tile, wl, bl, n, oy0, ox0, cv0, ncv, IW);
}

static size_t xdw_lds_bytes(bool f32, int k, int s, int TH, int TW, int CBV, int Cin, int KS) {
    const int IH = (TH - 1) * s + k, IW = (TW - 1) * s + k;
    const size_t npix = (size_t)IH * IW;
    const int es = f32 ? 4 : 2, ns = f32 ? 2 : 1;
    return npix * CBV * 8 * es + (size_t)k * k * CBV * 32 + (size_t)CBV * 64       // tile, dw weights, dw bias/slope
         + npix * (Cin * es + 16) + (size_t)((CBV + 3) / 4) * KS * ns * 1024 + (size_t)((CBV + 3) / 4) * 256;
}

// Tile / channel-chunk choice: chunk = whole 32-channel MFMA tiles; LDS <= 64 KB; minimise an estimate of
// (depthwise passes + expansion MFMA work + staged bytes) per useful output.
static bool xdw_pick_tile(bool f32, int k, int s, DwArgs& a, int Cin, int KS, size_t* lds_out) {
    const int CV = a.C / 8, NT = (CV + 3) / 4;
    double best = 1e30;
    bool found = false;
    const int ths[] = {2, 4, 7, 8, 14, 16};
    const int tws[] = {4, 8, 12, 16, 28, 32, 56};
    for (int th : ths) for (int tw : tws) {
        if (th - a.Ho >= 2 && th != ths[0]) continue;
        if (tw - a.Wo >= 4) continue;
        for (int ntc = NT; ntc >= 1; --ntc) {
            const int cbv = std::min(CV, 4 * ntc);
            const size_t lds = xdw_lds_bytes(f32, k, s, th, tw, cbv, Cin, KS);
            if (lds > 64 * 1024) continue;
            const int tyc = cdiv(a.Ho, th), txc = cdiv(a.Wo, tw), ccs = cdiv(CV, cbv);
            const double IH = (th - 1) * s + k, IW = (tw - 1) * s + k, npix = IH * IW;
            const double passes = cdiv(th * (tw / 4) * cbv, 256);
            const double mfma = std::ceil(npix / 32.0 / 4.0) * ntc * KS * (f32 ? 3 : 1);      // per-wave MFMA count
            double cost = (double)tyc * txc * ccs *
                (passes * (k * k * 1.2 + 2.0) + mfma * 0.5 + npix * (Cin / 8) / 256.0 * 1.5 + npix * cbv / 256.0 * 0.5);
            cost *= (lds > 40 * 1024) ? 1.15 : 1.0;                                          // occupancy
            if (cost < best) {
                best = cost; found = true; *lds_out = lds;
                a.TH = th; a.TW = tw; a.CBV = cbv; a.tiles_y = tyc; a.tiles_x = txc; a.cchunks = ccs;
            }
            break;
        }
    }
    return found;
}

bool xdw_supported(int Cin, int k, int stride) {
    return Cin % 8 == 0 && Cin <= 16 * XDW_MAX_KS && (k == 3 || k == 5) && (stride == 1 || stride == 2);
}

template <typename T>
static int launch_xdw_t(int k, int s, const XdwArgs& xa, size_t lds, hipStream_t st) {
    const DwArgs& a = xa.dw;
    dim3 grid((unsigned)((long)a.N * a.tiles_y * a.tiles_x * a.cchunks));
#define FR_XDW_CASE(KK, SS) if (k == KK && s == SS) { hipLaunchKernelGGL((xdw_kernel<T, KK, SS>), grid, dim3(256), lds, st, xa); }
    FR_XDW_CASE(3, 1) else FR_XDW_CASE(3, 2) else FR_XDW_CASE(5, 1) else FR_XDW_CASE(5, 2)
    else FR_FAIL(FR_ERR_UNSUPPORTED, "xdw: only k in {3,5}, stride in {1,2}");
#undef FR_XDW_CASE
    FR_LAUNCH_CHECK("xdw_kernel");
    return FR_OK;
}

int launch_xdw(bool f32, int k, int stride, XdwArgs xa, hipStream_t st) {
    DwArgs& a = xa.dw;
    if (a.C % 8 || a.N <= 0 || !xdw_supported(xa.Cin, k, stride) || xa.KS != cdiv(xa.Cin, 16))
        FR_FAIL(FR_ERR_ARG, "xdw: unsupported shape");
    size_t lds = 0;
    if (!xdw_pick_tile(f32, k, stride, a, xa.Cin, xa.KS, &lds)) FR_FAIL(FR_ERR_UNSUPPORTED, "xdw: no tile fits LDS");
    if ((long)a.N * a.tiles_y * a.tiles_x * a.cchunks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "xdw: grid too large");
    return f32 ? launch_xdw_t<float>(k, stride, xa, lds, st) : launch_xdw_t<bf16>(k, stride, xa, lds, st);
}
