// Dense KxK convolution (K = 3, stride 1|2, pad 1: SphereFace) and large-K Linear (K = 1 on a 1x1
// map: SphereFace fc 25088 -> 512) as a bf16 MFMA implicit GEMM with fused bias + PReLU + residual.
// Replaces Conv2dNormActivation(norm_layer=None) + PReLU and `x + block(x)` (models/sphereface.py:
// 18-48, 107-117) and nn.Linear(512*7*7, 512) (models/sphereface.py:84, 126-127).
//
// D[cout][pixel] = sum_{tap,c} W[cout][tap][c] * X[pixel shifted by tap][c]  (compute-bound, MFMA).
// The K loop walks (tap, 32-channel chunk) slices.  Per slice a block stages, through registers
// (issue the global loads for slice s+1, run the MFMAs of slice s, then write LDS: T14 in the CDNA
// guide), a [BN couts][32] weight tile (host-packed in exactly this order) and a [BM pixels][32]
// activation tile gathered from NHWC with zero fill at the borders.  LDS rows are 64 B + 16 B pad
// so every ds_read_b128 fragment read is bank-conflict free.  LDS use is 20-50 KB, so 3-4 blocks
// share a CU and hide each other's barriers.  Wave tile = TI x TJ MFMA 32x32x16 tiles; the weight
// rows are permuted (common.h) so each lane holds 16 consecutive output channels of a pixel.
// fp32 plans (TIn = float) split both operands into bf16 hi+lo planes at staging: 3 MFMAs per product.
#include "kernels.h"

#define CV_ROW 80     // bytes per LDS row: 32 bf16 + 16 B pad

template <typename TIn, typename TOut, int TI, int TJ, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs a) {
    constexpr bool SPLIT = std::is_same<TIn, float>::value;
    constexpr int NS = SPLIT ? 2 : 1;
    constexpr int BN = 32 * TI * WM, BM = 32 * TJ * WN;
    static_assert(WM * WN == 4 && BM == 128, "4 waves, 128 pixels per block");
    constexpr int A_BYTES = BN * CV_ROW, B_BYTES = BM * CV_ROW;
    constexpr int BUF = NS * (A_BYTES + B_BYTES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const long m0 = (long)blockIdx.x * BM;
    const int nt0 = blockIdx.y;                          // cout tile of BN
    const int chunks = a.Cin >> 5;
    const int KS = a.ksize * a.ksize * chunks;

    // staging roles: unit = (row, 32-byte half of the 64-byte row)
    const int srow = tid >> 1, shalf = tid & 1;
    // B operand: this thread's pixel
    const long spix = m0 + srow;
    const bool spix_ok = spix < a.M;
    int sn = 0, soy = 0, sox = 0;
    if (spix_ok) { const int hw = a.Ho * a.Wo; sn = (int)(spix / hw); const int r = (int)(spix - (long)sn * hw); soy = r / a.Wo; sox = r - soy * a.Wo; }
    const TIn* X = reinterpret_cast<const TIn*>(a.x);
    const bf16* Wp = a.w + (size_t)nt0 * KS * NS * BN * 32;       // [ks][NS][BN][32]

    // Hoisted addressing: the centre-pixel pointer and a 9-bit per-tap validity mask are computed once
    // per thread; inside the K loop a slice only adds wave-uniform (scalar) offsets.
    const unsigned char* xc = reinterpret_cast<const unsigned char*>(
        X + (((long)sn * a.H + (long)soy * a.stride) * a.W + (long)sox * a.stride) * a.Cin + shalf * 16);
    unsigned tapmask = 0;
    for (int t = 0; t < a.ksize * a.ksize; ++t) {
        const int ky = t / a.ksize, kx = t - ky * a.ksize;
        const int iy = soy * a.stride + ky - a.pad, ix = sox * a.stride + kx - a.pad;
        if (spix_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) tapmask |= 1u << t;
    }
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(Wp + (size_t)srow * 32 + shalf * 16);
    constexpr long W_SLICE = (long)NS * BN * 32 * 2;                 // bytes of packed weights per slice
    // scalar slice cursor (tap, chunk) advanced once per prefetch
    int cur_tap = 0, cur_chunk = 0, cur_ky = 0, cur_kx = 0;
    long cur_w = 0;

    u32x4 pa[NS][2], pb[SPLIT ? 4 : 2];
    auto prefetch = [&](int) {
        if (srow < BN) {
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                const unsigned char* src = wsrc + cur_w + (long)sp * BN * 64;
                pa[sp][0] = *reinterpret_cast<const u32x4*>(src);
                pa[sp][1] = *reinterpret_cast<const u32x4*>(src + 16);
            }
        }
        const long toff = (((long)(cur_ky - a.pad) * a.W + (cur_kx - a.pad)) * a.Cin + (cur_chunk << 5)) * (long)sizeof(TIn);
        if ((tapmask >> cur_tap) & 1u) {
            const unsigned char* src = xc + toff;
#pragma unroll
            for (int i = 0; i < (SPLIT ? 4 : 2); ++i) pb[i] = *reinterpret_cast<const u32x4*>(src + 16 * i);
        } else {
#pragma unroll
            for (int i = 0; i < (SPLIT ? 4 : 2); ++i) pb[i] = u32x4{0, 0, 0, 0};
        }
        cur_w += W_SLICE;
        ++cur_tap;                                         // tap inner, channel chunk outer (see pack_conv_weights)
        if (++cur_kx == a.ksize) {
            cur_kx = 0;
            if (++cur_ky == a.ksize) { cur_ky = 0; cur_tap = 0; ++cur_chunk; }
        }
    };
    auto commit = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
        if (srow < BN) {
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                unsigned char* dst = base + sp * A_BYTES + srow * CV_ROW + shalf * 32;
                *reinterpret_cast<u32x4*>(dst) = pa[sp][0];
                *reinterpret_cast<u32x4*>(dst + 16) = pa[sp][1];
            }
        }
        unsigned char* dst = base + NS * A_BYTES + srow * CV_ROW + shalf * 32;
        if constexpr (SPLIT) {
            // 16 fp32 -> 16 bf16 hi + 16 bf16 lo
            u32x4 hi[2], lo[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float x0 = __uint_as_float(pb[i][2 * e]), x1 = __uint_as_float(pb[i][2 * e + 1]);
                    const bf16 h0 = (bf16)x0, h1 = (bf16)x1;
                    hi[i >> 1][(i & 1) * 2 + e] = pack2bf((float)h0, (float)h1);
                    lo[i >> 1][(i & 1) * 2 + e] = pack2bf(x0 - (float)h0, x1 - (float)h1);
                }
            }
            *reinterpret_cast<u32x4*>(dst) = hi[0];
            *reinterpret_cast<u32x4*>(dst + 16) = hi[1];
            *reinterpret_cast<u32x4*>(dst + B_BYTES) = lo[0];
            *reinterpret_cast<u32x4*>(dst + B_BYTES + 16) = lo[1];
        } else {
            *reinterpret_cast<u32x4*>(dst) = pb[0];
            *reinterpret_cast<u32x4*>(dst + 16) = pb[1];
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    prefetch(0);
    commit(0);
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KS) prefetch(ks + 1);
        const unsigned char* ab = smem + buf * BUF;
        const unsigned char* bb = ab + NS * A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[NS][TI], bfr[NS][TJ];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    af[sp][i] = *reinterpret_cast<const bf16x8*>(ab + sp * A_BYTES + ((wm * TI + i) * 32 + p) * CV_ROW + (s * 16 + 8 * h) * 2);
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    bfr[sp][j] = *reinterpret_cast<const bf16x8*>(bb + sp * B_BYTES + ((wn * TJ + j) * 32 + p) * CV_ROW + (s * 16 + 8 * h) * 2);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    if constexpr (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (ks + 1 < KS) commit(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const long m = m0 + (wn * TJ + j) * 32 + p;
#pragma unroll
        for (int i = 0; i < TI; ++i)
            epilogue_store<TOut>(acc[i][j], a.e, m, nt0 * BN + (wm * TI + i) * 32 + 16 * h, m < a.M);
    }
}


// ---- bf16 path: LDS-DMA staging (global_load_lds, 16 B per lane) --------------------------------
// The register-staged kernel above is LDS-write bound (ds_write_b128 moves ~79 B/clk/CU); here both
// operand tiles go global -> LDS directly.  One wave instruction writes 1 KiB linearly = 16 rows of
// 64 B, so the bank-conflict swizzle is applied on the SOURCE side (lane (row, c) fetches logical
// chunk c ^ ((row>>2)&3)) and undone on the fragment read.  Out-of-image lanes fetch from a zero
// page.  Double-buffered; one vmcnt(0)+barrier per K slice, hidden by 3-4 co-resident blocks.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// XCD-aware block order: workgroups go to the 8 XCDs round-robin in dispatch order and each XCD has its own
// L2.  Dispatch slot L = (XCD k = L % 8, turn j = L / 8) works on pixel tile k * (tiles / 8) + j / ny and
// cout tile j % ny: one XCD owns a contiguous range of pixel tiles (vertical halos of neighbouring tiles
// meet in one L2) and runs the cout tiles of a pixel tile back to back (the pixels are fetched once, not
// once per cout tile).  The last tiles % 8 pixel tiles keep the natural order.
__device__ __forceinline__ void conv_tile_of_block(unsigned& pt, unsigned& nt) {
    const unsigned gx = gridDim.x, ny = gridDim.y, per = gx >> 3, full = per * 8u * ny;
    const unsigned L = blockIdx.x + gx * blockIdx.y;
    if (L < full) {
        const unsigned k = L & 7u, j = L >> 3, q = j / ny;
        pt = k * per + q; nt = j - q * ny;
    } else {
        const unsigned rest = gx - 8u * per, Lt = L - full, q = Lt / rest;
        pt = 8u * per + (Lt - q * rest); nt = q;
    }
}

template <typename TOut, int TI, int TJ, int WM, int WN, int BK>
__global__ __launch_bounds__(256, 2) void conv_glds_kernel(ConvArgs a) {
    constexpr int BN = 32 * TI * WM, BM = 32 * TJ * WN;
    static_assert(WM * WN == 4 && BM == 128, "4 waves, 128 pixels per block");
    static_assert(BK == 32 || BK == 64, "K slice of 32 or 64 channels");
    constexpr int ROWB = BK * 2;                      // bytes per LDS row
    constexpr int CR = BK / 8;                        // 16-byte chunks per row
    constexpr int RG = 64 / CR;                       // rows written by one wave instruction (1 KiB)
    constexpr int A_BYTES = BN * ROWB, B_BYTES = BM * ROWB, BUF = A_BYTES + B_BYTES;
    constexpr int GA = BN / RG / 4, GB = BM / RG / 4; // groups per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // bank swizzle of row r: the 16 rows of a ds_read_b128 lane group must hit 16 distinct 16-B slots
    auto swz = [](int row) { return BK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7); };

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    unsigned pt, nt;
    conv_tile_of_block(pt, nt);
    const long m0 = (long)pt * BM;
    const int nt0 = (int)nt;
    const int chunks = a.Cin / BK;
    const int KS = a.ksize * a.ksize * chunks;

    const int sub = lane / CR, pc = lane % CR;
    const unsigned char* wsrc[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int row = RG * (wave + 4 * j) + sub;
        const int lc = pc ^ swz(row);
        wsrc[j] = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)nt0 * KS * BN + row) * ROWB + lc * 16;
    }
    const unsigned char* xc[GB];
    unsigned tapmask[GB];
    const bf16* X = reinterpret_cast<const bf16*>(a.x);
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int row = RG * (wave + 4 * j) + sub;
        const int lc = pc ^ swz(row);
        // the launcher guarantees M + 128 < 2^31: 32-bit unsigned divisions, no 64-bit ones
        const unsigned pix = (unsigned)m0 + (unsigned)row;
        const bool ok = pix < (unsigned)a.M;
        unsigned sn = 0, soy = 0, sox = 0;
        if (ok) { const unsigned hw = (unsigned)(a.Ho * a.Wo); sn = pix / hw; const unsigned r = pix - sn * hw; soy = r / (unsigned)a.Wo; sox = r - soy * (unsigned)a.Wo; }
        xc[j] = reinterpret_cast<const unsigned char*>(
            X + (((long)sn * a.H + (long)(soy * a.stride)) * a.W + (long)(sox * a.stride)) * a.Cin) + lc * 16;
        const int iy0 = (int)soy * a.stride - a.pad, ix0 = (int)sox * a.stride - a.pad;
        unsigned mask = 0, bit = 1;
        for (int ky = 0; ky < a.ksize; ++ky) {
            const bool yok = ok && (unsigned)(iy0 + ky) < (unsigned)a.H;
            for (int kx = 0; kx < a.ksize; ++kx, bit <<= 1)
                if (yok && (unsigned)(ix0 + kx) < (unsigned)a.W) mask |= bit;
        }
        tapmask[j] = mask;
    }
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(a.zeros);
    int cur_tap = 0, cur_chunk = 0, cur_ky = 0, cur_kx = 0;
    long cur_w = 0;

    auto stage = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < GA; ++j)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[j] + cur_w), (lds_ptr_t)(base + (wave + 4 * j) * 1024), 16, 0, 0);
        const long toff = (((long)(cur_ky - a.pad) * a.W + (cur_kx - a.pad)) * a.Cin + cur_chunk * BK) * 2;
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const unsigned char* src = ((tapmask[j] >> cur_tap) & 1u) ? xc[j] + toff : zsrc;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(base + A_BYTES + (wave + 4 * j) * 1024), 16, 0, 0);
        }
        cur_w += (long)BN * ROWB;
        ++cur_tap;                                         // tap inner, channel chunk outer (see pack_conv_weights)
        if (++cur_kx == a.ksize) {
            cur_kx = 0;
            if (++cur_ky == a.ksize) { cur_ky = 0; cur_tap = 0; ++cur_chunk; }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: logical chunk 2s+h sits at physical (2s+h)^sw = (h^sw) ^ 2s
    int aoff[TI], boff[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) { const int row = (wm * TI + i) * 32 + p; aoff[i] = row * ROWB + ((h ^ swz(row)) << 4); }
#pragma unroll
    for (int j = 0; j < TJ; ++j) { const int row = (wn * TJ + j) * 32 + p; boff[j] = A_BYTES + row * ROWB + ((h ^ swz(row)) << 4); }

    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KS) stage(buf ^ 1);
        const unsigned char* bb = smem + buf * BUF;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(bb + (aoff[i] ^ (s << 5)));
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(bb + (boff[j] ^ (s << 5)));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (std::is_same<TOut, bf16>::value) {
        // bf16 output leaves through LDS (both operand buffers are free after the last barrier): the MFMA layout
        // gives every lane 16 channels of ONE pixel, i.e. isolated 32-byte pieces in global memory (PMC: 1.45x the
        // algorithmic write bytes); staged as [128 pixels][BN couts], the block writes whole BN*2-byte row segments
        constexpr int PITCH = BN * 2 + 16;
        // same arithmetic as epilogue_stage; the loads of a cout block (bias, slopes, the residual rows of its TJ
        // pixels) are issued together and unconditionally at clamped addresses instead of one dependent round trip
        // after the other behind the bounds branches
        const bool prelu = a.e.act == FR_ACT_PRELU;
        const bf16* const R = reinterpret_cast<const bf16*>(a.e.residual);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int col = (wm * TI + i) * 32 + 16 * h, c0 = nt0 * BN + col;
            const bool ok0 = c0 < a.e.cout, ok1 = c0 + 8 < a.e.cout;
            const int cs0 = ok0 ? c0 : 0, cs1 = ok1 ? c0 + 8 : 0;
            f32x4 bs[4], sl[4];
            u32x4 rr[TJ][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bs[q] = *reinterpret_cast<const f32x4*>(a.e.bias + cs0 + 4 * q);
                bs[2 + q] = *reinterpret_cast<const f32x4*>(a.e.bias + cs1 + 4 * q);
            }
            if (prelu) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    sl[q] = *reinterpret_cast<const f32x4*>(a.e.slope + cs0 + 4 * q);
                    sl[2 + q] = *reinterpret_cast<const f32x4*>(a.e.slope + cs1 + 4 * q);
                }
            }
            if (R) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const long mm = min(m0 + (wn * TJ + j) * 32 + p, a.M - 1) * (long)a.e.cout;
                    rr[j][0] = *reinterpret_cast<const u32x4*>(R + mm + cs0);
                    rr[j][1] = *reinterpret_cast<const u32x4*>(R + mm + cs1);
                }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int row = (wn * TJ + j) * 32 + p;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    Vec8f v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v.v[r] = acc[i][j][8 * g + r] + bs[2 * g][r]; v.v[4 + r] = acc[i][j][8 * g + 4 + r] + bs[2 * g + 1][r]; }
                    if (prelu) prelu8(v, sl[2 * g], sl[2 * g + 1]);
                    else if (a.e.act != FR_ACT_NONE) act8(v, a.e.act, nullptr);
                    if (R) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v.v[2 * q] += __uint_as_float(rr[j][g][q] << 16);
                            v.v[2 * q + 1] += __uint_as_float(rr[j][g][q] & 0xFFFF0000u);
                        }
                    }
                    if (m0 + row < a.M && (g ? ok1 : ok0)) stage_put(smem, PITCH, row, col + 8 * g, v);
                }
            }
        }
        __syncthreads();
        constexpr int CPR = BN / 8;                                   // 16-byte chunks per staged row
        const int cols = min(BN, a.e.cout - nt0 * BN);                 // couts of this tile that exist
        bf16* Y = reinterpret_cast<bf16*>(a.e.y);
        for (int c = tid; c < 128 * CPR; c += 256) {
            const int row = c / CPR, ch = c - row * CPR;
            const long m = m0 + row;
            if (m < a.M && ch * 8 < cols)
                *reinterpret_cast<u32x4*>(Y + m * (long)a.e.cout + nt0 * BN + ch * 8) =
                    *reinterpret_cast<const u32x4*>(smem + row * PITCH + ch * 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const long m = m0 + (wn * TJ + j) * 32 + p;
#pragma unroll
            for (int i = 0; i < TI; ++i)
                epilogue_store<TOut>(acc[i][j], a.e, m, nt0 * BN + (wm * TI + i) * 32 + 16 * h, m < a.M);
        }
    }
}


// ---- 3x3, stride 1, bf16: the pixel operand is staged ONCE per 32-channel chunk and shared by the 9 taps ----
// conv_glds_kernel fetches a [128 px][BK] activation tile per (tap, chunk) slice: nine fetches of nearly the same
// pixels, and its L2 -> LDS fill runs at the rate the microarchitecture guide measures as the ceiling for
// gathering rows from L2 into LDS (~28 B/clk/CU; PMC: profiles/r01_conv_pmc_*.txt), which caps a 128 x 128 tile
// (64 flop per filled byte) near 1.1 PFLOP/s.  Fewer filled bytes per flop is the only way up.  With stride 1
// and NHWC the tap (ky,kx) of output pixel m is input pixel m + (ky-1)*W + (kx-1) in the SAME linear order, so
// a block stages the linear pixel range [m0 - W - 1, m0 + 128 + W + 1) of one channel chunk (128 + 2W + 2 rows
// of 64 B, one LDS-DMA row per pixel) and every tap reads its B fragments from that range at a shifted row.
// Pixels of that range that are not neighbours in the image (row wrap, other image, outside the tensor) are
// never used: a lane whose tap falls outside the image reads zeros instead (per-lane 9-bit mask).
// Filled bytes per block and chunk: 9 x 8 KB of weights + <= 16 KB of pixels instead of 9 x (8 + 8) KB.
// Rows are 64 B with the same XOR swizzle as above keyed on the physical row; 32 consecutive rows at any shift
// cover every residue mod 16 within each 16-lane group of ds_read_b128, so the shifted reads stay conflict-free.
template <typename TOut, int TI, int TJ, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_halo_kernel(ConvArgs a, int RB) {
    constexpr int NW = WM * WN, BN = 32 * TI * WM, BM = 32 * TJ * WN, ROWB = 64;
    static_assert((NW == 4 || NW == 8) && (BM == 128 || BM == 256), "4 or 8 waves, 128 or 256 pixels per block");
    constexpr int A_BYTES = BN * ROWB;                // one (tap, chunk) slice of weights
    constexpr int NA = BN / 16, GA = (NA + NW - 1) / NW;   // weight DMA instructions per slice, and per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B_BYTES = RB * ROWB;                    // one chunk of the pixel range
    unsigned char* const bbase = smem + 2 * A_BYTES;
    const int zrow = 2 * A_BYTES + 2 * B_BYTES;       // 256 B of zeros = one pass over the 64 banks (256-B aligned)

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    unsigned pt, nt;
    conv_tile_of_block(pt, nt);
    const long m0 = (long)pt * BM;
    const int nt0 = (int)nt;
    const int chunks = a.Cin >> 5;
    const int KS = 9 * chunks;
    const int W = a.W;

    if (tid < 64) *reinterpret_cast<unsigned*>(smem + zrow + tid * 4) = 0u;

    // DMA roles: one wave instruction fills 16 rows x 64 B; lane = (row sub, physical 16-B chunk pc)
    const int sub = lane >> 2, pc = lane & 3;
    const int lc = pc ^ ((sub >> 2) & 3);             // logical chunk this lane fetches (16 | row base, so sub decides)
    const unsigned char* wsrc[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j)
        wsrc[j] = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)nt0 * KS * BN + 16 * (wave + NW * j) + sub) * ROWB + lc * 16;
    const unsigned char* const X = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* const zsrc = reinterpret_cast<const unsigned char*>(a.zeros);
    const long gbase = m0 - W - 1;                    // input pixel of row 0 of the staged range
    const long rowbytes = (long)a.Cin * 2;
    // piece i (16 rows) of the pixel range of channel chunk c -> buffer c & 1
    auto stage_b = [&](int i, int c) {
        if (16 * i < RB) {
            const long g = gbase + 16 * i + sub;
            const unsigned char* src = (g >= 0 && g < a.M) ? X + g * rowbytes + (long)c * 64 + lc * 16 : zsrc;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(bbase + (c & 1) * B_BYTES + i * 1024), 16, 0, 0);
        }
    };
    auto stage_a = [&](int ks) {
#pragma unroll
        for (int j = 0; j < GA; ++j)
            if (NA % NW == 0 || wave + NW * j < NA)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[j] + (long)ks * A_BYTES), (lds_ptr_t)(smem + (ks & 1) * A_BYTES + (wave + NW * j) * 1024), 16, 0, 0);
    };

    // per-lane tap validity of the TJ pixels this lane multiplies, and their centre rows in the staged range
    unsigned bmask[TJ];
    int rcen[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int row = (wn * TJ + j) * 32 + p;
        rcen[j] = row + W + 1;
        const unsigned pix = (unsigned)m0 + (unsigned)row;        // the launcher guarantees M + 128 < 2^31
        unsigned mask = 0;
        if (pix < (unsigned)a.M) {
            const unsigned hw = (unsigned)(a.H * W), r = pix % hw, oy = r / (unsigned)W, ox = r - oy * (unsigned)W;
            unsigned bit = 1;
            for (int ky = -1; ky <= 1; ++ky) {
                const bool yok = (unsigned)((int)oy + ky) < (unsigned)a.H;
                for (int kx = -1; kx <= 1; ++kx, bit <<= 1)
                    if (yok && (unsigned)((int)ox + kx) < (unsigned)W) mask |= bit;
            }
        }
        bmask[j] = mask;
    }
    int aoff[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) { const int row = (wm * TI + i) * 32 + p; aoff[i] = row * ROWB + ((h ^ ((row >> 2) & 3)) << 4); }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int i = wave; 16 * i < RB; i += NW) stage_b(i, 0);
    stage_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int tap = 0, chunk = 0, toff = -W - 1, kx = 0;
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) stage_a(ks + 1);
        if (chunk + 1 < chunks) stage_b(wave + NW * tap, chunk + 1);      // the next chunk's pixels, a ninth per tap
        const unsigned char* ab = smem + (ks & 1) * A_BYTES;
        const int bb = 2 * A_BYTES + (chunk & 1) * B_BYTES;
        int baddr[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int r = rcen[j] + toff;
            const int at = bb + r * ROWB + ((h ^ ((r >> 2) & 3)) << 4);
            // a masked lane reads zeros from the SAME banks its pixel row would have used: redirecting it to one fixed
            // address put it on top of another lane's banks (PMC: one conflict cycle per LDS instruction)
            baddr[j] = ((bmask[j] >> tap) & 1u) ? at : zrow + (at & 255);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ab + (aoff[i] ^ (s << 5)));
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(smem + (baddr[j] ^ (s << 5)));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++tap; ++toff;                                     // tap inner, channel chunk outer (see pack_conv_weights)
        if (++kx == 3) { kx = 0; toff += W - 3; }
        if (tap == 9) { tap = 0; toff = -W - 1; ++chunk; }
    }

    if constexpr (std::is_same<TOut, bf16>::value) {
        constexpr int PITCH = BN * 2 + 16;                 // staged [BM pixels][BN couts], as in conv_glds_kernel
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int row = (wn * TJ + j) * 32 + p;
            const long m = m0 + row;
#pragma unroll
            for (int i = 0; i < TI; ++i)
                epilogue_stage(acc[i][j], a.e, m, nt0 * BN + (wm * TI + i) * 32 + 16 * h, m < a.M, smem, PITCH, row,
                               (wm * TI + i) * 32 + 16 * h);
        }
        __syncthreads();
        constexpr int CPR = BN / 8;
        const int cols = min(BN, a.e.cout - nt0 * BN);
        bf16* Y = reinterpret_cast<bf16*>(a.e.y);
        for (int c = tid; c < BM * CPR; c += 64 * NW) {
            const int row = c / CPR, ch = c - row * CPR;
            const long m = m0 + row;
            if (m < a.M && ch * 8 < cols)
                *reinterpret_cast<u32x4*>(Y + m * (long)a.e.cout + nt0 * BN + ch * 8) =
                    *reinterpret_cast<const u32x4*>(smem + row * PITCH + ch * 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const long m = m0 + (wn * TJ + j) * 32 + p;
#pragma unroll
            for (int i = 0; i < TI; ++i)
                epilogue_store<TOut>(acc[i][j], a.e, m, nt0 * BN + (wm * TI + i) * 32 + 16 * h, m < a.M);
        }
    }
}

// ---- the same kernel on v_mfma_f32_16x16x32_bf16 ----------------------------------------------------------------
// One MFMA covers the whole 32-channel K slice (K = 32), 16 couts x 16 pixels per instruction; per wave and slice
// 2*TI A fragments + 2*TJ B fragments feed 4*TI*TJ MFMAs of 16 cycles (the same LDS reads and MFMA cycles as the
// 32x32x16 form), but the chip holds a higher clock on this shape (micro-architecture guide, DVFS note 7; measured on
// the 1:N search: +10 %).  Differences from conv_halo_kernel: the fragment of a 16-row block is row l & 15, 16-byte
// piece l >> 4 of the 64-byte row; rows are swizzled with ((row >> 2) & 1) << 1 (conflict-free for that read pattern
// at ANY row shift -- checked exhaustively over the four ds_read_b128 lane groups); the weight rows of a wave's 64
// couts are packed so that lane group kg = l >> 4 ends up with couts 16 kg .. 16 kg + 15 of its pixels
// (plan.hip: pack_conv_weights, halo16 order), which keeps the epilogue's 32-byte NHWC pieces.
// SPB = K slices per barrier: the weight ring holds two STAGES of SPB consecutive slices (contiguous in the packed
// weights), one DMA + one barrier per stage.  With SPB = 2 the pixel frame of the next chunk is staged only during
// iterations whose slices all belong to the current chunk (a stage may straddle a chunk boundary: 9 taps per chunk),
// so a frame is never written while one of its taps is still being read.  Same accumulation order for every SPB.
template <typename TOut, int TI, int TJ, int WM, int WN, int SPB>
__global__ __launch_bounds__(64 * WM * WN, 4) void conv_halo16_kernel(ConvArgs a, int RB) {   // 4 waves per SIMD: <= 128 VGPRs
    constexpr int NW = WM * WN, BN = 32 * TI * WM, BM = 32 * TJ * WN, ROWB = 64;
    static_assert(TI == 2, "a wave owns 64 couts (four 16-row blocks)");
    static_assert((NW == 4 || NW == 8) && (BM == 128 || BM == 256), "4 or 8 waves, 128 or 256 pixels per block");
    static_assert(SPB == 1 || SPB == 2 || SPB == 3, "slices per barrier");
    constexpr int A_BYTES = BN * ROWB, S_BYTES = SPB * A_BYTES;
    constexpr int NA = SPB * BN / 16, GA = (NA + NW - 1) / NW;     // 1-KiB pieces of a stage, pieces per wave
    constexpr int NBI = 2 * TI, NBJ = 2 * TJ;          // 16-row blocks per wave: couts, pixels
    constexpr int BCALLS = SPB;                        // frame pieces per wave and iteration (see the stage_b calls below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B_BYTES = RB * ROWB;
    unsigned char* const bbase = smem + 2 * S_BYTES;
    const int zrow = 2 * S_BYTES + 2 * B_BYTES;

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, kg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    unsigned pt, nt;
    conv_tile_of_block(pt, nt);
    const long m0 = (long)pt * BM;
    const int nt0 = (int)nt;
    const int chunks = a.Cin >> 5;
    const int KS = 9 * chunks;
    const int W = a.W;

    if (tid < 64) *reinterpret_cast<unsigned*>(smem + zrow + tid * 4) = 0u;

    const int sub = lane >> 2, pc = lane & 3;
    const int lc = pc ^ (((sub >> 2) & 1) << 1);       // logical chunk this lane fetches
    const unsigned char* wsrc[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j)
        wsrc[j] = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)nt0 * KS * BN + 16 * (wave + NW * j) + sub) * ROWB + lc * 16;
    const unsigned char* const X = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* const zsrc = reinterpret_cast<const unsigned char*>(a.zeros);
    const int gbase = (int)m0 - W - 1;                  // launcher: M + 256 < 2^31
    const int rowbytes = a.Cin * 2;
    auto stage_b = [&](int i, int c) {
        if (16 * i < RB) {
            const int g = gbase + 16 * i + sub;
            const unsigned char* src = (g >= 0 && g < (int)a.M) ? X + (long)g * rowbytes + (c * 64 + lc * 16) : zsrc;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(bbase + (c & 1) * B_BYTES + i * 1024), 16, 0, 0);
        }
    };
    auto stage_a = [&](int it) {
#pragma unroll
        for (int j = 0; j < GA; ++j)
            if (NA % NW == 0 || wave + NW * j < NA)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[j] + (long)it * S_BYTES), (lds_ptr_t)(smem + (it & 1) * S_BYTES + (wave + NW * j) * 1024), 16, 0, 0);
    };

    // per-lane tap validity of the NBJ pixels this lane multiplies (pixel 16 pb + c16 of the wave's range)
    unsigned bmask[NBJ];
    int rcen[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int row = wn * TJ * 32 + j * 16 + c16;
        rcen[j] = row + W + 1;
        const unsigned pix = (unsigned)m0 + (unsigned)row;
        unsigned mask = 0;
        if (pix < (unsigned)a.M) {
            const unsigned hw = (unsigned)(a.H * W), r = pix % hw, oy = r / (unsigned)W, ox = r - oy * (unsigned)W;
            unsigned bit = 1;
            for (int ky = -1; ky <= 1; ++ky) {
                const bool yok = (unsigned)((int)oy + ky) < (unsigned)a.H;
                for (int kx = -1; kx <= 1; ++kx, bit <<= 1)
                    if (yok && (unsigned)((int)ox + kx) < (unsigned)W) mask |= bit;
            }
        }
        bmask[j] = mask;
    }
    int aoff[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) { const int row = wm * 32 * TI + i * 16 + c16; aoff[i] = row * ROWB + ((kg ^ (((row >> 2) & 1) << 1)) << 4); }

    f32x4 acc[NBI][NBJ];
#pragma unroll
    for (int i = 0; i < NBI; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i = wave; 16 * i < RB; i += NW) stage_b(i, 0);
    stage_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int tap = 0, chunk = 0, toff = -W - 1, kx = 0, bnext = 0;
    const int NIT = KS / SPB;                          // host: KS % SPB == 0
    // LDS addresses of the NBJ pixel fragments of the slice (tap, chunk): shifted row of the frame, or the zero row
    int baddr[NBJ];
    auto frag_addresses = [&]() {
        const int bb = 2 * S_BYTES + (chunk & 1) * B_BYTES;
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int r = rcen[j] + toff;
            const int at = bb + r * ROWB + ((kg ^ (((r >> 2) & 1) << 1)) << 4);
            baddr[j] = ((bmask[j] >> tap) & 1u) ? at : zrow + (at & 255);
        }
    };
    frag_addresses();
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        if (it + 1 < NIT) stage_a(it + 1);
        // next chunk's pixel frame: only while every slice of this iteration reads the current chunk's frame
        if (chunk + 1 < chunks && tap + SPB <= 9) {
#pragma unroll
            for (int q = 0; q < BCALLS; ++q) stage_b(wave + NW * (bnext + q), chunk + 1);
            bnext += BCALLS;
        }
        const unsigned char* const ab0 = smem + (it & 1) * S_BYTES;
#pragma unroll
        for (int s = 0; s < SPB; ++s) {
            const unsigned char* ab = ab0 + s * A_BYTES;
            // all eight fragment reads of the slice first (one exposed LDS round trip per slice instead of the three
            // the scheduler leaves when it re-uses fragment registers), then the 16 MFMAs back to back with the
            // address arithmetic of the NEXT slice between them (VALU issue is free beside a 16-cycle MFMA)
            bf16x8 af[NBI], bfr[NBJ];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NBJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(smem + baddr[j]);
#pragma unroll
            for (int i = 0; i < NBI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i]);
            ++tap; ++toff;
            if (++kx == 3) { kx = 0; toff += W - 3; }
            if (tap == 9) { tap = 0; toff = -W - 1; ++chunk; bnext = 0; }
            __builtin_amdgcn_sched_barrier(0);
            frag_addresses();
#pragma unroll
            for (int i = 0; i < NBI; ++i)
#pragma unroll
                for (int j = 0; j < NBJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NBI * NBJ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // up to three VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // lane: pixels 16 j + c16 (j < NBJ), couts 16 kg .. 16 kg + 15 of the wave's 64: value 4 i + r = acc[i][j][r].
    // Same arithmetic as epilogue_stage (bias, activation, residual, one rounding to bf16), but every global load of
    // the epilogue is issued up front and unconditionally at a clamped address -- bias and slopes once per lane (they
    // do not depend on the pixel), the residual rows of two pixels at a time -- instead of 6 dependent round trips
    // per pixel behind the bounds branches.
    constexpr int PITCH = BN * 2 + 16;
    static_assert(std::is_same<TOut, bf16>::value, "the shared-pixel kernels write bf16");
    {
        const int c0 = nt0 * BN + wm * 64 + 16 * kg;
        const bool ok0 = c0 < a.e.cout, ok1 = c0 + 8 < a.e.cout;
        const int cs0 = ok0 ? c0 : 0, cs1 = ok1 ? c0 + 8 : 0;
        const bool prelu = a.e.act == FR_ACT_PRELU;
        const bf16* const R = reinterpret_cast<const bf16*>(a.e.residual);
        long mm[NBJ];
#pragma unroll
        for (int j = 0; j < NBJ; ++j) mm[j] = min(m0 + wn * TJ * 32 + j * 16 + c16, a.M - 1) * (long)a.e.cout;
#pragma unroll
        for (int g = 0; g < 2; ++g) {                  // couts c0 + 8 g .. + 7 of all NBJ pixels: 32 registers of loads
            const int cs = g ? cs1 : cs0;
            f32x4 bs[2], sl[2];
            u32x4 rr[NBJ];
#pragma unroll
            for (int q = 0; q < 2; ++q) bs[q] = *reinterpret_cast<const f32x4*>(a.e.bias + cs + 4 * q);
            if (prelu) {
#pragma unroll
                for (int q = 0; q < 2; ++q) sl[q] = *reinterpret_cast<const f32x4*>(a.e.slope + cs + 4 * q);
            }
            if (R) {
#pragma unroll
                for (int j = 0; j < NBJ; ++j) rr[j] = *reinterpret_cast<const u32x4*>(R + mm[j] + cs);
            }
#pragma unroll
            for (int j = 0; j < NBJ; ++j) {
                const int row = wn * TJ * 32 + j * 16 + c16;
                Vec8f v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v.v[r] = acc[2 * g][j][r] + bs[0][r]; v.v[4 + r] = acc[2 * g + 1][j][r] + bs[1][r]; }
                if (prelu) prelu8(v, sl[0], sl[1]);
                else if (a.e.act != FR_ACT_NONE) act8(v, a.e.act, nullptr);
                if (R) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v.v[2 * q] += __uint_as_float(rr[j][q] << 16);
                        v.v[2 * q + 1] += __uint_as_float(rr[j][q] & 0xFFFF0000u);
                    }
                }
                if (m0 + row < a.M && (g ? ok1 : ok0)) stage_put(smem, PITCH, row, wm * 64 + 16 * kg + 8 * g, v);
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
    const int cols = min(BN, a.e.cout - nt0 * BN);
    bf16* Y = reinterpret_cast<bf16*>(a.e.y);
    for (int c = tid; c < BM * CPR; c += 64 * NW) {
        const int row = c / CPR, ch = c - row * CPR;
        const long m = m0 + row;
        if (m < a.M && ch * 8 < cols)
            *reinterpret_cast<u32x4*>(Y + m * (long)a.e.cout + nt0 * BN + ch * 8) =
                *reinterpret_cast<const u32x4*>(smem + row * PITCH + ch * 16);
    }
}

// rows of the staged pixel range (multiple of 16), or 0 when the shape does not take the shared-pixel kernel
int conv_halo_rows(int ksize, int stride, int pad, int W, int bk) {
    if (ksize != 3 || stride != 1 || pad != 1 || bk != 32) return 0;
    static const int min_w = [] { const char* e = getenv("FR_CONV_HALO_MINW"); return e ? atoi(e) : 7; }();   // tuning (12: the 7x7 maps stay on the per-tap kernel, round 2's choice)
    if (W < min_w) return 0;
    const int rb = (128 + 2 * W + 2 + 15) / 16 * 16;
    return rb <= 576 ? rb : 0;                       // 9 taps x 4 waves x 16 rows: a ninth of the range per tap
}

// couts per block tile (the packing granularity).  Layers that run the shared-pixel kernel on maps below 12 x 12 (the 7x7
// stage of SphereFace: 49 pixels per image, 98 pixel tiles of 128 for a 256-image sub-batch) take 64-cout tiles: twice the
// blocks for a stage whose 128-cout blocks leave the chip at 1-2 waves per SIMD (sphere64 stage 4: 0.464 -> 0.448 ms per
// 256 images; with 128-cout tiles the shared-pixel kernel is SLOWER than the per-tap kernel there, 0.477).  The per-tap
// kernel keeps 128 (64: 0.665 ms).  FR_CONV_BN64_HW (tuning): map size below which that applies (default 144, 0 = never).
int conv_block_couts(int cout, int out_hw, bool halo) {
    static const int small_hw = [] { const char* s = getenv("FR_CONV_BN64_HW"); return s ? atoi(s) : 144; }();
    return cout >= 128 && !(halo && out_hw < small_hw) ? 128 : 64;
}

template <typename TIn, typename TOut>
static int launch_conv_t(const ConvArgs& a, hipStream_t st) {
    constexpr int NS = std::is_same<TIn, float>::value ? 2 : 1;
    const int bn = conv_block_couts(a.e.cout, a.Ho * a.Wo, a.halo != 0);
    dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)cdiv(a.e.cout, bn));
    if constexpr (std::is_same<TIn, bf16>::value) {
        if (a.zeros) {
            if (const int rb = a.halo ? conv_halo_rows(a.ksize, a.stride, a.pad, a.W, a.bk) : 0) {
                // pixels per block: 256 (8 waves) halves the weight bytes filled per flop; 128 (4 waves) keeps more
                // blocks for small layers.  FR_CONV_BM = 128 | 256 forces one.
                static const int env_bm = [] { const char* s = getenv("FR_CONV_BM"); return s ? atoi(s) : 0; }();
                // measured (sphere64, 256-image sub-batches): 256 wins from about one block per CU upwards
                const int bm = env_bm == 128 || env_bm == 256 ? env_bm : ((a.M + 255) / 256 * (long)grid.y >= 256 ? 256 : 128);
                const int rbm = rb + (bm - 128);
                dim3 g((unsigned)((a.M + bm - 1) / bm), grid.y);
                size_t lds = 2 * (size_t)bn * 64 + 2 * (size_t)rbm * 64 + 256;
                if (std::is_same<TOut, bf16>::value) lds = std::max(lds, (size_t)bm * (bn * 2 + 16));
#define FR_CONV_HALO_LAUNCH(TI_, TJ_, WM_, WN_)                                                                         \
    do {                                                                                                                \
        auto k = conv_halo_kernel<TOut, TI_, TJ_, WM_, WN_>;                                                            \
        FR_BIG_LDS_ONCE(k, lds);                                                                                        \
        hipLaunchKernelGGL(k, g, dim3(64 * WM_ * WN_), lds, st, a, rbm);                                                \
    } while (0)
#define FR_CONV_HALO16_LAUNCH1(TI_, TJ_, WM_, WN_, SPB_)                                                                \
    do {                                                                                                                \
        auto k = conv_halo16_kernel<bf16, TI_, TJ_, WM_, WN_, SPB_>;                                                    \
        FR_BIG_LDS_ONCE(k, lds16);                                                                                      \
        hipLaunchKernelGGL(k, g, dim3(64 * WM_ * WN_), lds16, st, a, rbm);                                              \
    } while (0)
#define FR_CONV_HALO16_LAUNCH(TI_, TJ_, WM_, WN_)                                                                       \
    do {                                                                                                                \
        if (spb == 3) FR_CONV_HALO16_LAUNCH1(TI_, TJ_, WM_, WN_, 3);                                                    \
        else if (spb == 2) FR_CONV_HALO16_LAUNCH1(TI_, TJ_, WM_, WN_, 2);                                               \
        else FR_CONV_HALO16_LAUNCH1(TI_, TJ_, WM_, WN_, 1);                                                             \
    } while (0)
                if (a.halo == 2) {                      // weights packed in the 16x16x32 row order (bf16 output only)
                    if constexpr (std::is_same<TOut, bf16>::value) {
                        // slices per barrier: the largest of a.spb (the plan's FR_CONV_SPB; default 2) .. 1 that divides the slice count,
                        // whose frame pieces fit the staging schedule and that keeps two blocks per CU (<= 80 KB)
                        const int ks = 9 * (a.Cin / 32), nw = bm == 256 ? 8 : 4;
                        int spb = std::min(std::max(a.spb > 0 ? a.spb : 2, 1), 3);
                        size_t lds16 = 0;
                        for (;; --spb) {
                            lds16 = std::max(2 * (size_t)spb * bn * 64 + 2 * (size_t)rbm * 64 + 256, (size_t)bm * (bn * 2 + 16));
                            const int pieces = (spb == 2 ? 4 : 9 / spb) * spb * nw;
                            if (spb == 1 || (ks % spb == 0 && lds16 <= 80 * 1024 && (rbm + 15) / 16 <= pieces)) break;
                        }
                        if (bm == 256) { if (bn == 128) FR_CONV_HALO16_LAUNCH(2, 2, 2, 4); else FR_CONV_HALO16_LAUNCH(2, 1, 1, 8); }
                        else { if (bn == 128) FR_CONV_HALO16_LAUNCH(2, 2, 2, 2); else FR_CONV_HALO16_LAUNCH(2, 1, 1, 4); }
                        FR_LAUNCH_CHECK("conv_halo16_kernel");
                        return FR_OK;
                    } else {
                        FR_FAIL(FR_ERR_UNSUPPORTED, "conv: the 16x16x32 shared-pixel kernel writes bf16 only");
                    }
                }
                if (bm == 256) { if (bn == 128) FR_CONV_HALO_LAUNCH(2, 2, 2, 4); else FR_CONV_HALO_LAUNCH(2, 1, 1, 8); }
                else { if (bn == 128) FR_CONV_HALO_LAUNCH(2, 2, 2, 2); else FR_CONV_HALO_LAUNCH(2, 1, 1, 4); }
#undef FR_CONV_HALO16_LAUNCH
#undef FR_CONV_HALO16_LAUNCH1
#undef FR_CONV_HALO_LAUNCH
                FR_LAUNCH_CHECK("conv_halo_kernel");
                return FR_OK;
            }
            size_t lds = 2 * (size_t)(bn + 128) * a.bk * 2;
            if (std::is_same<TOut, bf16>::value) lds = std::max(lds, (size_t)128 * (bn * 2 + 16));   // output staging tile
            if (a.bk == 64) {
                if (bn == 128) hipLaunchKernelGGL((conv_glds_kernel<TOut, 2, 2, 2, 2, 64>), grid, dim3(256), lds, st, a);
                else hipLaunchKernelGGL((conv_glds_kernel<TOut, 2, 1, 1, 4, 64>), grid, dim3(256), lds, st, a);
            } else {
                if (bn == 128) hipLaunchKernelGGL((conv_glds_kernel<TOut, 2, 2, 2, 2, 32>), grid, dim3(256), lds, st, a);
                else hipLaunchKernelGGL((conv_glds_kernel<TOut, 2, 1, 1, 4, 32>), grid, dim3(256), lds, st, a);
            }
            FR_LAUNCH_CHECK("conv_glds_kernel");
            return FR_OK;
        }
    }
    const size_t lds = 2 * (size_t)NS * (bn + 128) * CV_ROW;
    if (bn == 128) {
        auto k = conv_gemm_kernel<TIn, TOut, 2, 2, 2, 2>;
        FR_BIG_LDS_ONCE(k, lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    } else {
        auto k = conv_gemm_kernel<TIn, TOut, 2, 1, 1, 4>;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    }
    FR_LAUNCH_CHECK("conv_gemm_kernel");
    return FR_OK;
}

int launch_conv(bool in_f32, bool out_f32, const ConvArgs& a, hipStream_t st) {
    if (a.Cin % 32 || a.Cin % a.bk || (a.bk != 32 && a.bk != 64) || (in_f32 && a.bk != 32) || a.e.cout % 8 || a.M <= 0 || (a.ksize != 1 && a.ksize != 3))
        FR_FAIL(FR_ERR_ARG, "conv: Cin must be a multiple of 32, Cout of 8, ksize 1 or 3");
    if (a.M + 256 >= 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "conv: more than 2^31 output pixels");
    if (in_f32 && out_f32) return launch_conv_t<float, float>(a, st);
    if (!in_f32 && !out_f32) return launch_conv_t<bf16, bf16>(a, st);
    if (!in_f32 && out_f32) return launch_conv_t<bf16, float>(a, st);
    FR_FAIL(FR_ERR_UNSUPPORTED, "conv: fp32 input with bf16 output is not instantiated");
}
