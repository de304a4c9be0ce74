// Depthwise KxK convolution (K = 3 or 5, stride 1 or 2), NHWC, LDS-staged, fused folded-BN bias +
// PReLU / Hardswish.  Replaces nn.Conv2d(groups=C)+BatchNorm2d+activation (utils/layers.py:15-84
// as used at :143-149, models/mobilenetv2.py:36-43, models/mobilenetv3.py:70-80).
//
// A block stages a (TH*S+K-S) x (TW*S+K-S) x (8*CBV channel) input tile into LDS by LDS-DMA
// (global_load_lds, 16 B per lane; out-of-image taps read a zero page), then every thread produces
// strips of horizontally adjacent outputs for one 8-channel vector, re-using each LDS read for up
// to K/S outputs.  fp32 accumulate, fp32 weights (tap-major [K*K][C], BN folded in), one rounding at
// the store.
//
// What bounds it (rocprofv3 PMC, profiles/): HBM in principle, but with 4-6 waves
// per SIMD the VALU is 80-100 % busy, and more than half of the issued instructions used to be
// per-thread set-up (64-bit index divisions, 64-bit address arithmetic) rather than the taps.  The
// kernel therefore gets its divisors as host-computed 32-bit reciprocals, keeps image bases in
// scalar registers with 32-bit per-lane offsets, and puts (image, tile row) in blockIdx.z/.y.
#include "dw_core.h"
#ifndef FR_DWX_U2
#define FR_DWX_U2 0      // tuning: 1 = two row tiles per iteration in the expand phase (measured: mobilenetv3_large -4 %, mobilenetv2 -0.5 %: registers)
#endif
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

typedef __attribute__((address_space(3))) void* dw_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* dw_glb_ptr_t;

// n / d for n * d < 2^32 with mul = floor(2^32 / d) + 1 computed on the host (d = 1: mul unused).
struct Div32 { unsigned mul, d; };
static inline Div32 make_div32(unsigned d) { Div32 r; r.d = d; r.mul = d > 1 ? (unsigned)((1ull << 32) / d + 1ull) : 0u; return r; }
FR_DEV unsigned div32(unsigned n, Div32 v) { return v.d == 1 ? n : __umulhi(n, v.mul); }
FR_DEV Div32 make_div32_dev(unsigned d) { Div32 r; r.d = d; r.mul = d > 1 ? (unsigned)((1ull << 32) / d + 1ull) : 0u; return r; }

struct DwGeom {          // derived launch geometry (host computed)
    Div32 d_cbv, d_iw, d_ns, d_tw, d_pp, d_th;
    int IH, IW, nstrips, nitems, nunits;
    int order;           // work item -> (cv, strip, row): 0 strips before rows (it = (oy * nstrips + sx) * CBV + cv), 1 rows before strips
    int PP;              // LDS pitch of one tile pixel in 8-channel vectors (CBV, or CBV+1 when the tile is computed in the kernel)
    int PPL;             // f24 tiles staged from HBM: 16-byte slots per pixel of the lo part (two channel vectors per slot)
    int tile_bytes;      // LDS bytes of the input tile (taps, bias and slopes follow it)
    int n0;              // first image of this launch slice (blockIdx.z is relative to it)
    // XCD-aware block order (dw_block): images in groups of 8, one image per XCD
    Div32 d_per8, d_gx;
    int per_img, full;   // blocks per image; blocks covered by whole groups of 8 images (0: identity order)
};

// Which (image, tile) a block works on.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order
// and every XCD has its own L2: with the natural order the tiles of one image -- whose halos overlap --
// land on 8 different L2s and each fetches the overlap from the fabric again (PMC: up to 3x the input
// bytes on the 5x5 layers).  Here dispatch slot L goes to image 8*(L / (8*per_img)) + L % 8 and tile
// (L / 8) % per_img, so one XCD walks through one image's tiles back to back.
FR_DEV void dw_block(const DwGeom& g, int gx, int gy, int& bx, int& by, int& n) {
    bx = blockIdx.x; by = blockIdx.y; n = blockIdx.z;
    const unsigned L = blockIdx.x + (unsigned)gx * (blockIdx.y + (unsigned)gy * blockIdx.z);
    if (L < (unsigned)g.full) {
        const unsigned grp = div32(L, g.d_per8), r = L - grp * 8u * g.per_img;
        const unsigned w = r >> 3;
        n = (int)(grp * 8u + (r & 7u));
        by = (int)div32(w, g.d_gx);
        bx = (int)(w - (unsigned)by * gx);
    }
    n += g.n0;
}

static void dw_block_geom(DwGeom& g, int gx, int gy, int nslice) {
    g.per_img = gx * gy;
    const unsigned long long full = (unsigned long long)(nslice / 8) * 8ull * g.per_img;
    g.full = (full * 8ull * g.per_img < (1ull << 32) && full < (1ull << 31)) ? (int)full : 0;
    g.d_per8 = make_div32(8u * g.per_img); g.d_gx = make_div32(gx);
}

template <int S> __host__ __device__ constexpr int dw_sw() { return S == 2 ? 2 : 4; }

// Stage the input tile, the tap weights and bias/slopes of (image n, channel chunk cc, tile ty,tx)
// into LDS by LDS-DMA and wait for them (block-wide).  LDS layout: tile [IH*IW][CBV] V8<T>, then
// weights [K*K][2][CBV][4] fp32, then bias [2][CBV][4], then slopes [2][CBV][4].
template <typename T, int K, int S, bool TILE = true>
FR_DEV void dw_stage(const DwArgs& a, const DwGeom& g, unsigned char* smem, int n, int cc, int ty, int tx) {
    constexpr int UPE = std::is_same<T, float>::value ? 2 : 1;     // 16-byte units per 8-channel vector (f24: of its hi part)
    constexpr int EB = kHiBytes<T>;
    const int CBV = a.CBV, CV = a.C >> 3, IW = g.IW;
    const int cv0 = cc * CBV, ncv = min(CBV, CV - cv0);
    unsigned char* wl_b = smem + g.tile_bytes;
    const unsigned char* Xn = reinterpret_cast<const unsigned char*>(a.x) + (size_t)n * a.H * a.W * a.C * EB;
    const unsigned char* Z = reinterpret_cast<const unsigned char*>(a.zeros);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int gy0 = ty * a.TH * S - a.pad, gx0 = tx * a.TW * S - a.pad;
    for (int u0 = wave * 64; TILE && u0 < g.nunits; u0 += 256) {
        const int u = u0 + lane;
        if (u < g.nunits) {
            // bf16: unit u = (pixel, cv).  fp32: unit u = (pixel, half, cv) -- the two 16-byte halves of an 8-channel vector
            // sit in separate planes, so that lanes with consecutive cv read consecutive 16-byte slots (32-byte vectors
            // put every ds_read_b128 of the taps on half of the banks: PMC, two conflict cycles per busy cycle)
            const unsigned r = div32(u, g.d_pp), cv = u - r * g.PP;       // PP >= CBV slots per pixel (and half): the spare ones read the zero page
            const unsigned pix = UPE == 2 ? r >> 1 : r, half = UPE == 2 ? r & 1u : 0u;
            const unsigned iy = div32(pix, g.d_iw), ix = pix - iy * IW;
            const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
            const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && (int)cv < ncv;
            const int off = ((gy * a.W + gx) * a.C + (cv0 + (int)cv) * 8) * EB + (int)half * 16;
            const unsigned char* src = ok ? Xn + off : Z;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + u0 * 16), 16, 0, 0);
        }
    }
    if constexpr (kIsF24<T>) {
        // lo bytes: unit (pixel, j) = the 16 lo bytes of channel vectors 2j, 2j+1 of the chunk (cv0 is even; lo rows are
        // padded to 16 B, so a trailing half unit reads padding); behind the hi part of the tile
        const int PPL = g.PPL, nlo = g.IH * IW * PPL, LP = f24_lo_pitch(a.C);
        const unsigned char* XLn = a.x_lo + (size_t)n * a.H * a.W * LP;
        unsigned char* lo_b = smem + (size_t)g.IH * IW * g.PP * 16;
        const Div32 d_ppl = make_div32_dev(PPL);
        for (int u0 = wave * 64; TILE && u0 < nlo; u0 += 256) {
            const int u = u0 + lane;
            if (u < nlo) {
                const unsigned pix = div32(u, d_ppl), j = u - pix * PPL;
                const unsigned iy = div32(pix, g.d_iw), ix = pix - iy * IW;
                const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && (int)(2 * j) < ncv;
                const unsigned char* src = ok ? XLn + (gy * a.W + gx) * LP + (cv0 + 2 * (int)j) * 8 : Z;
                __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(lo_b + u0 * 16), 16, 0, 0);
            }
        }
    }
    // tap weights, unit (tap, half, cv) = 4 floats w[tap][(cv0+cv)*8 + 4*half ..]
    const int nw = K * K * 2 * CBV;
    for (int u0 = wave * 64; u0 < nw; u0 += 256) {
        const int u = u0 + lane;
        if (u < nw) {
            const unsigned r = div32(u, g.d_cbv), cv = u - r * CBV;
            const void* src = (int)cv < ncv ? (const void*)(a.w + (r >> 1) * a.C + (cv0 + cv) * 8 + (r & 1) * 4) : (const void*)Z;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(wl_b + u0 * 16), 16, 0, 0);
        }
    }
    // bias [2][CBV][4], then slopes [2][CBV][4]   (4*CBV <= 64 lanes)
    if (wave == 0 && lane < 4 * CBV) {
        const int nb = 2 * CBV;
        const int which = lane >= nb, r = lane - which * nb;
        const int half = r >= CBV, cv = r - half * CBV;
        const float* pb = which ? a.slope : a.bias;
        const void* src = (pb && cv < ncv) ? (const void*)(pb + (cv0 + cv) * 8 + half * 4) : (const void*)Z;
        __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(wl_b + K * K * CBV * 32), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// One work item: SW horizontally adjacent outputs (row oy, strip sx of the tile) x channel vector cv,
// bias included, activation applied.  out[o] = activated fp32 outputs.
// T: format of the LDS tile -- bf16 vectors, fp32 half-planes, or f24 as staged from HBM (hi part [pixel][PP] 16-byte
// slots, lo part `tile_lo` [pixel][PPL] 16-byte slots of two vectors each).  R24: the activated result is rounded to f24
// (f24 plans: it is stored as f24, or handed to the fused 1x1 conv as the unfused plan would re-load it).
// ACT: the activation at compile time, or -1: a.act at run time (the fused kernels)
template <typename T, int K, int S, bool R24 = false, int ACT = -1>
FR_DEV void dw_item(const DwArgs& a, const void* tile, const float* wl, const float* bl, int IW, int PP,
                    unsigned cv, unsigned oy, unsigned sx, Vec8f (&out)[dw_sw<S>()], const unsigned char* tile_lo = nullptr, int PPL = 0) {
    constexpr int SW = dw_sw<S>();
    constexpr int IWS = (SW - 1) * S + K;
    const int CBV = a.CBV;
    float acc[SW][8];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + (0 * CBV + cv) * 4);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + (1 * CBV + cv) * 4);
#pragma unroll
        for (int o = 0; o < SW; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[o][j] = b0[j]; acc[o][4 + j] = b1[j]; }
    }
    // 16-byte slots: bf16 (pixel * PP + cv); fp32 ((pixel * 2 + half) * PP + cv), see dw_stage
    constexpr int HP = std::is_same<T, float>::value ? 2 : 1;
    const unsigned char* row = reinterpret_cast<const unsigned char*>(tile) + ((((oy * S) * IW + sx * SW * S) * HP) * PP + cv) * 16;
    const unsigned char* rowl = kIsF24<T> ? tile_lo + ((oy * S) * IW + sx * SW * S) * PPL * 16 + cv * 8 : nullptr;
    (void)rowl;
    const float* wrow = wl + cv * 4;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        float wr[K][8];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + (kx * 2 + 0) * CBV * 4);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrow + (kx * 2 + 1) * CBV * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { wr[kx][j] = w0[j]; wr[kx][4 + j] = w1[j]; }
        }
#pragma unroll
        for (int ix = 0; ix < IWS; ++ix) {
            Vec8f v;
            if constexpr (HP == 2) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + (ix * 2) * PP * 16), v1 = *reinterpret_cast<const f32x4*>(row + (ix * 2 + 1) * PP * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v.v[j] = v0[j]; v.v[4 + j] = v1[j]; }
            } else if constexpr (kIsF24<T>) {
                const u32x4 hh = *reinterpret_cast<const u32x4*>(row + ix * PP * 16);
                const unsigned long long ll = *reinterpret_cast<const unsigned long long*>(rowl + ix * PPL * 16);
                v = f24_decode(hh, (unsigned)ll, (unsigned)(ll >> 32));
            } else {
                v = load8(reinterpret_cast<const bf16*>(row + ix * PP * 16));
            }
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                const int kx = ix - o * S;
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] = fmaf(v.v[j], wr[kx][j], acc[o][j]);
                }
            }
        }
        row += IW * HP * PP * 16;
        if constexpr (kIsF24<T>) rowl += IW * PPL * 16;
        wrow += K * 2 * CBV * 4;
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (ACT == FR_ACT_PRELU || (ACT < 0 && a.act == FR_ACT_PRELU)) {
        s0 = *reinterpret_cast<const f32x4*>(bl + CBV * 8 + (0 * CBV + cv) * 4);
        s1 = *reinterpret_cast<const f32x4*>(bl + CBV * 8 + (1 * CBV + cv) * 4);
    }
#pragma unroll
    for (int o = 0; o < SW; ++o) {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[o].v[j] = acc[o][j];
        if constexpr (ACT == FR_ACT_PRELU) prelu8(out[o], s0, s1);
        else if constexpr (ACT >= 0) act8(out[o], ACT, nullptr);           // (a constant: the other forms fold away)
        else if (a.act == FR_ACT_PRELU) prelu8(out[o], s0, s1);
        else act8(out[o], a.act, nullptr);
        if constexpr (R24) {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[o].v[j] = f24_round(out[o].v[j]);
        }
    }
}

template <typename T, int K, int S, int ACT>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a, DwGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SW = dw_sw<S>();                  // outputs per work item (see dw_pick_tile)
    const int CBV = a.CBV, CV = a.C >> 3;
    int bx, ty, n;
    dw_block(g, a.tiles_x * a.cchunks, a.tiles_y, bx, ty, n);
    const int cc = bx % a.cchunks, tx = bx / a.cchunks;
    const int cv0 = cc * CBV, ncv = min(CBV, CV - cv0);
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int IW = g.IW;

    const void* tile = smem;                                             // [IH*IW][CBV]
    const unsigned char* tile_lo = smem + (size_t)g.IH * IW * g.PP * 16; // f24 only
    const float* wl = reinterpret_cast<const float*>(smem + g.tile_bytes);
    const float* bl = wl + K * K * CBV * 8;
    constexpr int EB = kHiBytes<T>;
    unsigned char* Yn = reinterpret_cast<unsigned char*>(a.y) + (size_t)n * a.Ho * a.Wo * a.C * EB;
    const int LPo = f24_lo_pitch(a.C);
    unsigned char* YLn = kIsF24<T> ? a.y_lo + (size_t)n * a.Ho * a.Wo * LPo : nullptr;
    (void)YLn;
    dw_stage<T, K, S>(a, g, smem, n, cc, ty, tx);

    // psum receives the sum of the valid outputs (squeeze-excite pooling) when a.pool is set
    auto item = [&](int it, float* psum) {
        const unsigned r = div32(it, g.d_cbv), cv = it - r * CBV;
        unsigned oy, sx;
        if (g.order) { sx = div32(r, g.d_th); oy = r - sx * a.TH; }
        else { oy = div32(r, g.d_ns); sx = r - oy * g.nstrips; }
        if ((int)cv >= ncv) return;
        Vec8f out[SW];
        dw_item<T, K, S, kIsF24<T>, ACT>(a, tile, wl, bl, IW, g.PP, cv, oy, sx, out, tile_lo, g.PPL);
        const int gy = oy0 + (int)oy;
        if (gy < a.Ho) {
            const int gx0o = ox0 + (int)sx * SW;
            const int yoff = ((gy * a.Wo + gx0o) * a.C + (cv0 + (int)cv) * 8) * EB;
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                if (gx0o + o < a.Wo) {
                    if (psum) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) psum[j] += out[o].v[j];
                    }
                    if constexpr (kIsF24<T>)
                        f24_store8(reinterpret_cast<bf16*>(Yn + yoff + o * a.C * EB), YLn + (gy * a.Wo + gx0o + o) * LPo + (cv0 + (int)cv) * 8, out[o]);
                    else store8(reinterpret_cast<T*>(Yn + yoff + o * a.C * EB), out[o]);
                }
            }
        }
    };

    if (!a.pool) {
        for (int it = threadIdx.x; it < g.nitems; it += 256) item(it, nullptr);
        return;
    }
    // pooled form (nitems <= 512, checked by the launcher): per-item sums -> LDS -> fixed-order
    // reduction per channel -> pool[n][tile][c].  Deterministic: no atomics.
    float ps0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((int)threadIdx.x < g.nitems) item(threadIdx.x, ps0);
    if ((int)threadIdx.x + 256 < g.nitems) item(threadIdx.x + 256, ps1);
    __syncthreads();                                   // every wave is done reading the input tile
    float* sums = reinterpret_cast<float*>(smem);      // [nitems][8], aliases the tile
    if ((int)threadIdx.x < g.nitems) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sums[threadIdx.x * 8 + j] = ps0[j];
    }
    if ((int)threadIdx.x + 256 < g.nitems) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sums[(threadIdx.x + 256) * 8 + j] = ps1[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < ncv * 8) {
        const int cv = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int per = g.nitems / CBV;                // items per channel vector: it = r * CBV + cv
        float t = 0.f;
        for (int r = 0; r < per; ++r) t += sums[(r * CBV + cv) * 8 + j];
        a.pool[((size_t)n * a.pool_tiles + (size_t)ty * a.tiles_x + tx) * a.C + (cv0 + cv) * 8 + j] = t;
    }
}

static size_t dw_tile_bytes(int fmt, int IH, int IW, int PP) {      // input tile as staged from HBM
    if (fmt == FMT_F24) return (size_t)IH * IW * (PP + (PP + 1) / 2) * 16;
    return (size_t)IH * IW * PP * (fmt == FMT_F32 ? 32 : 16);
}
static size_t dw_lds_bytes(int fmt, int k, int s, int TH, int TW, int CBV) {
    const int IH = (TH - 1) * s + k, IW = (TW - 1) * s + k;
    return dw_tile_bytes(fmt, IH, IW, CBV) + (size_t)k * k * CBV * 32 + (size_t)CBV * 64;
}

// Pick the output tile and channel chunk: keep LDS <= 40 KB (>= 3 blocks per CU), prefer all
// channels in one block (long contiguous NHWC runs), then the least padded-work + halo overhead.
// Work items are strips of 4 outputs at stride 1 (each LDS read serves up to K outputs) and of 2 at
// stride 2 (2+K columns for 2 outputs is within 10 % of a strip of 4, and the LDS-limited stride-2
// tiles get twice the work items).
// Measured-best tiles for the layer shapes of the four MobileNets (bf16, 512-image launches, MI355X):
// tools/dw_probe.py sweeps (TH, TW, CBV) per shape, tools/make_dw_table.py keeps the entries that beat the cost
// model below by at least 3 % (profiles/tuning/).  Shapes that are not listed use the cost model.
struct DwTileEntry { int C, Ho, k, s, TH, TW, CBV; };
static const DwTileEntry kDwTiles[] = {
#include "dw_tiles.inc"
};
static const DwTileEntry kDwTilesF32[] = {        // fp32 plans (tools/dw_probe.py <model> 512 f32)
#include "dw_tiles_f32.inc"
};

static void dw_pick_tile(int fmt, int k, int s, DwArgs& a) {
    const int CV = a.C / 8;
    const int sw = s == 2 ? 2 : 4;
    static const bool use_table = !(getenv("FR_DW_TABLE") && atoi(getenv("FR_DW_TABLE")) == 0);     // 0: cost model only (tools/dw_probe.py measures against it)
    // f24 tiles take 24 B per channel vector: the bf16 winners where they still fit, else the fp32 ones.  A channel chunk
    // must start on an even vector (the lo part is staged in 16-byte units of two vectors).
    for (int pass = 0; pass < 2 && a.Ho == a.Wo && use_table; ++pass) {
        const bool f32tab = fmt == FMT_F32 || (fmt == FMT_F24 && pass == 1);
        if (pass == 1 && fmt != FMT_F24) break;
        const DwTileEntry* tb = f32tab ? kDwTilesF32 : kDwTiles;
        const size_t ne = f32tab ? sizeof(kDwTilesF32) / sizeof(DwTileEntry) : sizeof(kDwTiles) / sizeof(DwTileEntry);
        for (size_t i = 0; i < ne; ++i) {
            const DwTileEntry& e = tb[i];
            if (e.C == a.C && e.Ho == a.Ho && e.k == k && e.s == s && e.TW % sw == 0 && e.CBV <= CV &&
                (fmt != FMT_F24 || e.CBV % 2 == 0 || e.CBV >= CV) &&
                e.TH * (e.TW / sw) * e.CBV <= 512 && dw_lds_bytes(fmt, k, s, e.TH, e.TW, e.CBV) <= 60 * 1024) {
                a.TH = e.TH; a.TW = e.TW; a.CBV = e.CBV;
                a.tiles_y = cdiv(a.Ho, e.TH); a.tiles_x = cdiv(a.Wo, e.TW); a.cchunks = cdiv(CV, e.CBV);
                return;
            }
        }
    }
    double best = 1e30;
    const int ths[] = {2, 4, 7, 8, 14, 16};
    const int tws[] = {4, 8, 12, 16, 28, 32, 56};
    for (int th : ths) for (int tw : tws) {
        if (th > a.Ho && th != ths[0]) { if (th - a.Ho >= 2) continue; }
        if (tw - a.Wo >= 4) continue;
        for (int cbv = std::min(CV, 16); cbv >= 1; cbv = (cbv > 1 ? (cbv + 1) / 2 : 0)) {
            if (fmt == FMT_F24 && cbv < CV && (cbv & 1)) { if (cbv > 1) ++cbv; else continue; }     // even chunks (see above); the loop halves it again
            if (fmt == FMT_F24 && cbv > 16) continue;
            if (dw_lds_bytes(fmt, k, s, th, tw, cbv) > 40 * 1024) continue;
            const int tyc = cdiv(a.Ho, th), txc = cdiv(a.Wo, tw), ccs = cdiv(CV, cbv);
            const int items = th * (tw / sw) * cbv;
            if (items > 512) continue;                      // two passes at most (the pooled form keeps per-pass sums in registers)
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
static int launch_dw_t(int k, int s, const DwArgs& a, DwGeom g, size_t lds, hipStream_t st) {
    for (int n0 = 0; n0 < a.N; n0 += 65535) {          // blockIdx.z carries the image
        g.n0 = n0;
        dim3 grid((unsigned)(a.tiles_x * a.cchunks), (unsigned)a.tiles_y, (unsigned)std::min(65535, a.N - n0));
        dw_block_geom(g, (int)grid.x, (int)grid.y, (int)grid.z);
        // the activation is a template parameter: per output the run-time form walks a ladder of uniform branches
#define FR_DW_CASE(KK, SS) if (k == KK && s == SS) { switch (a.act) { \
            case FR_ACT_NONE: hipLaunchKernelGGL((dw_kernel<T, KK, SS, FR_ACT_NONE>), grid, dim3(256), lds, st, a, g); break; \
            case FR_ACT_PRELU: hipLaunchKernelGGL((dw_kernel<T, KK, SS, FR_ACT_PRELU>), grid, dim3(256), lds, st, a, g); break; \
            case FR_ACT_HSWISH: hipLaunchKernelGGL((dw_kernel<T, KK, SS, FR_ACT_HSWISH>), grid, dim3(256), lds, st, a, g); break; \
            case FR_ACT_RELU: hipLaunchKernelGGL((dw_kernel<T, KK, SS, FR_ACT_RELU>), grid, dim3(256), lds, st, a, g); break; \
            case FR_ACT_HSIGMOID: hipLaunchKernelGGL((dw_kernel<T, KK, SS, FR_ACT_HSIGMOID>), grid, dim3(256), lds, st, a, g); break; \
            default: FR_FAIL(FR_ERR_ARG, "depthwise: unknown activation"); } }
        FR_DW_CASE(3, 1) else FR_DW_CASE(3, 2) else FR_DW_CASE(5, 1) else FR_DW_CASE(5, 2)
        else FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: only k in {3,5}, stride in {1,2}");
#undef FR_DW_CASE
        FR_LAUNCH_CHECK("dw_kernel");
    }
    return FR_OK;
}

// ---- LDS bank model of dw_item's tile reads (host) ------------------------------------------------------------------
// A ds_read_b128 is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31}
// and the same + 32), one cycle per group when its lanes fall on 16 different 16-byte slots mod 16.  With the item ->
// lane order (cv, strip, row) and a pixel pitch of CBV slots, lanes of different strips land on the same slots mod 16:
// the counters show 3.4-6.4 conflict cycles per LDS instruction in the fp32 depthwise kernels (round 3,
// profiles/r03_pmc_sq_mnv3large_b1024_f32.txt), this model 8-16 cycles per read where 4 is conflict-free.  The launcher
// evaluates the model for both item orders and 0..3 spare slots per pixel / spare pixels per tile row and takes the
// cheapest layout that keeps the blocks per CU.  hp: 16-byte halves per vector (fp32: 2).
static double dw_bank_cycles(int K, int S, int TH, int TW, int CBV, int PP, int hp, int IW, int order) {
    static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                   {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int SW = S == 2 ? 2 : 4, ns = TW / SW, nitems = TH * ns * CBV, IWS = (SW - 1) * S + K;
    long total = 0; int reads = 0;
    for (int wave = 0; wave < 2 && wave * 64 < nitems; ++wave) {
        int base[64];
        for (int l = 0; l < 64; ++l) {
            const int it = wave * 64 + l;
            if (it >= nitems) { base[l] = -1; continue; }
            const int r = it / CBV, cv = it % CBV;
            const int oy = order ? r % TH : r / ns, sx = order ? r / TH : r % ns;
            base[l] = ((oy * S) * IW + sx * SW * S) * hp * PP + cv;
        }
        for (int ky = 0; ky < K; ++ky)
            for (int ix = 0; ix < IWS; ++ix)
                for (int h = 0; h < hp; ++h) {
                    const int off = ((ky * IW + ix) * hp + h) * PP;
                    for (int g = 0; g < 4; ++g) {
                        int seen[16][16], cnt[16] = {0}, worst = 1;
                        for (int q = 0; q < 16; ++q) {
                            const int b = base[grp[g][q]];
                            if (b < 0) continue;
                            const int slot = b + off, m = slot & 15;
                            bool dup = false;
                            for (int e = 0; e < cnt[m]; ++e) dup = dup || seen[m][e] == slot;
                            if (!dup) { seen[m][cnt[m]++] = slot; worst = std::max(worst, cnt[m]); }
                        }
                        total += worst;
                    }
                    ++reads;
                }
    }
    return reads ? (double)total / reads : 4.0;
}

// Picks g.order, g.PP, g.IW for a tile (TH, TW, CBV) already chosen: least modelled cycles, then least LDS.
// FR_DW_BANK=0: the round-2 layout (strips before rows, no spare slots).
static void dw_pick_layout(int fmt, int k, int s, const DwArgs& a, DwGeom& g) {
    static const bool on = !(getenv("FR_DW_BANK") && atoi(getenv("FR_DW_BANK")) == 0);
    const int hp = fmt == FMT_F32 ? 2 : 1, IW0 = (a.TW - 1) * s + k, IH = (a.TH - 1) * s + k;
    g.order = 0; g.PP = a.CBV; g.IW = IW0;
    if (!on) return;
    // the model costs tens of microseconds per call: evaluated once per (format, kernel, stride, tile) and remembered (a batch-1
    // forward is bound by the host's enqueue time: 0.46 -> 0.76 ms when every launch ran the model)
    struct Pick { int order, PP, IW; };
    static std::mutex mu;
    static std::map<std::array<int, 6>, Pick> cache;
    const std::array<int, 6> key = {fmt, k, s, a.TH, a.TW, a.CBV};
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it != cache.end()) { g.order = it->second.order; g.PP = it->second.PP; g.IW = it->second.IW; return; }
    }
    const size_t rest = (size_t)k * k * a.CBV * 32 + (size_t)a.CBV * 64;
    const size_t lds0 = dw_tile_bytes(fmt, IH, IW0, a.CBV) + rest;
    const int bpc0 = (int)std::min<size_t>(8, 160 * 1024 / lds0);
    double best = dw_bank_cycles(k, s, a.TH, a.TW, a.CBV, a.CBV, hp, IW0, 0) - 1e-9;
    size_t best_lds = lds0;
    for (int order = 0; order < 2; ++order)
        for (int pp = a.CBV; pp <= a.CBV + 3; ++pp)
            for (int iw = IW0; iw <= IW0 + 3; ++iw) {
                if (fmt == FMT_F24 && pp != a.CBV) continue;              // (the lo part of an f24 tile is laid out by pairs of vectors)
                const size_t lds = dw_tile_bytes(fmt, IH, iw, pp) + rest;
                if (lds > 64 * 1024 || (int)std::min<size_t>(8, 160 * 1024 / lds) < bpc0) continue;
                const double c = dw_bank_cycles(k, s, a.TH, a.TW, a.CBV, pp, hp, iw, order);
                if (c < best - 0.25 || (c < best + 1e-9 && lds < best_lds)) { best = c; best_lds = lds; g.order = order; g.PP = pp; g.IW = iw; }
            }
    {
        std::lock_guard<std::mutex> lock(mu);
        cache[key] = Pick{g.order, g.PP, g.IW};
    }
    if (getenv("FR_DW_BANK_DEBUG"))
        fprintf(stderr, "dw layout fmt %d C %d Ho %d k %d s %d tile %dx%d cbv %d: cycles %.2f -> %.2f order %d PP %d IW %d (+%d) lds %zu -> %zu\n", fmt, a.C, a.Ho, k, s,
                a.TH, a.TW, a.CBV, dw_bank_cycles(k, s, a.TH, a.TW, a.CBV, a.CBV, hp, IW0, 0), best, g.order, g.PP, g.IW, g.IW - IW0, lds0, best_lds);
}

int dw_tiles_per_image(int fmt, int k, int stride, int C, int Ho, int Wo) {
    DwArgs a{};
    a.C = C; a.Ho = Ho; a.Wo = Wo; a.TH = 0;
    dw_pick_tile(fmt, k, stride, a);
    return a.TH ? a.tiles_y * a.tiles_x : 0;
}

int launch_dw(int fmt, int k, int stride, DwArgs a, hipStream_t st) {
    const bool f32 = fmt == FMT_F32;
    if (fmt == FMT_F24 && (!a.x_lo || !a.y_lo)) FR_FAIL(FR_ERR_ARG, "depthwise: f24 tensors without their lo planes");
    if (a.C % 8 || a.N <= 0) FR_FAIL(FR_ERR_ARG, "depthwise: C must be a multiple of 8");
    if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: only k in {3,5}, stride in {1,2}");
    if (!a.zeros) FR_FAIL(FR_ERR_ARG, "depthwise: no zero page");
    if ((long)a.H * a.W * a.C * 4 >= 0x7FFFFFFFL) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: image above 2 GB");
    a.TH = 0;
    dw_pick_tile(fmt, k, stride, a);
    if (a.TH == 0) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: no tile fits LDS");
    if (const char* ov = getenv("FR_DW_TILE")) {           // tuning override: "C,Ho,k,s,TH,TW,CBV" for the matching layers
        int c, ho, kk, ss, th, tw, cbv;
        if (sscanf(ov, "%d,%d,%d,%d,%d,%d,%d", &c, &ho, &kk, &ss, &th, &tw, &cbv) == 7 && c == a.C && ho == a.Ho && kk == k &&
            ss == stride && th > 0 && tw > 0 && tw % 4 == 0 && cbv > 0 && cbv <= 16) {
            a.TH = th; a.TW = tw; a.CBV = std::min(cbv, a.C / 8);
            a.tiles_y = cdiv(a.Ho, th); a.tiles_x = cdiv(a.Wo, tw); a.cchunks = cdiv(a.C / 8, a.CBV);
        }
    }
    if (fmt == FMT_F24 && a.cchunks > 1 && (a.CBV & 1)) FR_FAIL(FR_ERR_ARG, "depthwise (f24): channel chunks must hold an even number of 8-channel vectors");
    if (a.CBV > 16) FR_FAIL(FR_ERR_ARG, "depthwise: channel chunk above 128 channels");
    if (a.tiles_y > 65535 || (long)a.tiles_x * a.cchunks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "depthwise: grid too large");
    if (dw_lds_bytes(fmt, k, stride, a.TH, a.TW, a.CBV) > 64 * 1024) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: tile above 64 KB of LDS");
    if (a.pool && a.pool_tiles != a.tiles_y * a.tiles_x) FR_FAIL(FR_ERR_ARG, "depthwise: pool buffer sized for another tiling");
    DwGeom g{};
    const int sw = stride == 2 ? 2 : 4;
    g.IH = (a.TH - 1) * stride + k;
    dw_pick_layout(fmt, k, stride, a, g);              // g.order, g.PP (>= CBV), g.IW (>= (TW - 1) * stride + k)
    g.nstrips = a.TW / sw;
    g.nitems = a.TH * g.nstrips * a.CBV;
    g.nunits = g.IH * g.IW * g.PP * (f32 ? 2 : 1);
    g.PPL = (a.CBV + 1) / 2;
    g.tile_bytes = (int)dw_tile_bytes(fmt, g.IH, g.IW, g.PP);
    const size_t lds = (size_t)g.tile_bytes + (size_t)k * k * a.CBV * 32 + (size_t)a.CBV * 64;
    g.d_cbv = make_div32(a.CBV); g.d_iw = make_div32(g.IW); g.d_ns = make_div32(g.nstrips);
    g.d_pp = make_div32(g.PP); g.d_th = make_div32(a.TH);
    if (a.pool && g.nitems > 512) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: pooled form needs <= 512 work items per tile");
    if (a.pool && (size_t)g.nitems * 32 > (size_t)g.tile_bytes) FR_FAIL(FR_ERR_UNSUPPORTED, "depthwise: pooled sums do not fit the input tile");
    if (fmt == FMT_F24) return launch_dw_t<f24>(k, stride, a, g, lds, st);
    return f32 ? launch_dw_t<float>(k, stride, a, g, lds, st) : launch_dw_t<bf16>(k, stride, a, g, lds, st);
}


// ---- depthwise + following 1x1 conv in one kernel ---------------------------------------------------
// For the early blocks (all channels of a tile fit one block: C <= 128) the depthwise output never
// goes to HBM: the block keeps its TH x TW x C result in LDS (over the input tile, which is dead by
// then) and multiplies it by the 1x1 weights on MFMA right away -- the pointwise launch and one
// write + one read of the expanded activation disappear.  Arithmetic is identical to the separate
// kernels (same rounding points, same k order and pass order in the MFMAs), so the fused and the
// unfused plan give bit-identical results.
// T = bf16: activations bf16, one MFMA pass.  T = float (fp32 plans): activations fp32 in HBM and in the
// depthwise tile; the depthwise result is split into bf16 hi + lo planes when it is written to LDS and every
// product is the three passes (w_hi x_hi, w_hi x_lo, w_lo x_hi) of the fp32 1x1 kernels.
struct DwpArgs {
    DwArgs dw;            // depthwise part (dw.y unused; CBV == C/8, cchunks == 1)
    const bf16* wp;       // 1x1 weights packed [NT][KS][NS][64][8]
    EpiArgs e;            // 1x1 epilogue: bias, slopes, residual, y, cout, act
    int KS, NT;
    int npx, npx_pad;     // TH*TW and its round-up to 32
    int pitch;            // bytes per row of one plane of the LDS depthwise result
    int off_lo;           // fp32 plans: byte offset of the lo plane
    int off_wp, off_stage;
    int off_pb;           // bias then slopes of the 1x1 conv, [NT*32] floats each
    // optional expand 1x1 conv in front (EXPAND form): dw.x is then unused
    const void* xin;      // block input [N,H,W,Cin]
    const unsigned char* xin_lo;   // f24: its lo plane
    const bf16* we;       // expand weights packed [NTe][KSe][NS][64][8]
    const float* be;      // expand bias (padded to 32)
    const float* se;      // expand PReLU slopes or NULL
    int act_e, Cin, KSe, NTe;
    int off_we, off_in;
    Div32 d_cin8;
    // STEM form: the layer in front of the depthwise conv is the network's 3x3 stem conv (Cin = 3, stride 1, pad 1) and is
    // computed here from the fp32 NCHW input: K = 27 (padded to 32, k = c*9 + ky*3 + kx as stem_kernel), KSe = 2
    const float* xs;      // network input [N,3,H,W] (H, W = the depthwise input size)
    int spitch;           // row pitch of the per-wave output staging tiles (bf16 / fp32 plans; f24 keeps the compile-time pitch)
    int off_dwo;          // chunked form (dwc_kernel): LDS byte offset of the depthwise result of one 32-channel chunk
    int off_res;          // > 0: LDS byte offset of the stem output at the tile's own pixels, [TH*TW][C] as stored in HBM
                          // by the unfused plan: the residual of the 1x1 conv (which then never reads one from HBM)
};

// Expand phase of the fused block: out[pixel][C] = act(W_e . in[pixel][Cin] + b) for every pixel of the staged
// input tile, written in the depthwise tile layout (pitch g.PP vectors per pixel), zero outside the
// image.  The kernel is bound by instruction issue (one wave per SIMD and block, three blocks per CU), so
// everything uniform is decided at compile time: k-steps, channel tiles, the activation (ACT < 0: run-time
// act8) and whether the tile touches the image border at all (MASK).  Weights and bias live in registers.
// tbase: first 32-channel tile of the expand conv this call computes (NTE tiles from there; the chunked kernel calls it with
// NTE = 1 per chunk): weights, bias and slopes are taken from tile tbase + t, the result goes to slots 4 t .. of the pixel.
template <typename T, int KSE, int NTE, int ACT, bool MASK, bool STEM = false, int U = 1>
FR_DEV void dwp_expand(const DwpArgs& x, const DwGeom& g, unsigned char* smem, const float* bel, const float* sel,
                       int gy0, int gx0, int wave, int lane, int tbase = 0) {
    constexpr bool F32 = kWide<T>;                    // fp32-class arithmetic; the expanded tile is fp32 half-planes
    constexpr bool F24 = kIsF24<T>;
    constexpr int NS = F32 ? 2 : 1;
    const DwArgs& a = x.dw;
    const int IW = g.IW, npin = g.IH * IW;
    const int pe = lane & 31, he = lane >> 5;
    const unsigned char* inl = smem + x.off_in;
    const int rowb = x.Cin * kHiBytes<T>;
    const int LPin = f24_lo_pitch(x.Cin);
    const unsigned char* inl_lo = inl + (size_t)npin * rowb;          // f24: staged lo rows [pixel][LPin]
    (void)LPin; (void)inl_lo;
    bf16x8 wv[NTE][KSE][NS];
    f32x4 bb[NTE][4], ss[NTE][4];                     // bias and (PReLU) slopes of this lane's 16 channels per tile
    {
        const bf16x8* wef = reinterpret_cast<const bf16x8*>(smem + x.off_we) + lane;
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
#pragma unroll
            for (int sk = 0; sk < KSE; ++sk)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) wv[t][sk][sp] = wef[(((tbase + t) * KSE + sk) * NS + sp) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bb[t][q] = *reinterpret_cast<const f32x4*>(bel + 32 * (tbase + t) + 16 * he + 4 * q);
                if constexpr (ACT == FR_ACT_PRELU) ss[t][q] = *reinterpret_cast<const f32x4*>(sel + 32 * (tbase + t) + 16 * he + 4 * q);
            }
        }
    }
    // STEM: LDS offsets (in bytes) of the 16 im2col taps this lane feeds, k = 16 sk + 8 he + j = c*9 + ky*3 + kx, inside the
    // staged fp32 patch [3][IH+2][IW+2] (stem_kernel's order, the natural one of weight.reshape(cout, 27))
    const int PW = IW + 2;
    int koff[STEM ? KSE : 1][8];
    if constexpr (STEM) {
#pragma unroll
        for (int sk = 0; sk < KSE; ++sk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * sk + 8 * he + j;
                const int c = k / 9, r9 = k - 9 * c;
                const int ky = r9 / 3, kx = r9 - 3 * ky;
                koff[sk][j] = k < 27 ? ((c * (g.IH + 2) + ky) * PW + kx) * 4 : 0;     // k >= 27 (sk = 1, he = 1, j >= 3): loaded anyway, replaced by 0
            }
    }
    (void)koff; (void)PW;
    // U row tiles (32 pixels each) per iteration and wave: their fragment loads, MFMA chains and epilogues are independent, so
    // the scheduler overlaps one tile's LDS and MFMA latencies with the other's instructions (one to two waves per SIMD here)
    for (int rt0 = wave; rt0 * 32 < npin; rt0 += 4 * U) {
        int pix[U];
        bool inimg[U];
        unsigned iy[U], ix[U];
        bf16x8 xf[U][KSE][NS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pix[u] = (rt0 + 4 * u) * 32 + pe;
            const int pc = pix[u] < npin ? pix[u] : npin - 1;
            inimg[u] = true; iy[u] = 0; ix[u] = 0;
            if constexpr (MASK || STEM) { iy[u] = div32(pc, g.d_iw); ix[u] = pc - iy[u] * IW; }
            if constexpr (MASK) {
                const int gy = gy0 + (int)iy[u], gx = gx0 + (int)ix[u];
                inimg[u] = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            }
#pragma unroll
            for (int sk = 0; sk < KSE; ++sk) {
                const int kk = 16 * sk + 8 * he;
                if constexpr (STEM) {
                    const unsigned char* patch = inl + (iy[u] * PW + ix[u]) * 4;
                    // unconditional loads, then a select: `k < 27 ? patch[..] : 0` makes the compiler branch around every single load
                    // (exec mask, ds_read, lgkmcnt(0) per element: 16 dependent LDS round trips per 32 pixels)
                    Vec8f v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.v[j] = *reinterpret_cast<const float*>(patch + koff[sk][j]);
                    if (sk == 1) {
#pragma unroll
                        for (int j = 3; j < 8; ++j) v.v[j] = he ? 0.f : v.v[j];
                    }
                    if constexpr (F32) split8(v, xf[u][sk][0], xf[u][sk][1]);
                    else xf[u][sk][0] = to_frag(v);
                } else {
                    // loaded unconditionally at a clamped channel offset, then zeroed where the half k-step lies beyond Cin: a load
                    // behind `kk < Cin ? .. : 0` is an exec-masked branch and a dependent LDS round trip of its own per k-step
                    const int kc = kk < x.Cin ? kk : 0;
                    const bool dead = kk >= x.Cin;
                    const bf16x8 z8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    if constexpr (F24) {
                        const u32x4 hh = *reinterpret_cast<const u32x4*>(inl + pc * rowb + kc * 2);
                        const unsigned long long ll = *reinterpret_cast<const unsigned long long*>(inl_lo + pc * LPin + kc);
                        f24_frags(hh, (unsigned)ll, (unsigned)(ll >> 32), xf[u][sk][0], xf[u][sk][1]);
                    } else if constexpr (F32) {
                        split8(load8(reinterpret_cast<const float*>(inl + pc * rowb) + kc), xf[u][sk][0], xf[u][sk][1]);
                    } else {
                        xf[u][sk][0] = *reinterpret_cast<const bf16x8*>(inl + pc * rowb + kc * 2);
                    }
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) xf[u][sk][sp] = dead ? z8 : xf[u][sk][sp];
                }
            }
        }
        // all channel tiles' MFMAs first, then their epilogues: an epilogue right behind its own (single) MFMA waits out
        // the MFMA's latency in s_nops, once per tile, with one wave per SIMD and block to hide it
        f32x16 accs[U][NTE];
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[u][t][r] = 0.f;
#pragma unroll
            for (int sk = 0; sk < KSE; ++sk) {
#pragma unroll
                for (int u = 0; u < U; ++u) accs[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[t][sk][0], xf[u][sk][0], accs[u][t], 0, 0, 0);
                if constexpr (F32) {
#pragma unroll
                    for (int u = 0; u < U; ++u) accs[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[t][sk][0], xf[u][sk][1], accs[u][t], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < U; ++u) accs[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[t][sk][1], xf[u][sk][0], accs[u][t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
        // 8-channel group cg of this pixel sits at (pix * PP + cg) vectors of sizeof(V8<T>) bytes
        // (fp32: (pixel * 2 + half) * PP + cg in 16-byte slots, the layout dw_item reads)
        unsigned char* dst = smem + ((size_t)pix[u] * g.PP * (F32 ? 2 : 1) + 2 * he) * 16;
#pragma unroll
        for (int t = 0; t < NTE; ++t) {
            const f32x16& acc = accs[u][t];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int c = 32 * (tbase + t) + 16 * he + 8 * gg;
                Vec8f v;
#pragma unroll
                for (int i = 0; i < 4; ++i) { v.v[i] = acc[8 * gg + i] + bb[t][2 * gg][i]; v.v[4 + i] = acc[8 * gg + 4 + i] + bb[t][2 * gg + 1][i]; }
                if constexpr (ACT == FR_ACT_RELU) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v.v[i] = relu1(v.v[i]);
                } else if constexpr (ACT == FR_ACT_PRELU) {
                    // slopes from registers: a pointer here makes the compiler branch and load per element
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v.v[i] = v.v[i] >= 0.f ? v.v[i] : v.v[i] * ss[t][2 * gg][i];
                        v.v[4 + i] = v.v[4 + i] >= 0.f ? v.v[4 + i] : v.v[4 + i] * ss[t][2 * gg + 1][i];
                    }
                } else if constexpr (ACT == FR_ACT_HSWISH) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.v[j] = v.v[j] * fminf(fmaxf(fmaf(v.v[j], 1.f / 6.f, 0.5f), 0.f), 1.f);
                } else {
                    act8(v, x.act_e, sel ? sel + c : nullptr);
                }
                const bool live = pix[u] < npin && ((tbase == 0 && t < NTE - 1) || c < a.C);
                if constexpr (STEM) {
                    // the tile's own pixels once more, as the unfused plan stores them: the residual of the 1x1 conv
                    const unsigned oy = iy[u] - (unsigned)a.pad, ox = ix[u] - (unsigned)a.pad;
                    if (x.off_res && live && oy < (unsigned)a.TH && ox < (unsigned)a.TW) {
                        unsigned char* rd = smem + x.off_res + ((size_t)(oy * a.TW + ox) * a.C + c) * kHiBytes<T>;
                        if constexpr (F32) {
                            f32x4 r0, r1;
#pragma unroll
                            for (int i = 0; i < 4; ++i) { r0[i] = v.v[i]; r1[i] = v.v[4 + i]; }
                            *reinterpret_cast<f32x4*>(rd) = r0;
                            *reinterpret_cast<f32x4*>(rd + 16) = r1;
                        } else {
                            u32x4 r;
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] = pack2bf(v.v[2 * i], v.v[2 * i + 1]);
                            *reinterpret_cast<u32x4*>(rd) = r;
                        }
                    }
                }
                if constexpr (F32) {
                    if constexpr (F24) {                 // what the unfused plan stores and re-loads
#pragma unroll
                        for (int i = 0; i < 8; ++i) v.v[i] = f24_round(v.v[i]);
                    }
                    f32x4 r0, r1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { r0[i] = (!MASK || inimg[u]) ? v.v[i] : 0.f; r1[i] = (!MASK || inimg[u]) ? v.v[4 + i] : 0.f; }
                    if (live) {
                        *reinterpret_cast<f32x4*>(dst + (4 * t + gg) * 16) = r0;
                        *reinterpret_cast<f32x4*>(dst + (g.PP + 4 * t + gg) * 16) = r1;
                    }
                } else {
                    u32x4 r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = pack2bf(v.v[2 * i], v.v[2 * i + 1]);
                    if constexpr (MASK) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = inimg[u] ? r[i] : 0u;
                    }
                    if (live) *reinterpret_cast<u32x4*>(dst + (4 * t + gg) * 16) = r;
                }
            }
        }
        }
    }
}

template <typename T, int K, int S, int KSE, int NTE, bool STEM = false>
__global__ __launch_bounds__(256, 2) void dwp_kernel(DwpArgs x, DwGeom g) {
    constexpr bool EXPAND = KSE > 0;      // KSE, NTE: k-steps and 32-channel tiles of the expand conv (compile time)
    // STEM: the conv in front of the depthwise conv is the 3x3 stem conv on the fp32 NCHW network input (KSE = 2: K = 27 -> 32)
    static_assert(!STEM || (KSE == 2 && !kIsF24<T>), "stem form: two k-steps, bf16 or fp32 plans");
    constexpr bool F32 = kWide<T>;        // fp32-class arithmetic (fp32 or f24 activations): split operands, hi + lo planes
    constexpr bool F24 = kIsF24<T>;
    constexpr int NS = F32 ? 2 : 1;
    constexpr int UPE = std::is_same<T, float>::value ? 2 : 1;     // 16-byte units of 8 channels in HBM (f24: of the hi plane)
    constexpr int EB = kHiBytes<T>;
    typedef typename std::conditional<EXPAND && F32, float, T>::type TT;    // format of the depthwise tile in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SW = dw_sw<S>();
    const DwArgs& a = x.dw;
    const int CBV = a.CBV;
    int tx, ty, n;
    dw_block(g, a.tiles_x, a.tiles_y, tx, ty, n);
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int IW = g.IW;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // 1x1 weights: 1-KiB fragments by LDS-DMA (waited for inside dw_stage together with the tile)
    for (int f = wave; f < x.NT * x.KS * NS; f += 4)
        __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)(reinterpret_cast<const u32x4*>(x.wp) + f * 64 + lane),
                                         (dw_lds_ptr_t)(smem + x.off_wp + f * 1024), 16, 0, 0);
    // bias and slopes of the 1x1 conv go to LDS as well: read from global memory in the epilogue they cost a
    // memory round trip per 32-pixel step that nothing hides at two to four blocks per CU
    {
        const int npb = x.NT * 8;
        if (wave == 2 && lane < 2 * npb) {
            const int which = lane >= npb, ch = (lane - which * npb) * 4;
            const float* pb = which ? x.e.slope : x.e.bias;
            const void* src = (pb && ch < x.e.cout) ? (const void*)(pb + ch) : a.zeros;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + x.off_pb), 16, 0, 0);
        }
    }
    const void* tile = smem;
    const unsigned char* tile_lo = smem + (size_t)g.IH * IW * g.PP * 16;     // f24 tile staged from HBM only
    const float* wl = reinterpret_cast<const float*>(smem + g.tile_bytes);
    const float* bl = wl + K * K * CBV * 8;
    if constexpr (!EXPAND) {
        dw_stage<T, K, S>(a, g, smem, n, 0, ty, tx);
    } else {
        // ---- expand 1x1 conv computed here: stage the (4x narrower) block input instead of the expanded
        // tile, multiply it on MFMA and write the activated result in the depthwise tile layout;
        // pixels outside the image are zero (the padding of the depthwise conv), not act(bias)
        const int cin8 = x.Cin >> 3, npin = g.IH * IW;
        const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;
        const unsigned char* Xn = reinterpret_cast<const unsigned char*>(x.xin) + (size_t)n * a.H * a.W * x.Cin * EB;
        const unsigned char* Z = reinterpret_cast<const unsigned char*>(a.zeros);
        const int nin_hi = STEM ? 0 : npin * cin8 * UPE;           // 16-byte units
        if constexpr (STEM) {
            // fp32 patch [3][IH+2][IW+2] of the network input by LDS-DMA, 4 bytes per lane (lane after lane = the row-major
            // patch, as stem_kernel stages it); outside the image: the zero page (the stem conv pads with zeros)
            const int PH = g.IH + 2, PW = IW + 2, total = 3 * PH * PW;
            const float* Xs = x.xs + (size_t)n * 3 * a.H * a.W;
            const Div32 d_hw = make_div32_dev(PH * PW), d_w = make_div32_dev(PW);
            for (int u0 = wave * 64; u0 < total; u0 += 256) {
                const int u = u0 + lane;
                if (u < total) {
                    const unsigned c = div32(u, d_hw), rem = u - c * PH * PW;
                    const unsigned iy = div32(rem, d_w), ix = rem - iy * PW;
                    const int gy = gy0 - 1 + (int)iy, gx = gx0 - 1 + (int)ix;
                    const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                    const float* src = ok ? Xs + ((int)c * a.H + gy) * a.W + gx : reinterpret_cast<const float*>(Z);
                    __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + x.off_in + u0 * 4), 4, 0, 0);
                }
            }
        }
        for (int u0 = wave * 64; u0 < nin_hi; u0 += 256) {
            const int u = u0 + lane;
            if (u < nin_hi) {
                const unsigned q = (unsigned)u / UPE;
                const unsigned pix = div32(q, x.d_cin8), cv = q - pix * cin8;
                const unsigned iy = div32(pix, g.d_iw), ix = pix - iy * IW;
                const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const unsigned char* src = ok ? Xn + ((gy * a.W + gx) * x.Cin + (int)cv * 8) * EB + (u % UPE) * 16 : Z;
                __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + x.off_in + u0 * 16), 16, 0, 0);
            }
        }
        int nin = nin_hi;
        if constexpr (F24) {                            // the lo rows of the same pixels, [pixel][LPin], behind the hi rows
            const int LPin = f24_lo_pitch(x.Cin), upp = LPin >> 4, nlo = npin * upp;
            const unsigned char* XLn = x.xin_lo + (size_t)n * a.H * a.W * LPin;
            const Div32 d_upp = make_div32_dev(upp);
            for (int u0 = wave * 64; u0 < nlo; u0 += 256) {
                const int u = u0 + lane;
                if (u < nlo) {
                    const unsigned pix = div32(u, d_upp), j = u - pix * upp;
                    const unsigned iy = div32(pix, g.d_iw), ix = pix - iy * IW;
                    const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
                    const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                    const unsigned char* src = ok ? XLn + (gy * a.W + gx) * LPin + (int)j * 16 : Z;
                    __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + x.off_in + (nin_hi + u0) * 16), 16, 0, 0);
                }
            }
            nin += nlo;
        }
        for (int f = wave; f < NTE * KSE * NS; f += 4)
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)(reinterpret_cast<const u32x4*>(x.we) + f * 64 + lane),
                                             (dw_lds_ptr_t)(smem + x.off_we + f * 1024), 16, 0, 0);
        // expand bias and slopes: [NTe*32] floats each, behind the staged input
        const int off_be = x.off_in + (STEM ? (3 * (g.IH + 2) * (IW + 2) * 4 + 15) & ~15 : (nin * 16 + 15) & ~15), nbe = NTE * 8;
        if (wave == 1 && lane < 2 * nbe) {
            const int which = lane >= nbe, ch = (lane - which * nbe) * 4;
            const float* pb = which ? x.se : x.be;
            const void* src = (pb && ch < a.C) ? (const void*)(pb + ch) : (const void*)Z;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + off_be), 16, 0, 0);
        }
        const float* bel = reinterpret_cast<const float*>(smem + off_be);
        const float* sel = x.se ? bel + NTE * 32 : nullptr;
        dw_stage<T, K, S, false>(a, g, smem, n, 0, ty, tx);               // taps + bias; waits for everything
        const bool border = gy0 < 0 || gx0 < 0 || gy0 + g.IH > a.H || gx0 + IW > a.W;
        // row tiles per iteration of the expand phase (see dwp_expand): two where the accumulators of both fit (<= 2 channel tiles)
        constexpr int XU = (FR_DWX_U2 && (STEM || NTE <= 2)) ? 2 : 1;     // (off: see FR_DWX_U2)
        if (x.act_e == FR_ACT_RELU) {
            if (border) dwp_expand<T, KSE, NTE, FR_ACT_RELU, true, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
            else dwp_expand<T, KSE, NTE, FR_ACT_RELU, false, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
        } else if (x.act_e == FR_ACT_PRELU && sel) {
            if (border) dwp_expand<T, KSE, NTE, FR_ACT_PRELU, true, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
            else dwp_expand<T, KSE, NTE, FR_ACT_PRELU, false, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
        } else if (x.act_e == FR_ACT_HSWISH) {
            if (border) dwp_expand<T, KSE, NTE, FR_ACT_HSWISH, true, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
            else dwp_expand<T, KSE, NTE, FR_ACT_HSWISH, false, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
        } else {
            dwp_expand<T, KSE, NTE, -1, true, STEM, XU>(x, g, smem, bel, sel, gy0, gx0, wave, lane);
        }
        __syncthreads();
    }

    // depthwise into registers (two passes at most): bf16, or bf16 hi + lo of the fp32 result
    u32x4 keep[2][SW][NS];
    unsigned kpx[2], kcv[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int it = threadIdx.x + 256 * ps;
        kpx[ps] = 0xffffffffu;
        if (it < g.nitems) {
            const unsigned r = div32(it, g.d_cbv), cv = it - r * CBV;
            const unsigned oy = div32(r, g.d_ns), sx = r - oy * g.nstrips;
            Vec8f out[SW];
            dw_item<TT, K, S, F24>(a, tile, wl, bl, IW, g.PP, cv, oy, sx, out, tile_lo, g.PPL);
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                if constexpr (F32) {
                    bf16x8 hi, lo;
                    if constexpr (F24) f24_split8(out[o], hi, lo);
                    else split8(out[o], hi, lo);
                    keep[ps][o][0] = __builtin_bit_cast(u32x4, hi);
                    keep[ps][o][1] = __builtin_bit_cast(u32x4, lo);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) keep[ps][o][0][i] = pack2bf(out[o].v[2 * i], out[o].v[2 * i + 1]);
                }
            }
            kpx[ps] = oy * a.TW + sx * SW;
            kcv[ps] = cv;
        }
    }
    __syncthreads();                                   // the input tile is dead: its LDS becomes the depthwise result
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
        if (kpx[ps] != 0xffffffffu) {
#pragma unroll
            for (int o = 0; o < SW; ++o)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp)
                    *reinterpret_cast<u32x4*>(smem + sp * x.off_lo + (kpx[ps] + o) * x.pitch + kcv[ps] * 16) = keep[ps][o][sp];
        }
    __syncthreads();

    // 1x1 conv on MFMA: D[cout][pixel] = W . dwo^T, 32 pixels per wave step
    const int p = lane & 31, h = lane >> 5;
    constexpr int SG = STEM ? 1 : 2;     // 32-channel tiles per staged group (stem form: 16..64 couts, and LDS for the patch and the residual)
    // rows of min(cout, SG * 32) channels: the pitch follows the layer (a 16-cout layer staged at the 64-channel pitch held 18 KB
    // of LDS per block for 4 KB of data), always an odd number of 16-byte slots; f24 keeps the fixed pitch of its two tiles
    const int PITCH = F24 ? stage_pitch(SG * 32 * EB) : x.spitch;
    constexpr int PITCH_LO = stage_pitch_lo(SG * 32);
    unsigned char* stage = smem + x.off_stage + wave * 32 * PITCH;
    unsigned char* stage_lo = smem + x.off_stage + 4 * 32 * PITCH + wave * 32 * PITCH_LO;     // f24 only
    (void)stage_lo;
    const int LPo = f24_lo_pitch(x.e.cout);
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(smem + x.off_wp) + lane;
    EpiArgs e2 = x.e;
    e2.bias = reinterpret_cast<const float*>(smem + x.off_pb);
    if (x.e.slope) e2.slope = e2.bias + x.NT * 32;
    const int cout = x.e.cout;
    const bool res_lds = STEM && x.off_res > 0;        // the residual is the stem output kept in LDS, row = pixel of the tile
    if (res_lds) e2.residual = smem + x.off_res;
    for (int rt = wave; rt * 32 < x.npx_pad; rt += 4) {
        const int px = rt * 32 + p;
        const unsigned oy = div32(px, g.d_tw), ox = px - oy * a.TW;
        const int gy = oy0 + (int)oy, gx = ox0 + (int)ox;
        const bool ok = px < x.npx && gy < a.Ho && gx < a.Wo;
        const long m = res_lds ? (long)px : ((long)n * a.Ho + gy) * a.Wo + gx;
        const unsigned char* xrow = smem + px * x.pitch + 16 * h;
        for (int tg = 0; tg < x.NT; tg += SG) {
#pragma unroll
            for (int tt = 0; tt < SG; ++tt) {
                const int t = tg + tt;
                if (t < x.NT) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    // run-time k loop (no guarded 8-step unroll: the early blocks have 1..5 steps and the kernel is
                    // bound by instruction issue); the half-step beyond C of the last step reads zeros
                    const bf16x8* wt = wf + t * x.KS * NS * 64;
#pragma unroll 1
                    for (int s = 0; s < x.KS; ++s) {
                        bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + 32 * s);
                        const bool dead = 16 * s + 8 * h >= a.C;
                        if (dead) xf = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[s * NS * 64], xf, acc, 0, 0, 0);
                        if constexpr (F32) {
                            bf16x8 xl = *reinterpret_cast<const bf16x8*>(xrow + x.off_lo + 32 * s);
                            if (dead) xl = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[s * NS * 64], xl, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[(s * NS + 1) * 64], xf, acc, 0, 0, 0);
                        }
                    }
                    if constexpr (F24) epilogue_stage_f24_h(acc, e2, m, 32 * t + 16 * h, ok, stage, PITCH, stage_lo, PITCH_LO, p, 32 * tt + 16 * h);
                    else if constexpr (F32) epilogue_stage_f32_h(acc, e2, m, 32 * t + 16 * h, ok, stage, PITCH, p, 32 * tt + 16 * h);
                    else epilogue_stage_h(acc, e2, m, 32 * t + 16 * h, ok, stage, PITCH, p, 32 * tt + 16 * h);
                }
            }
            // write the staged [32 pixels][<= 64 channels] back: consecutive pixels of a tile row are contiguous
            const int ncol = min(SG * 32, cout - tg * 32), cr = (ncol * EB) >> 4;
            const unsigned magic = (65536u + cr - 1) / cr;
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < 32 * cr; c += 64) {
                const int q = (int)(((unsigned)c * magic) >> 16), ch = c - q * cr;
                const int qpx = rt * 32 + q;
                const unsigned qy = div32(qpx, g.d_tw), qx = qpx - qy * a.TW;
                const int y = oy0 + (int)qy, xx = ox0 + (int)qx;
                if (qpx < x.npx && y < a.Ho && xx < a.Wo) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(stage + q * PITCH + ch * 16);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(x.e.y) +
                                              ((((long)n * a.Ho + y) * a.Wo + xx) * cout + tg * 32) * (long)EB + ch * 16) = v;
                }
            }
            if constexpr (F24) {
                const int crl = (ncol + 15) >> 4;
                const unsigned magl = (65536u + crl - 1) / crl;
                for (int c = lane; c < 32 * crl; c += 64) {
                    const int q = (int)(((unsigned)c * magl) >> 16), ch = c - q * crl;
                    const int qpx = rt * 32 + q;
                    const unsigned qy = div32(qpx, g.d_tw), qx = qpx - qy * a.TW;
                    const int y = oy0 + (int)qy, xx = ox0 + (int)qx;
                    if (qpx < x.npx && y < a.Ho && xx < a.Wo) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stage_lo + q * PITCH_LO + ch * 16);
                        *reinterpret_cast<u32x4*>(x.e.y_lo + (((long)n * a.Ho + y) * a.Wo + xx) * LPo + tg * 32 + ch * 16) = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- expand + depthwise + 1x1 with the expanded channels walked in chunks of 32 ------------------------------------------------
// dwp_kernel's expand form holds the expanded tile of ALL channels in LDS (9 slots of 16 / 32 bytes per input pixel for 64-72
// channels), which leaves 32..112 output pixels per block: its projection phase keeps one wave of four busy.  Here the block walks
// the expanded channels in chunks of 32 (one MFMA channel tile): expand chunk -> depthwise chunk -> the chunk's two k-steps of the
// 1x1 conv into accumulators that stay in registers.  LDS holds 5 slots per input pixel, so a block takes 2-4 x the pixels.
// The k order of the 1x1 conv (ascending k-steps, hi*hi, hi*lo, lo*hi per step) and every rounding point are those of dwp_kernel
// and of the separate kernels: bit-identical results.  cout <= 32 (one channel tile), <= 256 output pixels, <= 256 work items per
// chunk (one depthwise pass), bf16 and fp32 plans.
template <typename T, int K, int S, int KSE, int NTE>
__global__ __launch_bounds__(256, 2) void dwc_kernel(DwpArgs x, DwGeom g) {
    static_assert(!kIsF24<T> && KSE > 0 && NTE >= 2, "chunked form: bf16 / fp32 plans, expand conv with two or more channel tiles");
    constexpr bool F32 = kWide<T>;
    constexpr int NS = F32 ? 2 : 1;
    constexpr int UPE = std::is_same<T, float>::value ? 2 : 1;
    constexpr int EB = kHiBytes<T>;
    typedef typename std::conditional<F32, float, T>::type TT;    // format of the depthwise tile in LDS
    constexpr int SW = dw_sw<S>();
    constexpr int CVC = 4;                                         // 8-channel vectors per chunk
    constexpr int PITCHC = CVC * 16 + 16;                          // row pitch of the chunk's depthwise result (5 slots: odd)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DwArgs& a = x.dw;
    const int CV = a.C >> 3;
    int tx, ty, n;
    dw_block(g, a.tiles_x, a.tiles_y, tx, ty, n);
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int IW = g.IW;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    for (int f = wave; f < x.NT * x.KS * NS; f += 4)
        __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)(reinterpret_cast<const u32x4*>(x.wp) + f * 64 + lane),
                                         (dw_lds_ptr_t)(smem + x.off_wp + f * 1024), 16, 0, 0);
    {
        const int npb = x.NT * 8;
        if (wave == 2 && lane < 2 * npb) {
            const int which = lane >= npb, ch = (lane - which * npb) * 4;
            const float* pb = which ? x.e.slope : x.e.bias;
            const void* src = (pb && ch < x.e.cout) ? (const void*)(pb + ch) : a.zeros;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + x.off_pb), 16, 0, 0);
        }
    }
    const void* tile = smem;
    const float* wl = reinterpret_cast<const float*>(smem + g.tile_bytes);
    const float* bl = wl + K * K * a.CBV * 8;
    const int cin8 = x.Cin >> 3, npin = g.IH * IW;
    const int gy0 = oy0 * S - a.pad, gx0 = ox0 * S - a.pad;
    {
        const unsigned char* Xn = reinterpret_cast<const unsigned char*>(x.xin) + (size_t)n * a.H * a.W * x.Cin * EB;
        const unsigned char* Z = reinterpret_cast<const unsigned char*>(a.zeros);
        const int nin = npin * cin8 * UPE;              // 16-byte units
        for (int u0 = wave * 64; u0 < nin; u0 += 256) {
            const int u = u0 + lane;
            if (u < nin) {
                const unsigned q = (unsigned)u / UPE;
                const unsigned pix = div32(q, x.d_cin8), cv = q - pix * cin8;
                const unsigned iy = div32(pix, g.d_iw), ix = pix - iy * IW;
                const int gy = gy0 + (int)iy, gx = gx0 + (int)ix;
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const unsigned char* src = ok ? Xn + ((gy * a.W + gx) * x.Cin + (int)cv * 8) * EB + (u % UPE) * 16 : Z;
                __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + x.off_in + u0 * 16), 16, 0, 0);
            }
        }
        for (int f = wave; f < NTE * KSE * NS; f += 4)
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)(reinterpret_cast<const u32x4*>(x.we) + f * 64 + lane),
                                             (dw_lds_ptr_t)(smem + x.off_we + f * 1024), 16, 0, 0);
        const int off_be = x.off_in + ((nin * 16 + 15) & ~15), nbe = NTE * 8;
        if (wave == 1 && lane < 2 * nbe) {
            const int which = lane >= nbe, ch = (lane - which * nbe) * 4;
            const float* pb = which ? x.se : x.be;
            const void* src = (pb && ch < a.C) ? (const void*)(pb + ch) : (const void*)Z;
            __builtin_amdgcn_global_load_lds((dw_glb_ptr_t)src, (dw_lds_ptr_t)(smem + off_be), 16, 0, 0);
        }
    }
    const int off_be = x.off_in + ((npin * cin8 * UPE * 16 + 15) & ~15);
    const float* bel = reinterpret_cast<const float*>(smem + off_be);
    const float* sel = x.se ? bel + NTE * 32 : nullptr;
    dw_stage<T, K, S, false>(a, g, smem, n, 0, ty, tx);               // taps + bias of ALL channels; waits for everything
    const bool border = gy0 < 0 || gx0 < 0 || gy0 + g.IH > a.H || gx0 + IW > a.W;

    const int p = lane & 31, h = lane >> 5;
    f32x16 pacc[2];                                    // this wave's row tiles wave, wave + 4 of the 1x1 conv
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[i][r] = 0.f;
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(smem + x.off_wp) + lane;
    const int nitems_c = a.TH * g.nstrips * CVC;       // <= 256 (launcher)

#pragma unroll 1
    for (int tc = 0; tc < NTE; ++tc) {
        // (the previous chunk's projection reads of the chunk result are fenced by the barrier behind this expand phase, its
        // depthwise reads of the tile by the barrier in front of its projection)
        if (x.act_e == FR_ACT_RELU) {
            if (border) dwp_expand<T, KSE, 1, FR_ACT_RELU, true>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
            else dwp_expand<T, KSE, 1, FR_ACT_RELU, false>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
        } else if (x.act_e == FR_ACT_PRELU && sel) {
            if (border) dwp_expand<T, KSE, 1, FR_ACT_PRELU, true>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
            else dwp_expand<T, KSE, 1, FR_ACT_PRELU, false>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
        } else if (x.act_e == FR_ACT_HSWISH) {
            if (border) dwp_expand<T, KSE, 1, FR_ACT_HSWISH, true>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
            else dwp_expand<T, KSE, 1, FR_ACT_HSWISH, false>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
        } else {
            dwp_expand<T, KSE, 1, -1, true>(x, g, smem, bel, sel, gy0, gx0, wave, lane, tc);
        }
        __syncthreads();
        // depthwise conv of the chunk's channel vectors (one pass), result as bf16 (hi + lo) into the chunk result
        {
            const int ncv = min(CVC, CV - tc * CVC);
            const int it = threadIdx.x;
            const unsigned cv = it & (CVC - 1), r = (unsigned)it >> 2;
            if (it < nitems_c && (int)cv < ncv) {
                const unsigned oy = div32(r, g.d_ns), sx = r - oy * g.nstrips;
                Vec8f out[SW];
                dw_item<TT, K, S, false>(a, tile, wl + tc * CVC * 4, bl + tc * CVC * 4, IW, g.PP, cv, oy, sx, out);
                unsigned char* d = smem + x.off_dwo + (oy * a.TW + sx * SW) * PITCHC + cv * 16;
#pragma unroll
                for (int o = 0; o < SW; ++o) {
                    if constexpr (F32) {
                        bf16x8 hi, lo;
                        split8(out[o], hi, lo);
                        *reinterpret_cast<u32x4*>(d + o * PITCHC) = __builtin_bit_cast(u32x4, hi);
                        *reinterpret_cast<u32x4*>(d + x.off_lo + o * PITCHC) = __builtin_bit_cast(u32x4, lo);
                    } else {
                        u32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = pack2bf(out[o].v[2 * i], out[o].v[2 * i + 1]);
                        *reinterpret_cast<u32x4*>(d + o * PITCHC) = v;
                    }
                }
            }
        }
        __syncthreads();
        // the chunk's two k-steps of the 1x1 conv
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rt = wave + 4 * i;
            if (rt * 32 < x.npx_pad) {
                const unsigned char* xrow = smem + x.off_dwo + (rt * 32 + p) * PITCHC + 16 * h;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ks = 2 * tc + s;
                    if (ks < x.KS) {
                        bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + 32 * s);
                        const bool dead = 16 * ks + 8 * h >= a.C;
                        if (dead) xf = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks * NS * 64], xf, pacc[i], 0, 0, 0);
                        if constexpr (F32) {
                            bf16x8 xl = *reinterpret_cast<const bf16x8*>(xrow + x.off_lo + 32 * s);
                            if (dead) xl = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                            pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks * NS * 64], xl, pacc[i], 0, 0, 0);
                            pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(ks * NS + 1) * 64], xf, pacc[i], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // epilogue: bias / activation / residual, staged per wave (over the dead tile) and written back as contiguous rows
    constexpr int PITCH = stage_pitch(32 * EB);
    unsigned char* stage = smem + x.off_stage + wave * 32 * PITCH;
    EpiArgs e2 = x.e;
    e2.bias = reinterpret_cast<const float*>(smem + x.off_pb);
    if (x.e.slope) e2.slope = e2.bias + x.NT * 32;
    const int cout = x.e.cout;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = wave + 4 * i;
        if (rt * 32 >= x.npx_pad) continue;
        const int px = rt * 32 + p;
        const unsigned oy = div32(px, g.d_tw), ox = px - oy * a.TW;
        const int gy = oy0 + (int)oy, gx = ox0 + (int)ox;
        const bool ok = px < x.npx && gy < a.Ho && gx < a.Wo;
        const long m = ((long)n * a.Ho + gy) * a.Wo + gx;
        if constexpr (F32) epilogue_stage_f32_h(pacc[i], e2, m, 16 * h, ok, stage, PITCH, p, 16 * h);
        else epilogue_stage_h(pacc[i], e2, m, 16 * h, ok, stage, PITCH, p, 16 * h);
        const int cr = (cout * EB) >> 4;
        const unsigned magic = (65536u + cr - 1) / cr;
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < 32 * cr; c += 64) {
            const int q = (int)(((unsigned)c * magic) >> 16), ch = c - q * cr;
            const int qpx = rt * 32 + q;
            const unsigned qy = div32(qpx, g.d_tw), qx = qpx - qy * a.TW;
            const int y = oy0 + (int)qy, xx = ox0 + (int)qx;
            if (qpx < x.npx && y < a.Ho && xx < a.Wo) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(stage + q * PITCH + ch * 16);
                *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(x.e.y) + ((((long)n * a.Ho + y) * a.Wo + xx) * cout) * (long)EB + ch * 16) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Row pitch of the output staging tiles of dwp_kernel: min(cout, sg * 32) channels, rounded up to 16 bytes, plus one slot when that makes
// the slot count odd (b128 rows of 8 consecutive pixels on distinct bank groups), never above the fixed pitch
static int dwp_stage_pitch(int cout, int sg, int eb) {
    const int row = ((std::min(cout, sg * 32) * eb + 15) & ~15);
    const int slots = row / 16 + ((row / 16) % 2 == 0 ? 1 : 0);
    return std::min(slots * 16, stage_pitch(sg * 32 * eb));
}

struct DwpPlan { int TH, TW; size_t lds; int off_wp, off_stage, npx_pad, pitch, off_we, off_in, off_pb, off_lo, off_res, off_dwo; bool chunked; };

// Tile for the fused form: all channels in one block, <= 512 work items, everything within 64 KB of LDS
// (80 KB for fp32 plans: two blocks per CU).  cin > 0: the expand 1x1 conv (cin -> C) is computed in the kernel too.
// Measured-best tiles of the fused kernels (see kDwTiles above; tools/dwp_probe.py).
struct DwpTileEntry { int C, cin, Ho, k, s, cout, TH, TW; };
static const DwpTileEntry kDwpTiles[] = {
    // re-measured after the output staging tiles were sized by the layer (less LDS per block; tools/dwp_tile_check.sh): first match wins
    {16, 0, 112, 3, 1, 16, 8, 56},        // mobilenetv3_large features.1, bf16: 0.165 vs 0.182 ms (14 x 28)
#include "dwp_tiles.inc"
};

// fp32 plans (tools/dwp_probe.py <model> 512 f32; gpurun_out/dwp_probe_b512_*_f32.txt): LDS holds half as many pixels,
// so the winners are small tiles; shapes not listed use the cost model
static const DwpTileEntry kDwpTilesF32[] = {
    {16, 0, 56, 3, 2, 32, 14, 8}, {32, 0, 28, 3, 2, 64, 4, 28}, {32, 0, 112, 3, 1, 16, 16, 16}, {16, 0, 112, 3, 1, 16, 16, 16},
    {64, 16, 56, 3, 2, 24, 4, 8}, {72, 16, 28, 3, 2, 24, 7, 4}, {32, 0, 56, 3, 1, 32, 8, 28},
    // ({16, 0, 112, ..}: 16 x 16 re-measured after the staging tiles shrank: stem + this layer 0.429 ms vs 0.484 at 16 x 28, and vs 0.480 for
    // the stem form, which 16-channel plans therefore no longer take)
    // one block per CU (80-150 KB), measured against the best tile under 80 KB:
    {72, 24, 56, 3, 1, 24, 8, 28},        // mobilenetv3_large features.3: 0.372 ms (8 x 8 the same) vs 0.455 at 14 x 8, the winner (0.437 vs 0.603) while the staging tiles had the fixed 64-channel pitch
    {96, 16, 56, 3, 2, 24, 7, 8},         // mobilenetv2 features.2: 0.871 vs 1.10 ms
    {88, 24, 28, 3, 1, 24, 4, 28},        // mobilenetv3_small features.3: 0.148 vs 0.202 ms
};

// spare 16-byte slots per pixel (and half-plane) of a tile that the kernel computes itself (expand form)
static int dwp_tile_pad() {
    static const int pad = [] { const char* e = getenv("FR_DWP_PAD"); return e ? std::min(std::max(atoi(e), 0), 4) : 1; }();
    return pad;
}

// Shapes that take the chunked kernel (dwc_kernel) and their tiles: measured against dwp_kernel's best tile, 512 images, expand +
// depthwise + 1x1 together (tools/chunk_tile_sweep.sh; profiles/tuning/r3_chunk_tiles.txt).  bf16 plans may use 80 KB here (two blocks
// per CU).  72 channels are chunks of 32 + 32 + 8 -- the third chunk pays a whole expand pass for 8 channels -- and never won.
struct DwcTileEntry { int f32, C, cin, Ho, k, s, cout, TH, TW; };
static const DwcTileEntry kDwcTiles[] = {
    {0, 64, 16, 56, 3, 2, 24, 14, 8},     // mobilenetv3_large features.2, bf16: 0.265 vs 0.295 ms
    {1, 64, 16, 56, 3, 2, 24, 14, 8},     //                               f32: 0.546 vs 0.559
    {0, 96, 16, 56, 3, 2, 24, 14, 8},     // mobilenetv2 features.2,       bf16: 0.356 vs 0.416 (f32: 0.887 vs 0.880 at best, not taken)
    {0, 88, 24, 28, 3, 1, 24, 8, 28},     // mobilenetv3_small features.3, bf16: 0.077 vs 0.102
    {1, 88, 24, 28, 3, 1, 24, 14, 16},    //                               f32: 0.138 vs 0.161
    // not taken: mobilenetv3_large features.3 (72 channels) bf16 0.215 vs 0.215, f32 0.438 vs 0.444; mobilenetv3_small features.2
    // (72 channels, stride 2) bf16 0.119 vs 0.110, f32 0.218 vs 0.202
};

static bool dwp_pick(int fmt, int k, int s, int C, int Ho, int Wo, int cout, int cin, DwpPlan* out) {
    const bool f32 = fmt != FMT_BF16;                     // fp32-class arithmetic: hi + lo planes, fp32 expanded tile
    const bool f24 = fmt == FMT_F24;
    const bool stem = cin == DWP_STEM || cin == DWP_STEM_RES, stem_res = cin == DWP_STEM_RES;
    // chunked form (dwc_kernel): expand conv with two or more channel tiles, one cout tile, bf16 / fp32 plans; FR_DWP_CHUNK=0 keeps dwp_kernel
    // FR_DWP_CHUNK: 1 (default) the measured shapes of kDwcTiles, 2 every eligible shape (tile by the cost model or FR_DWP_TILE:
    // tools/chunk_tile_sweep.sh), 0 none
    const char* chunk_env = getenv("FR_DWP_CHUNK");       // (read per call, like FR_DWP_TILE: tests and sweeps switch it inside one process)
    const int chunk_mode = chunk_env ? atoi(chunk_env) : 1;
    bool chunked = chunk_mode > 0 && !stem && cin > 0 && !f24 && k == 3 && C > 32 && cout <= 32;
    int cth = 0, ctw = 0;
    if (chunked && chunk_mode == 1) {
        chunked = false;
        for (const DwcTileEntry& e : kDwcTiles)
            if (e.f32 == (int)f32 && e.C == C && e.cin == cin && e.Ho == Ho && Wo == Ho && e.k == k && e.s == s && e.cout == cout) { chunked = true; cth = e.TH; ctw = e.TW; break; }
    }
    if (stem) {
        if (f24 || k != 3 || C > 32 || (stem_res && (s != 1 || C != cout))) return false;
        cin = 3;
    }
    const int CV = C / 8, sw = s == 2 ? 2 : 4;
    const int ns = f32 ? 2 : 1;
    // output staging per block: 4 waves x 32 pixels x 64 channels (f24: a bf16-shaped tile for the hi halves + a byte tile)
    const size_t stage_bytes = chunked ? 4 * 32 * (size_t)stage_pitch(32 * (f32 ? 4 : 2))      // (chunked form: one 32-channel tile per staged group)
                             : !f24 ? 4 * 32 * (size_t)dwp_stage_pitch(cout, stem ? 1 : 2, f32 ? 4 : 2)    // (rows of the layer's couts, see dwp_stage_pitch)
                             : 4 * 32 * (f24 ? (size_t)stage_pitch(2 * 32 * 2) + stage_pitch_lo(2 * 32) : (size_t)stage_pitch(2 * 32 * (f32 ? 4 : 2)));
    const size_t in_px_bytes = f24 ? (size_t)cin * 2 + f24_lo_pitch(cin) : (size_t)cin * (f32 ? 4 : 2);   // staged block input per pixel (expand form)
    if (C % 8 || CV > 16 || C > 128 || cout % 8 || cout > 128 || (k != 3 && k != 5) || (s != 1 && s != 2)) return false;
    // fp32 plans without the expand conv: measured per shape on mobilenetv1 / v3 (512-image launches), the fused form wins
    // for 16..32 channels (0.146 vs 0.170 ms, 0.130 vs 0.178, 0.079 vs 0.094, 0.34 vs 0.43) and loses for 8 channels
    // (0.269 vs 0.242) and from 64 channels up (0.104 vs 0.092, 0.083 vs 0.055): the hi + lo planes leave too few pixels per tile
    static const int f32_cmax = getenv("FR_DWP_F32_CMAX") ? atoi(getenv("FR_DWP_F32_CMAX")) : 32;     // tuning
    if (f32 && !cin && (C < 16 || C > f32_cmax)) return false;
    // bf16 plans, same measurement (mobilenetv1): wins for 16..64 channels (0.066 vs 0.078 ms, 0.067 vs 0.086, 0.039 vs 0.051,
    // 0.054 vs 0.057), loses for 8 (0.146 vs 0.132) and badly for 128 channels on the 14x14 maps (0.076 vs 0.039 per layer)
    if (!f32 && !cin && (C < 16 || C > 64)) return false;
    if (cin && !stem) {        // expand form: 3x3 only, and only the (k-steps, channel tiles) shapes that are instantiated
        const int kse = cdiv(cin, 16), nte = cdiv(C, 32);
        if (cin % 8 || k != 3 || !((kse == 1 && (nte == 2 || nte == 3)) || (kse == 2 && nte == 3))) return false;
    }
    const int KS = cdiv(C, 16), NT = cdiv(cout, 32);
    const int KSe = stem ? 2 : cdiv(cin, 16), NTe = cdiv(C, 32);
    // row pitch of the LDS depthwise result (one bf16 plane): an odd number of 16-byte slots keeps the b128
    // fragment reads of 8 consecutive rows on distinct bank groups
    const int pitch = C * 2 + ((CV & 1) ? 32 : 16);
    static const int cap_env = getenv("FR_DWP_CAP_KB") ? atoi(getenv("FR_DWP_CAP_KB")) : 0;            // tuning: LDS cap of the fused kernels
    // fp32 plans may take ONE block per CU (up to 150 KB): the hi+lo planes leave so few pixels per tile at 80 KB that the expand conv
    // is recomputed 1.9x on the halo (72 channels: 8x4 tiles); a 14x8 tile in 100 KB was 28 % faster (0.44 vs 0.60 ms per 512 images)
    const size_t cap = cap_env > 0 ? (size_t)cap_env * 1024 : (f32 ? 150 * 1024 : chunked ? 80 * 1024 : 64 * 1024);
    double best = 1e30; bool found = false;
    const int ths[] = {2, 4, 7, 8, 14, 16};
    const int tws[] = {4, 8, 16, 28, 32, 56};
    int fth = 0, ftw = 0;                                 // tuning override: FR_DWP_TILE="C,cin,TH,TW" for the matching layers
    if (const char* ov = getenv("FR_DWP_TILE")) { int c0, ci; if (sscanf(ov, "%d,%d,%d,%d", &c0, &ci, &fth, &ftw) != 4 || c0 != C || ci != cin) fth = ftw = 0; }
    const bool from_env = fth != 0;                       // (a tile from the environment is tried alone; one from a table is a preference)
    if (!fth && cth) { fth = cth; ftw = ctw; }
    // stem form (cin = 3 here): 16 x 16 tiles measured best for 16 and 32 channels in both precisions (tools/stem_tile_sweep.sh, 112 x 112
    // maps, 512 images: mobilenetv3_large bf16 0.323 ms against 0.353..0.51, f32 0.462 against 0.49..0.77; mobilenetv2 0.332 / 0.469)
    if (!fth && stem && s == 1 && Ho >= 16 && Wo >= 16) { fth = 16; ftw = 16; }     // (stride 2: the cost model; a 33 x 33 input tile does not fit)
    if (!fth && Ho == Wo && !f32 && !chunked)            // (the tables hold dwp_kernel's winners)
        for (const DwpTileEntry& e : kDwpTiles)
            if (e.C == C && e.cin == cin && e.Ho == Ho && e.k == k && e.s == s && e.cout == cout) { fth = e.TH; ftw = e.TW; break; }
    if (!fth && Ho == Wo && f32 && !chunked)
        for (const DwpTileEntry& e : kDwpTilesF32)
            if (e.C == C && e.cin == cin && e.Ho == Ho && e.k == k && e.s == s && e.cout == cout) { fth = e.TH; ftw = e.TW; break; }
    // second pass: the table's tile did not fit this format's LDS (f24 plans share the fp32 tables but stage their outputs through two
    // fixed-pitch tiles): the cost model decides, instead of leaving the layers unfused
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) { if (found || !fth || from_env) break; fth = ftw = 0; }
    for (int th : ths) for (int tw : tws) {
        if (fth && (th != fth || tw != ftw)) continue;
        if (th == 2 && !f32) continue;                    // (2-row tiles: fp32 plans only, where LDS is the limit)
        if (th - Ho >= 2 || tw - Wo >= 4) continue;
        const int items = th * (tw / sw) * (chunked ? 4 : CV);
        if (items > (chunked ? 256 : 512)) continue;      // (chunked: one depthwise pass per chunk, two row tiles per wave)
        const int npx_pad = cdiv(th * tw, 32) * 32;
        if (chunked && npx_pad > 256) continue;
        const size_t plane = (size_t)npx_pad * pitch, dwo = plane * ns;
        const double IH = (th - 1) * s + k, IW = (tw - 1) * s + k;
        // tile + taps + bias; a tile computed in the kernel has one spare vector slot per pixel, so that the
        // MFMA lanes (one pixel each) write it without bank conflicts
        // (a tile computed in the kernel is fp32 half-planes for both fp32-class formats; one staged from HBM is in the plan's format)
        const int tfmt = cin ? (f32 ? FMT_F32 : FMT_BF16) : fmt;
        size_t tile = dw_lds_bytes(tfmt, k, s, th, tw, CV) + (cin ? (size_t)dwp_tile_pad() * (IH * IW) * (f32 ? 32 : 16) : 0);
        if (chunked) {
            // tile of ONE chunk (4 vectors + the spare slot per pixel), taps and bias of all channels behind it; the chunk's depthwise
            // result [npx_pad][80 B] (x 2 planes) in a region of its own; the output staging over the tile
            tile = (size_t)(IH * IW) * (4 + dwp_tile_pad()) * (f32 ? 32 : 16) + (size_t)k * k * CV * 32 + (size_t)CV * 64;
            const size_t off_dwo = align_up(std::max(tile, stage_bytes), 16);
            const size_t off_wp = align_up(off_dwo + (size_t)npx_pad * 80 * ns, 1024);
            const size_t off_pb = off_wp + (size_t)NT * KS * ns * 1024;
            const size_t off_we = off_pb + (size_t)NT * 256;
            const size_t off_in = off_we + (size_t)NTe * KSe * ns * 1024;
            const size_t lds = off_in + align_up((size_t)(IH * IW) * in_px_bytes, 16) + (size_t)NTe * 256;
            if (lds > cap) continue;
            const double useful = (double)std::min(th, Ho) * std::min(tw, Wo);
            double cost = (IH * IW) / useful + 0.5 * npx_pad / useful + (lds > 80 * 1024 ? 1.0 : lds > 53 * 1024 ? 0.4 : lds > 40 * 1024 ? 0.2 : 0.0) + (items < 192 ? 0.5 : 0.0);
            cost *= (double)cdiv(Ho, th) * th * cdiv(Wo, tw) * tw / ((double)Ho * Wo);
            if (cost < best) {
                best = cost; found = true;
                *out = DwpPlan{th, tw, lds, (int)off_wp, 0, npx_pad, 80, (int)off_we, (int)off_in, (int)off_pb, npx_pad * 80, 0, (int)off_dwo, true};
            }
            continue;
        }
        // the output staging tiles live behind the depthwise result, inside the (by then dead) input tile
        const size_t off_stage = align_up(dwo, 16);
        const size_t off_wp = align_up(std::max(tile, off_stage + stage_bytes), 1024);
        const size_t off_pb = off_wp + (size_t)NT * KS * ns * 1024;
        const size_t off_we = off_pb + (size_t)NT * 256;
        const size_t off_in = off_we + (size_t)NTe * KSe * ns * 1024;
        // staged block input (stem form: the fp32 patch [3][IH+2][IW+2]) + the expand conv's bias and slopes; stem form with a
        // residual: the stem output of the tile's own pixels behind them
        const size_t in_bytes = stem ? (size_t)3 * (IH + 2) * (IW + 2) * 4 : (size_t)(IH * IW) * in_px_bytes;
        const size_t off_res = off_in + (cin ? align_up(in_bytes, 16) + (size_t)NTe * 256 : 0);
        const size_t lds = off_res + (stem_res ? align_up((size_t)th * tw * C * (f32 ? 4 : 2), 16) : 0);
        if (lds > cap) continue;
        const double useful = (double)std::min(th, Ho) * std::min(tw, Wo);
        // staged input per useful output, padded MFMA rows, and a preference for >= 3 blocks per CU
        double cost = (IH * IW) / useful + 0.5 * npx_pad / useful + (lds > 80 * 1024 ? 1.0 : lds > 53 * 1024 ? 0.4 : lds > 40 * 1024 ? 0.2 : 0.0) + (items < 192 ? 0.5 : 0.0);
        cost *= (double)cdiv(Ho, th) * th * cdiv(Wo, tw) * tw / ((double)Ho * Wo);   // edge waste
        if (cost < best) {
            best = cost; found = true;
            *out = DwpPlan{th, tw, lds, (int)off_wp, (int)off_stage, npx_pad, pitch, (int)off_we, (int)off_in, (int)off_pb, (int)plane, stem_res ? (int)off_res : 0, 0, false};
        }
    }
    }
    return found;
}

bool dwp_supported(int fmt, int k, int stride, int C, int Ho, int Wo, int cout, int cin) {
    DwpPlan pl;
    return dwp_pick(fmt, k, stride, C, Ho, Wo, cout, cin, &pl);
}
bool dwp_is_chunked(int fmt, int k, int stride, int C, int Ho, int Wo, int cout, int cin) {
    DwpPlan pl;
    return dwp_pick(fmt, k, stride, C, Ho, Wo, cout, cin, &pl) && pl.chunked;
}

template <typename T>
static int launch_dwp_t(int k, int stride, const DwpArgs& x, const DwGeom& g, const DwpPlan& pl, bool ex, dim3 grid, hipStream_t st) {
#define FR_DWP_GO(KK, SS, KE, NE, ...) do { auto kern = dwp_kernel<T, KK, SS, KE, NE, ##__VA_ARGS__>; \
        FR_BIG_LDS_ONCE(kern, pl.lds); \
        hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, st, x, g); } while (0)
#define FR_DWP_CASE(KK, SS) if (!ex && k == KK && stride == SS) FR_DWP_GO(KK, SS, 0, 0);
#define FR_DWX_CASE(SS, KE, NE) if (ex && k == 3 && stride == SS && x.KSe == KE && x.NTe == NE) FR_DWP_GO(3, SS, KE, NE);
    if (pl.chunked) {                                  // chunked form (bf16 / fp32 plans)
        if constexpr (!kIsF24<T>) {
#define FR_DWC_CASE(SS, KE, NE) if (ex && k == 3 && stride == SS && x.KSe == KE && x.NTe == NE) { auto kern = dwc_kernel<T, 3, SS, KE, NE>; \
            FR_BIG_LDS_ONCE(kern, pl.lds); hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, st, x, g); }
            FR_DWC_CASE(1, 1, 2) else FR_DWC_CASE(2, 1, 2) else FR_DWC_CASE(1, 1, 3) else FR_DWC_CASE(2, 1, 3)
            else FR_DWC_CASE(1, 2, 3) else FR_DWC_CASE(2, 2, 3)
            else FR_FAIL(FR_ERR_UNSUPPORTED, "expand+dw+1x1 (chunked): no kernel for this shape");
#undef FR_DWC_CASE
        } else FR_FAIL(FR_ERR_UNSUPPORTED, "expand+dw+1x1 (chunked): not built for f24 plans");
    } else
    if (x.xs) {                                        // stem form (bf16 / fp32 plans; dwp_pick refuses f24)
        if constexpr (!kIsF24<T>) {
            if (ex && k == 3 && stride == 1 && x.NTe == 1) FR_DWP_GO(3, 1, 2, 1, true);
            else if (ex && k == 3 && stride == 2 && x.NTe == 1) FR_DWP_GO(3, 2, 2, 1, true);
            else FR_FAIL(FR_ERR_UNSUPPORTED, "stem+dw+1x1: no kernel for this shape");
        } else FR_FAIL(FR_ERR_UNSUPPORTED, "stem+dw+1x1: not built for f24 plans");
    } else
    FR_DWP_CASE(3, 1) else FR_DWP_CASE(3, 2) else FR_DWP_CASE(5, 1) else FR_DWP_CASE(5, 2)
    else FR_DWX_CASE(1, 1, 2) else FR_DWX_CASE(2, 1, 2) else FR_DWX_CASE(1, 1, 3) else FR_DWX_CASE(2, 1, 3)
    else FR_DWX_CASE(1, 2, 3) else FR_DWX_CASE(2, 2, 3)
    else FR_FAIL(FR_ERR_UNSUPPORTED, "dw+1x1: no kernel for this shape");
#undef FR_DWP_CASE
#undef FR_DWX_CASE
#undef FR_DWP_GO
    FR_LAUNCH_CHECK("dwp_kernel");
    return FR_OK;
}

int launch_dwp(int fmt, int k, int stride, DwArgs a, const bf16* wp, const EpiArgs& e, const DwpExpand* ex, hipStream_t st) {
    DwpPlan pl;
    const bool f32 = fmt != FMT_BF16;
    const int cin = !ex ? 0 : ex->stem ? (ex->res_lds ? DWP_STEM_RES : DWP_STEM) : ex->Cin;
    if (!a.zeros || a.N <= 0 || !dwp_pick(fmt, k, stride, a.C, a.Ho, a.Wo, e.cout, cin, &pl)) FR_FAIL(FR_ERR_UNSUPPORTED, "dw+1x1: shape not supported");
    if (ex && ex->stem && (!ex->xs || a.pad != 1 || (ex->res_lds && e.residual))) FR_FAIL(FR_ERR_ARG, "stem+dw+1x1: inconsistent arguments");
    if (fmt == FMT_F24 && (!e.y_lo || (e.residual && !e.residual_lo) || (ex ? !ex->xin_lo : !a.x_lo)))
        FR_FAIL(FR_ERR_ARG, "dw+1x1: f24 tensors without their lo planes");
    if ((long)a.H * a.W * a.C * 4 >= 0x7FFFFFFFL) FR_FAIL(FR_ERR_UNSUPPORTED, "dw+1x1: image above 2 GB");
    a.TH = pl.TH; a.TW = pl.TW; a.CBV = a.C / 8; a.cchunks = 1;
    a.tiles_y = cdiv(a.Ho, a.TH); a.tiles_x = cdiv(a.Wo, a.TW);
    a.pool = nullptr;
    if (a.tiles_y > 65535) FR_FAIL(FR_ERR_ARG, "dw+1x1: grid too large");
    DwGeom g{};
    const int sw = stride == 2 ? 2 : 4;
    g.IH = (a.TH - 1) * stride + k; g.IW = (a.TW - 1) * stride + k;
    g.nstrips = a.TW / sw;
    g.nitems = a.TH * g.nstrips * a.CBV;
    g.nunits = g.IH * g.IW * (ex ? a.CBV : a.CBV) * (fmt == FMT_F32 ? 2 : 1);      // (tile staged from HBM only without the expand conv: PP == CBV)
    g.PP = pl.chunked ? 4 + dwp_tile_pad() : ex ? a.CBV + dwp_tile_pad() : a.CBV;
    g.PPL = (a.CBV + 1) / 2;
    g.tile_bytes = (int)dw_tile_bytes(ex ? (f32 ? FMT_F32 : FMT_BF16) : fmt, g.IH, g.IW, g.PP);
    g.d_cbv = make_div32(a.CBV); g.d_iw = make_div32(g.IW); g.d_ns = make_div32(g.nstrips); g.d_tw = make_div32(a.TW);
    g.d_pp = make_div32(g.PP); g.d_th = make_div32(a.TH); g.order = 0;
    DwpArgs x{};
    x.dw = a; x.wp = wp; x.e = e; x.KS = cdiv(a.C, 16); x.NT = cdiv(e.cout, 32);
    x.npx = a.TH * a.TW; x.npx_pad = pl.npx_pad; x.pitch = pl.pitch; x.off_lo = pl.off_lo;
    x.off_wp = pl.off_wp; x.off_stage = pl.off_stage; x.off_pb = pl.off_pb;
    x.spitch = dwp_stage_pitch(e.cout, (ex && ex->stem) ? 1 : 2, f32 ? 4 : 2);
    if (ex) {
        x.xin = ex->xin; x.xin_lo = ex->xin_lo; x.we = ex->we; x.be = ex->bias; x.se = ex->slope; x.act_e = ex->act; x.Cin = ex->Cin;
        x.KSe = ex->stem ? 2 : cdiv(ex->Cin, 16); x.NTe = cdiv(a.C, 32); x.off_we = pl.off_we; x.off_in = pl.off_in;
        x.d_cin8 = make_div32(std::max(1, ex->Cin / 8));
        if (ex->stem) { x.xs = ex->xs; x.Cin = 3; x.off_res = pl.off_res; }
        x.off_dwo = pl.off_dwo;
    }
    for (int n0 = 0; n0 < a.N; n0 += 65535) {
        g.n0 = n0;
        dim3 grid((unsigned)a.tiles_x, (unsigned)a.tiles_y, (unsigned)std::min(65535, a.N - n0));
        dw_block_geom(g, (int)grid.x, (int)grid.y, (int)grid.z);
        const int rc = fmt == FMT_F24 ? launch_dwp_t<f24>(k, stride, x, g, pl, ex != nullptr, grid, st)
                     : f32 ? launch_dwp_t<float>(k, stride, x, g, pl, ex != nullptr, grid, st)
                           : launch_dwp_t<bf16>(k, stride, x, g, pl, ex != nullptr, grid, st);
        if (rc) return rc;
    }
    return FR_OK;
}
