// Pointwise (1x1) convolution / Linear as a bf16 MFMA GEMM with a fused epilogue.
//
// Replaces nn.Conv2d(k=1)+BatchNorm2d+activation (utils/layers.py:15-84), the SE scale multiply
// (utils/layers.py:122), the residual add (models/mobilenetv3.py:96-97, mobilenetv2.py:55) and
// nn.Linear+BatchNorm1d (utils/layers.py:183-186).
//
// Orientation: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k] on v_mfma_f32_32x32x16_bf16, i.e.
// the (tiny, LDS-resident, pre-packed) weights are the MFMA "A" operand and the streamed NHWC
// activations are the "B" operand, loaded straight from global memory: lane (p = lane&31,
// h = lane>>5) reads the 16 contiguous bytes X[pixel p][16s+8h .. +7].  With the row permutation
// of common.h each lane ends up with 16 consecutive output channels of one pixel, so the epilogue
// (bias, PReLU/Hardswish/ReLU/Hardsigmoid, residual) stores 16-byte NHWC vectors.
//
// The op is HBM-bound: every activation is read once and written once; weights
// are staged into LDS once per block when they fit (k-chunked otherwise).
#include "kernels.h"

// Kernels that may need more than the default 64 KB of dynamic LDS raise their limit first.
#define FR_BIG_LDS(kernel, bytes) FR_BIG_LDS_ONCE(kernel, bytes)

typedef __attribute__((address_space(3))) void* pw_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pw_glb_ptr_t;

// Pixel-block x / column-group g from a 1-D grid laid out so that the column groups of one pixel
// block get ids that are equal mod 8 and close together: workgroups are dealt round-robin over the
// 8 XCDs, so those blocks share an XCD's L2 and read the activations from HBM once, not once per
// group (measured with FETCH_SIZE: the expansion layers fetched their input up to 15 times).
struct PwGrid { int gx, ng; };
FR_DEV bool pw_block(const PwGrid& pg, int& x, int& g) {
    const int id = blockIdx.x, per = 8 * pg.ng;
    const int xb = id / per, r = id - xb * per;
    g = r >> 3;
    x = xb * 8 + (r & 7);
    return x < pg.gx;
}
static inline unsigned pw_grid_blocks(int gx, int ng) { return (unsigned)(((gx + 7) / 8) * 8 * ng); }

template <typename TIn>
struct XFrag;
template <>
struct XFrag<bf16> {
    bf16x8 hi;
    FR_DEV void zero() { hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    FR_DEV void load(const bf16* p, const float* se) {
        if (se) {
            Vec8f x = load8(p), s = load8(se);
#pragma unroll
            for (int i = 0; i < 8; ++i) x.v[i] *= s.v[i];
            hi = to_frag(x);
        } else {
            hi = *reinterpret_cast<const bf16x8*>(p);
        }
    }
};
template <>
struct XFrag<float> {
    bf16x8 hi, lo;
    FR_DEV void zero() { hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; lo = hi; }
    FR_DEV void load(const float* p, const float* se) {
        Vec8f x = load8(p);
        if (se) {
            Vec8f s = load8(se);
#pragma unroll
            for (int i = 0; i < 8; ++i) x.v[i] *= s.v[i];
        }
        split8(x, hi, lo);
    }
};

template <>
struct XFrag<f24> {
    bf16x8 hi, lo;
    FR_DEV void zero() { hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; lo = hi; }
    // p: 8 channels of the hi plane, pl: the same 8 channels of the lo plane
    FR_DEV void load(const bf16* p, const unsigned char* pl, const float* se) {
        const u32x4 h = *reinterpret_cast<const u32x4*>(p);
        const unsigned long long l = *reinterpret_cast<const unsigned long long*>(pl);
        if (se) {
            Vec8f x = f24_decode(h, (unsigned)l, (unsigned)(l >> 32)), s = load8(se);
#pragma unroll
            for (int i = 0; i < 8; ++i) x.v[i] *= s.v[i];
            split8(x, hi, lo);
        } else {
            f24_frags(h, (unsigned)l, (unsigned)(l >> 32), hi, lo);
        }
    }
};

template <typename TIn, typename TOut, int NTB>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(PwArgs a, PwGrid pg) {
    constexpr bool SPLIT = std::is_same<TIn, float>::value;
    constexpr int NS = SPLIT ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int p = lane & 31, h = lane >> 5;
    int bx, bg;
    if (!pw_block(pg, bx, bg)) return;
    const int t0 = bg * NTB;
    const int nt = min(NTB, a.NT - t0);
    const int nchunks = (a.KS + a.chunk_ks - 1) / a.chunk_ks;
    const long n_super = (a.M + 127) / 128;
    const TIn* X = reinterpret_cast<const TIn*>(a.x);
    bool staged = false;

    for (long st = bx; st < n_super; st += pg.gx) {
        const long m = st * 128 + wave * 32 + p;
        const bool m_ok = m < a.M;
        const long mc = m_ok ? m : a.M - 1;
        const TIn* xrow = X + mc * (long)a.K;
        const float* serow = a.se ? a.se + (mc / a.HW) * (long)a.K : nullptr;

        f32x16 acc[NTB];
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            const int s0 = c * a.chunk_ks;
            const int ns = min(a.chunk_ks, a.KS - s0);
            if (nchunks > 1 || !staged) {
                if (nchunks > 1) __syncthreads();
                // LDS-DMA: one 1-KiB fragment per wave instruction, all issued before the single wait
                const int nfrag = nt * ns * NS;
                for (int f = wave_u; f < nfrag; f += 4) {
                    const int sp = f % NS, ts = f / NS;
                    const int s = ts % ns, t = ts / ns;
                    const u32x4* src = reinterpret_cast<const u32x4*>(a.w) +
                        ((((long)(t0 + t) * a.KS + s0 + s) * NS + sp) * 64 + lane);
                    __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)src, (pw_lds_ptr_t)(smem + f * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                staged = true;
            }
            for (int sg = 0; sg < ns; sg += 4) {
                XFrag<TIn> xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = (s0 + sg + j) * 16 + 8 * h;
                    if (sg + j < ns && kk < a.K) xf[j].load(xrow + kk, serow ? serow + kk : nullptr);
                    else xf[j].zero();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (sg + j < ns) {
#pragma unroll
                        for (int t = 0; t < NTB; ++t) {
                            if (t < nt) {
                                const bf16x8* wf = reinterpret_cast<const bf16x8*>(smem) +
                                    ((t * ns + sg + j) * NS) * 64 + lane;
                                bf16x8 wh = wf[0];
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[j].hi, acc[t], 0, 0, 0);
                                if constexpr (SPLIT) {
                                    bf16x8 wl = wf[64];
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[j].lo, acc[t], 0, 0, 0);
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xf[j].hi, acc[t], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (std::is_same<TOut, bf16>::value) {
            constexpr int SG = NTB < 2 ? NTB : 2;                 // n-tiles staged together: 128-byte rows
            constexpr int PITCH = stage_pitch(SG * 64);
            unsigned char* stage = smem + a.stage_off + (size_t)wave * 32 * PITCH;
            const long tile0 = m - p, left = a.M - tile0;
#pragma unroll
            for (int tg = 0; tg < NTB; tg += SG) {
                if (tg < nt) {
#pragma unroll
                    for (int t = tg; t < tg + SG; ++t)
                        if (t < nt) epilogue_stage(acc[t], a.e, m, 32 * (t0 + t) + 16 * h, m_ok, stage, PITCH, p, 32 * (t - tg) + 16 * h);
                    if (left > 0)
                        stage_flush(stage, PITCH, lane, reinterpret_cast<bf16*>(a.e.y) + tile0 * a.e.cout + (t0 + tg) * 32, a.e.cout,
                                    min(SG * 32, a.e.cout - (t0 + tg) * 32), left < 32 ? (int)left : 32);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NTB; ++t)
                if (t < nt) epilogue_store<TOut>(acc[t], a.e, m, 32 * (t0 + t) + 16 * h, m_ok);
        }
    }
}


// ---- large-K variant (K > 256, bf16): the project convs after squeeze-excite ------------------------
// The k-chunked kernel above stages a weight chunk, waits, computes, and only then asks for the
// next chunk and the next activations: every chunk pays a full memory round trip.  Here a block
// owns 128 pixels x (up to 5 n-tiles = 160 channels), and both streams are software-pipelined:
// while chunk c (4 k-steps = 64 input channels) is multiplied, the LDS-DMA of weight chunk c+1
// into the other LDS buffer and the loads of activation chunk c+1 (and its SE scales) into a
// second register set are already in flight.  One wait + one barrier per chunk.
constexpr int PJ_NT = 5;          // most n-tiles per block (NTC = 2..5, a template parameter)
constexpr int PJ_IMGS = 4;        // images a 128-pixel super-tile can span (needs H*W >= 43)

// T = bf16: chunks of 4 k-steps (64 input channels), bf16 output staged through LDS.
// T = float (fp32 plans, round 2): chunks of 2 k-steps with hi+lo weight fragments (the same 20 KB per buffer), fp32
// activations split on use, three MFMA passes in pw_gemm_kernel's order (same bits), fp32 output stored directly (these
// are the narrow outputs of the project convs: <= 160 channels, 64 contiguous bytes per lane).
// NTC: n-tiles per block at compile time (round 3).  With a run-time tile count every tile of the multiply sat behind
// a uniform branch, so each weight fragment was read, waited for and used on its own: two exposed LDS round trips per
// tile and k-step (fp32: 360 cycles for 96 cycles of MFMA).  With NTC fixed the fragments of all tiles are read
// up front and the passes run tile-interleaved.  A block whose column group has fewer than NTC tiles multiplies the
// (never written, never stored) LDS slots of the missing tiles.
template <bool SE, typename T, int NTC>
__global__ __launch_bounds__(256, 2) void pw_proj_kernel(PwArgs a) {
    constexpr bool F32 = kWide<T>;                      // fp32-class arithmetic: fp32 or f24 activations
    constexpr bool F24 = kIsF24<T>;
    typedef typename act_hi<T>::type TH;
    constexpr int NS = F32 ? 2 : 1;
    constexpr int CK = F32 ? 2 : 4;                     // k-steps per chunk
    constexpr int XU = (F32 && !F24) ? 2 : 1;           // 16-byte units of one lane's 8 input channels (f24: the hi plane's)
    struct Raw { u32x4 x[CK][XU]; unsigned long long xl[F24 ? CK : 1]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(NTC >= 2 && NTC <= PJ_NT, "2..5 n-tiles per block");
    constexpr int WBUF = NTC * CK * NS * 1024;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.y * a.ntg;
    const int nt = min(a.ntg, a.NT - t0);
    const int nch = (a.KS + CK - 1) / CK;
    const long n_super = (a.M + 127) / 128;
    const TH* X = reinterpret_cast<const TH*>(a.x);
    const int LPK = f24_lo_pitch(a.K);
    constexpr int SG = 2;
    constexpr int PITCH = stage_pitch(SG * 64);
    constexpr int STAGE_BYTES = F32 ? 0 : 4 * 32 * PITCH;
    unsigned char* stage = smem + 2 * WBUF + (size_t)wave * 32 * PITCH;
    // SE scales of the (<= 4) images this super-tile touches: [PJ_IMGS][K] fp32, staged once per
    // super-tile (per-lane global loads of the scales would be twice the activation bytes)
    float* sc = reinterpret_cast<float*>(smem + 2 * WBUF + STAGE_BYTES);
    const long n_img = a.M / a.HW;

    for (long st = blockIdx.x; st < n_super; st += gridDim.x) {
        const long m = st * 128 + wave * 32 + p;
        const bool m_ok = m < a.M;
        const long mc = m_ok ? m : a.M - 1;
        const TH* xrow = X + mc * (long)a.K;
        const unsigned char* xlrow = F24 ? a.x_lo + mc * (long)LPK : nullptr;
        (void)xlrow;
        const long img0 = (st * 128) / a.HW;
        const float* srow = sc + (int)(mc / a.HW - img0) * a.K;
        if constexpr (SE) {
            const int upr = a.K >> 2;                                   // 16-byte units per scale row
            const int nu = PJ_IMGS * upr;
            for (int u0 = wave * 64; u0 < nu; u0 += 256) {
                const int u = u0 + lane;
                if (u < nu) {
                    const int i = u / upr, r = u - i * upr;
                    const long img = img0 + i < n_img ? img0 + i : n_img - 1;
                    __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)(a.se + img * a.K + r * 4),
                                                     (pw_lds_ptr_t)(reinterpret_cast<unsigned char*>(sc) + u0 * 16), 16, 0, 0);
                }
            }
        }

        auto issue = [&](int c, int buf, Raw& raw) {
            const int s0 = c * CK;
            const int ns = min(CK, a.KS - s0);
            const int nfrag = nt * ns * NS;
            for (int f = wave; f < nfrag; f += 4) {
                const int sp = f % NS, ts = f / NS;
                const int t = ts / ns, s = ts - t * ns;
                const u32x4* src = reinterpret_cast<const u32x4*>(a.w) + ((((long)(t0 + t) * a.KS + s0 + s) * NS + sp) * 64 + lane);
                __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)src, (pw_lds_ptr_t)(smem + buf * WBUF + ((t * CK + s) * NS + sp) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                const int kk = (s0 + j) * 16 + 8 * h;
                const int kc = kk < a.K ? kk : 0;                       // clamped: the k-step is skipped or its weights are zero
#pragma unroll
                for (int u = 0; u < XU; ++u) raw.x[j][u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(xrow + kc) + 16 * u);
                if constexpr (F24) raw.xl[j] = *reinterpret_cast<const unsigned long long*>(xlrow + kc);
            }
        };

        f32x16 acc[NTC];
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        auto compute = [&](int c, int buf, const Raw& raw) {
            const int s0 = c * CK;
            const int ns = min(CK, a.KS - s0);
            const bf16x8* wb = reinterpret_cast<const bf16x8*>(smem + buf * WBUF) + lane;
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                if (j < ns) {
                    bf16x8 xf, xl;
                    const bool live = (s0 + j) * 16 + 8 * h < a.K;
                    if constexpr (F24 && !SE) {
                        f24_frags(raw.x[j][0], (unsigned)raw.xl[j], (unsigned)(raw.xl[j] >> 32), xf, xl);
                    } else if constexpr (SE || F32) {
                        Vec8f x;
                        if constexpr (F24) {
                            x = f24_decode(raw.x[j][0], (unsigned)raw.xl[j], (unsigned)(raw.xl[j] >> 32));
                        } else if constexpr (F32) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) { x.v[i] = __uint_as_float(raw.x[j][0][i]); x.v[4 + i] = __uint_as_float(raw.x[j][XU - 1][i]); }
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                x.v[2 * i] = __uint_as_float(raw.x[j][0][i] << 16);
                                x.v[2 * i + 1] = __uint_as_float(raw.x[j][0][i] & 0xffff0000u);
                            }
                        }
                        if constexpr (SE) {
                            const int kc = live ? (s0 + j) * 16 + 8 * h : 0;
                            const f32x4 sa = *reinterpret_cast<const f32x4*>(srow + kc), sb = *reinterpret_cast<const f32x4*>(srow + kc + 4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) { x.v[i] *= sa[i]; x.v[4 + i] *= sb[i]; }
                        }
                        if constexpr (F32) split8(x, xf, xl); else xf = to_frag(x);
                    } else {
                        xf = __builtin_bit_cast(bf16x8, raw.x[j][0]);
                    }
                    if (!live) { xf = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; xl = xf; }
                    // every accumulator still sees hi.hi, hi.lo, lo.hi in that order (same bits as before)
                    bf16x8 wh[NTC], wlo[NTC];
#pragma unroll
                    for (int t = 0; t < NTC; ++t) {
                        wh[t] = wb[((t * CK + j) * NS) * 64];
                        if constexpr (F32) wlo[t] = wb[((t * CK + j) * NS + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);     // (the scheduler otherwise sinks the lo fragments to their uses, one wait each)
#pragma unroll
                    for (int t = 0; t < NTC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], xf, acc[t], 0, 0, 0);
                    if constexpr (F32) {
#pragma unroll
                        for (int t = 0; t < NTC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], xl, acc[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NTC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[t], xf, acc[t], 0, 0, 0);
                    }
                }
            }
        };

        Raw r0, r1;
        issue(0, 0, r0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = 0; c < nch; c += 2) {
            if (c + 1 < nch) issue(c + 1, 1, r1);
            compute(c, 0, r0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c + 1 < nch) {
                if (c + 2 < nch) issue(c + 2, 0, r0);
                compute(c + 1, 1, r1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }

        if constexpr (F32 && !F24) {
            // fp32 output through a wave-private staging tile in the (now dead) weight buffers: the direct form wrote 64-byte
            // pieces per lane at a row stride of cout * 4 bytes and took 18 % of a block's time (cycle-counter stamps)
            constexpr int PITCH32 = stage_pitch(SG * 32 * 4);
            unsigned char* stage32 = smem + (size_t)wave * 32 * PITCH32;
            const long tile0 = m - p, left = a.M - tile0;
#pragma unroll
            for (int tg = 0; tg < NTC; tg += SG) {
                if (tg < nt) {
#pragma unroll
                    for (int t = tg; t < tg + SG; ++t)
                        if (t < NTC && t < nt) epilogue_stage_f32_h(acc[t], a.e, m, 32 * (t0 + t) + 16 * h, m_ok, stage32, PITCH32, p, 32 * (t - tg) + 16 * h);
                    if (left > 0)
                        stage_flush_f32(stage32, PITCH32, lane, reinterpret_cast<float*>(a.e.y) + tile0 * a.e.cout + (t0 + tg) * 32, a.e.cout,
                                        min(SG * 32, min(a.e.cout, (t0 + nt) * 32) - (t0 + tg) * 32), left < 32 ? (int)left : 32);
                }
            }
            __syncthreads();                           // the next super-tile's weight DMA writes this region
        } else if constexpr (F32) {
#pragma unroll
            for (int t = 0; t < NTC; ++t)
                if (t < nt) epilogue_store_h<T>(acc[t], a.e, m, 32 * (t0 + t) + 16 * h, m_ok);
        } else {
            const long tile0 = m - p, left = a.M - tile0;
#pragma unroll
            for (int tg = 0; tg < NTC; tg += SG) {
                if (tg < nt) {
#pragma unroll
                    for (int t = tg; t < tg + SG; ++t)
                        if (t < NTC && t < nt) epilogue_stage_h(acc[t], a.e, m, 32 * (t0 + t) + 16 * h, m_ok, stage, PITCH, p, 32 * (t - tg) + 16 * h);
                    if (left > 0)
                        stage_flush(stage, PITCH, lane, reinterpret_cast<bf16*>(a.e.y) + tile0 * a.e.cout + (t0 + tg) * 32, a.e.cout,
                                    min(SG * 32, min(a.e.cout, (t0 + nt) * 32) - (t0 + tg) * 32), left < 32 ? (int)left : 32);
                }
            }
        }
    }
}

template <typename T, int NTC>
static int launch_pw_proj_n(const PwArgs& a, hipStream_t st) {
    constexpr bool F32 = kWide<T>;
    const long n_super = (a.M + 127) / 128;
    PwArgs b = a;
    b.ntg = NTC;
    const int groups = cdiv(a.NT, b.ntg);
    dim3 grid((unsigned)std::min<long>(n_super, 4096), (unsigned)groups);
    size_t lds = 2 * (size_t)NTC * (F32 ? 2 * 2 : 4) * 1024 + (F32 ? 0 : 4 * 32 * (size_t)stage_pitch(2 * 64));
    if (std::is_same<T, float>::value) lds = std::max(lds, 4 * 32 * (size_t)stage_pitch(2 * 32 * 4));   // fp32 output staging lives in the weight buffers
    if (a.se) lds += (size_t)PJ_IMGS * a.K * 4;
    if (a.se) { FR_BIG_LDS((pw_proj_kernel<true, T, NTC>), lds); hipLaunchKernelGGL((pw_proj_kernel<true, T, NTC>), grid, dim3(256), lds, st, b); }
    else hipLaunchKernelGGL((pw_proj_kernel<false, T, NTC>), grid, dim3(256), lds, st, b);
    FR_LAUNCH_CHECK("pw_proj_kernel");
    return FR_OK;
}

// n-tiles per block: all of them when there are at most 5, else the count in 5, 4, 3 that wastes the fewest tiles
static int pw_proj_ntc(int NT, long n_super) {
    // tuning: FR_PJ_NTC=n splits the n-tiles of a layer with few 128-pixel super-tiles (<= FR_PJ_NTC_SUPER) into column groups of n.
    // Measured (tools/pj_ntc_sweep.sh, mobilenetv3_large, 512 images): the three 7x7 layers with 160 couts (196 blocks on 256 CUs) as
    // 3 + 2 tiles: -13 % per layer in bf16, -6 % in f32 alone on the GPU -- and nothing end to end (256.0 vs 256.7 k img/s, f32 128.1 vs
    // 128.0): the other sub-batch's stream already fills the idle CUs.  On the 14x14 layers (784 blocks) a split costs 30-40 %.  Off.
    static const int force = getenv("FR_PJ_NTC") ? atoi(getenv("FR_PJ_NTC")) : 0;
    static const long force_super = getenv("FR_PJ_NTC_SUPER") ? atol(getenv("FR_PJ_NTC_SUPER")) : 256;
    if (force >= 2 && force <= 4 && NT <= PJ_NT && NT > force && n_super <= force_super) return force;
    if (NT <= PJ_NT) return std::max(NT, 2);
    int best = 5;
    for (int n = 4; n >= 3; --n)
        if (cdiv(NT, n) * n < cdiv(NT, best) * best) best = n;
    return best;
}

template <typename T>
static int launch_pw_proj(const PwArgs& a, hipStream_t st) {
    switch (pw_proj_ntc(a.NT, (a.M + 127) / 128)) {
        case 2: return launch_pw_proj_n<T, 2>(a, st);
        case 3: return launch_pw_proj_n<T, 3>(a, st);
        case 4: return launch_pw_proj_n<T, 4>(a, st);
        default: return launch_pw_proj_n<T, 5>(a, st);
    }
}

// ---- streaming variant: all weights (+bias/slope) resident in LDS, K <= 128, bf16 input.
// Each wave owns MT consecutive 32-pixel tiles per iteration and issues ALL their activation (and
// residual) loads before the first MFMA, so that >= 2-4 KB per wave are in flight: the op is pure
// HBM streaming and bytes in flight per CU, not arithmetic, set its speed.
// KSM: max k-steps.  SG: n-tiles staged together for the coalesced write-back; SG = NTB when the
// block covers whole output rows (one column group), so that the 32-pixel tile leaves as ONE linear
// run (a trailing 8- or 16-channel group of its own would again write 16/32-byte pieces).
// ACT: the layer's activation at compile time (FR_ACT_NONE / PRELU / HSWISH / RELU).  With a run-time `act` every 8-channel
// group of the epilogue carried the whole ladder of uniform branches over all activation forms: the small-K layers are bound by
// instruction issue (24 -> 72 channels at 56 x 56: ~9 instructions per output element per lane, 61 of its 75 us), not by bytes.
template <typename TIn, typename TOut, int NTB, int MT, int KSM, int SG, int ACT>
__global__ __launch_bounds__(256, 2) void pw_stream_kernel(PwArgs a, PwGrid pg) {
    // TIn = float (fp32 plans): fp32 activations are split into bf16 hi + lo on load and every product is three
    // MFMA passes against the hi/lo weight fragments, in the order pw_gemm_kernel uses (same bits either way)
    constexpr bool SPLIT = kWide<TIn>;
    constexpr bool F24 = kIsF24<TIn>;                                    // f24 goes in and out together
    static_assert(kIsF24<TIn> == kIsF24<TOut>, "f24 plans: input and output are both f24");
    typedef typename act_hi<TIn>::type THI;
    typedef typename act_hi<TOut>::type THO;
    constexpr int NS = SPLIT ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    int bx, bg;
    if (!pw_block(pg, bx, bg)) return;
    const int t0 = bg * NTB;
    const int nt = min(NTB, a.NT - t0);
    const int KS = a.KS;
    bf16x8* wl = reinterpret_cast<bf16x8*>(smem);                         // [nt][KS][NS][64]
    float* bl = reinterpret_cast<float*>(smem + (size_t)NTB * KS * NS * 1024);  // [NTB*32] bias, then slope
    {
        // LDS-DMA: weights as 1-KiB fragments, bias and slopes as 16-byte units; one wait for all
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const u32x4* src = reinterpret_cast<const u32x4*>(a.w) + (size_t)t0 * KS * NS * 64;
        for (int f = wave_u; f < nt * KS * NS; f += 4)
            __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)(src + f * 64 + lane), (pw_lds_ptr_t)(smem + f * 1024), 16, 0, 0);
        if (wave_u == 0 && lane < nt * 8)
            __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)(a.e.bias + t0 * 32 + lane * 4), (pw_lds_ptr_t)(bl), 16, 0, 0);
        if (wave_u == 1 && lane < nt * 8 && a.e.slope)
            __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)(a.e.slope + t0 * 32 + lane * 4), (pw_lds_ptr_t)(bl + NTB * 32), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const THI* X = reinterpret_cast<const THI*>(a.x);
    const THO* R = reinterpret_cast<const THO*>(a.e.residual);
    THO* Y = reinterpret_cast<THO*>(a.e.y);
    const long per_iter = 128L * MT;
    const long n_iter = (a.M + per_iter - 1) / per_iter;
    const int cout = a.e.cout;
    constexpr bool STAGED = true;                          // bf16, fp32 and f24 outputs all leave through a wave-private LDS tile
    constexpr bool OUT32 = std::is_same<TOut, float>::value;
    constexpr int PITCH = stage_pitch(SG * 32 * kHiBytes<TOut>);
    constexpr int PITCH_LO = stage_pitch_lo(SG * 32);      // f24: second tile for the lo bytes
    unsigned char* stage = smem + (size_t)NTB * KS * NS * 1024 + (size_t)NTB * 256 + (size_t)wave * 32 * PITCH;
    unsigned char* stage_lo = smem + (size_t)NTB * KS * NS * 1024 + (size_t)NTB * 256 + (size_t)4 * 32 * PITCH + (size_t)wave * 32 * PITCH_LO;
    const int LPK = f24_lo_pitch(a.K), LPC = f24_lo_pitch(cout);
    (void)stage_lo; (void)LPK; (void)LPC;

    for (long it = bx; it < n_iter; it += pg.gx) {
        const long mbase = it * per_iter + (long)wave * 32 * MT + p;
        XFrag<TIn> xf[MT][KSM];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long m = mbase + mt * 32;
            const long mc = m < a.M ? m : a.M - 1;
            const THI* xrow = X + mc * (long)a.K;
            const float* serow = a.se ? a.se + (mc / a.HW) * (long)a.K : nullptr;
#pragma unroll
            for (int s = 0; s < KSM; ++s) {
                const int kk = s * 16 + 8 * h;
                if (s < KS && kk < a.K) {
                    if constexpr (F24) xf[mt][s].load(xrow + kk, a.x_lo + mc * (long)LPK + kk, serow ? serow + kk : nullptr);
                    else xf[mt][s].load(xrow + kk, serow ? serow + kk : nullptr);
                } else xf[mt][s].zero();
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long m = mbase + mt * 32;
            const bool m_ok = m < a.M;
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                if (t < nt) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KSM; ++s)
                        if (s < KS) {
                            const bf16x8 wh = wl[((t * KS + s) * NS) * 64 + lane];
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[mt][s].hi, acc, 0, 0, 0);
                            if constexpr (SPLIT) {
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[mt][s].lo, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[((t * KS + s) * NS + 1) * 64 + lane], xf[mt][s].hi, acc, 0, 0, 0);
                            }
                        }
                    if (m_ok) {
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const int cl = t * 32 + 16 * h + 8 * g;          // channel within this block's tiles
                            const int c = t0 * 32 + cl;
                            if (c < cout) {
                                Vec8f v;
                                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + cl);
                                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + cl + 4);
#pragma unroll
                                for (int i = 0; i < 4; ++i) { v.v[i] = acc[8 * g + i] + b0[i]; v.v[4 + i] = acc[8 * g + 4 + i] + b1[i]; }
                                if constexpr (ACT == FR_ACT_PRELU) {
                                    prelu8(v, *reinterpret_cast<const f32x4*>(bl + NTB * 32 + cl), *reinterpret_cast<const f32x4*>(bl + NTB * 32 + cl + 4));
                                } else if constexpr (ACT == FR_ACT_HSWISH) {
#pragma unroll
                                    for (int i = 0; i < 8; ++i) v.v[i] = v.v[i] * fminf(fmaxf(fmaf(v.v[i], 1.f / 6.f, 0.5f), 0.f), 1.f);
                                } else if constexpr (ACT == FR_ACT_RELU) {
#pragma unroll
                                    for (int i = 0; i < 8; ++i) v.v[i] = relu1(v.v[i]);
                                }
                                const long off = m * (long)cout + c;
                                if (R) {
                                    Vec8f r;
                                    if constexpr (F24) r = f24_load8(R + off, a.e.residual_lo + m * (long)LPC + c);
                                    else r = load8(R + off);
#pragma unroll
                                    for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
                                }
                                if constexpr (F24) {
                                    const int col = cl - (t / SG) * SG * 32;
                                    u32x4 hh; unsigned l0, l1;
                                    f24_encode(v, hh, l0, l1);
                                    *reinterpret_cast<u32x4*>(stage + p * PITCH + col * 2) = hh;
                                    *reinterpret_cast<unsigned long long*>(stage_lo + p * PITCH_LO + col) = (unsigned long long)l0 | ((unsigned long long)l1 << 32);
                                } else if constexpr (OUT32) stage_put_f32(stage, PITCH, p, cl - (t / SG) * SG * 32, v);
                                else stage_put(stage, PITCH, p, cl - (t / SG) * SG * 32, v);
                            }
                        }
                    }
                    if constexpr (STAGED) {
                        if ((t % SG) == SG - 1 || t == nt - 1) {               // group complete: write it out
                            const int tg = (t / SG) * SG;
                            const long tile0 = m - p;                          // first pixel of this wave's tile
                            const long left = a.M - tile0;
                            if (left > 0) {
                                if constexpr (OUT32)
                                    stage_flush_f32(stage, PITCH, lane, reinterpret_cast<float*>(Y) + tile0 * cout + (t0 + tg) * 32, cout,
                                                    min(SG * 32, cout - (t0 + tg) * 32), left < 32 ? (int)left : 32);
                                else
                                    stage_flush(stage, PITCH, lane, reinterpret_cast<bf16*>(Y) + tile0 * cout + (t0 + tg) * 32, cout,
                                                min(SG * 32, cout - (t0 + tg) * 32), left < 32 ? (int)left : 32);
                                if constexpr (F24)
                                    stage_flush_lo(stage_lo, PITCH_LO, lane, a.e.y_lo + tile0 * LPC + (t0 + tg) * 32, LPC,
                                                   min(SG * 32, cout - (t0 + tg) * 32), left < 32 ? (int)left : 32);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- small-M GEMM: SE fc1/fc2 on the pooled [batch, C] vectors, the final Linear(+BN1d), the
// SphereFace fc.  M is only the batch, so the pixel-tiled kernels above would launch a handful of
// blocks; here one block owns a 32-row x 32-column output tile, its four waves split K, keep eight
// k-steps of loads in flight each, and reduce through LDS.  fp32 output; fp32 input is split hi+lo.
template <typename TIn>
__global__ __launch_bounds__(256) void fc_kernel(PwArgs a) {
    constexpr int NS = std::is_same<TIn, float>::value ? 2 : 1;
    __shared__ float red[3][16][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const long m = (long)blockIdx.x * 32 + p;
    const bool m_ok = m < a.M;
    const int t = blockIdx.y;
    const int ks_per = (a.KS + 3) >> 2;
    const int s_begin = wave * ks_per, s_end = min(a.KS, s_begin + ks_per);
    const int xt = std::is_same<TIn, float>::value ? a.x_tiles : 0;      // > 0: rows are sums of xt partial rows
    const TIn* xrow = reinterpret_cast<const TIn*>(a.x) + (m_ok ? m : a.M - 1) * (long)a.K * (xt > 0 ? xt : 1);
    const bf16x8* W = reinterpret_cast<const bf16x8*>(a.w) + (size_t)t * a.KS * NS * 64 + lane;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int sg = s_begin; sg < s_end; sg += 8) {
        XFrag<TIn> xf[8];
        bf16x8 wf[8][NS];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = sg + j, kk = s * 16 + 8 * h;
            if (s < s_end && kk < a.K) {
                if constexpr (std::is_same<TIn, float>::value) {
                    if (xt > 0) {
                        // fold the pooling partials here (tile order, then the 1/(H*W) scale: the arithmetic the
                        // separate reduce launch performed), so squeeze-excite is two launches instead of three
                        Vec8f sum;
#pragma unroll
                        for (int i = 0; i < 8; ++i) sum.v[i] = 0.f;
                        for (int q = 0; q < xt; ++q) {
                            const Vec8f v = load8(xrow + (long)q * a.K + kk);
#pragma unroll
                            for (int i = 0; i < 8; ++i) sum.v[i] += v.v[i];
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) sum.v[i] *= a.x_scale;
                        split8(sum, xf[j].hi, xf[j].lo);
                    } else {
                        xf[j].load(xrow + kk, nullptr);
                    }
                } else {
                    xf[j].load(xrow + kk, nullptr);
                }
            } else {
                xf[j].zero();
            }
            const int sc = s < s_end ? s : s_end - 1;
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) wf[j][sp] = W[((size_t)sc * NS + sp) * 64];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (sg + j < s_end) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][0], xf[j].hi, acc, 0, 0, 0);
                if constexpr (NS == 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][0], xf[j].lo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][1], xf[j].hi, acc, 0, 0, 0);
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)                       // fixed order: deterministic
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[w][r][lane];
        epilogue_store<float>(acc, a.e, m, 32 * t + 16 * h, m_ok);
    }
}

int launch_fc(bool in_f32, const PwArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.K <= 0 || a.K % 8 || a.e.cout % 8 || a.KS != cdiv(a.K, 16) || a.NT != cdiv(a.e.cout, 32) || a.se || a.e.residual)
        FR_FAIL(FR_ERR_ARG, "fc: inconsistent shapes");
    const long mt = (a.M + 31) / 32;
    if (mt > 0x7FFFFFFFL || a.NT > 65535) FR_FAIL(FR_ERR_ARG, "fc: grid too large");
    dim3 grid((unsigned)mt, (unsigned)a.NT);
    if (in_f32) hipLaunchKernelGGL((fc_kernel<float>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((fc_kernel<bf16>), grid, dim3(256), 0, st, a);
    FR_LAUNCH_CHECK("fc_kernel");
    return FR_OK;
}

// ---- long-K Linear (the SphereFace fc: 25088 -> 512): split K over the grid --------------------------------------
// fc_kernel gives one block a 32 x 32 output tile and the whole K: 128 blocks for a 256-image sub-batch, each wave
// walking 392 k-steps of dependent loads, and every weight tile fetched once per 32 rows (0.076 ms per launch, 2 % of
// a sphere64 step).  Here a block owns 64 rows x 64 columns and one of S slices of K, its four waves split the slice,
// keep four k-steps (16 fragment loads) in flight each and reduce through LDS; partial sums [S][M][N] go to a
// workspace slot and fc_splitk_reduce_kernel adds them in slice order and applies the epilogue.  S and the wave
// split depend on K only, so a row's result does not depend on the batch it is in.
template <typename TIn>
__global__ __launch_bounds__(256) void fc_splitk_kernel(PwArgs a, float* __restrict__ partial, int S) {
    constexpr int NS = std::is_same<TIn, float>::value ? 2 : 1;
    __shared__ float red[3][4][16][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const long mb = (long)blockIdx.x * 64;
    const int t0 = blockIdx.y * 2, z = blockIdx.z;
    const int ks_z = (a.KS + S - 1) / S, zb = z * ks_z, ze = min(a.KS, zb + ks_z);
    const int ks_w = (ks_z + 3) >> 2;
    const int s_begin = zb + wave * ks_w, s_end = min(ze, s_begin + ks_w);
    const TIn* xrow[2];
    const bf16x8* W[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) xrow[j] = reinterpret_cast<const TIn*>(a.x) + min(mb + 32 * j + p, a.M - 1) * (long)a.K;
#pragma unroll
    for (int i = 0; i < 2; ++i) W[i] = reinterpret_cast<const bf16x8*>(a.w) + (size_t)min(t0 + i, a.NT - 1) * a.KS * NS * 64 + lane;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int sg = s_begin; sg < s_end; sg += 4) {
        XFrag<TIn> xf[4][2];
        bf16x8 wf[4][2][NS];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = sg + q, kk = s * 16 + 8 * h;
            const int sc = s < s_end ? s : s_end - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) wf[q][i][sp] = W[i][((size_t)sc * NS + sp) * 64];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (s < s_end && kk < a.K) xf[q][j].load(xrow[j] + kk, nullptr);
                else xf[q][j].zero();
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (sg + q < s_end) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q][i][0], xf[q][j].hi, acc[i][j], 0, 0, 0);
                        if constexpr (NS == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q][i][0], xf[q][j].lo, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q][i][1], xf[q][j].hi, acc[i][j], 0, 0, 0);
                        }
                    }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][2 * i + j][r][lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int w = 0; w < 3; ++w)                   // fixed order: deterministic
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[w][2 * i + j][r][lane];
                const long m = mb + 32 * j + p;
                const int c0 = 32 * (t0 + i) + 16 * h;
                if (m < a.M && t0 + i < a.NT) {
                    float* dst = partial + ((size_t)z * a.M + m) * a.e.cout + c0;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (c0 + 4 * g < a.e.cout)
                            *reinterpret_cast<f32x4*>(dst + 4 * g) = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                }
            }
    }
}

// y[m][c] = act(sum_z partial[z][m][c] + bias[c]), slices added in order (fp32 output, as epilogue_store<float>)
__global__ __launch_bounds__(256) void fc_splitk_reduce_kernel(const float* __restrict__ partial, EpiArgs e, long M, int S) {
    const long n4 = M * (long)(e.cout >> 2);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % (e.cout >> 2)) * 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(partial) + i;
    f32x4 v = src[0];
    for (int z = 1; z < S; ++z) { const f32x4 t = src[(size_t)z * n4]; v += t; }
    const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q] + b[q], e.act, 0.f);
    reinterpret_cast<f32x4*>(e.y)[i] = v;
}

// K slices of the split-K Linear for a reduction length of K (0: use fc_kernel).  A function of K only.
int fc_splitk_slices(int K) {
    const char* es = getenv("FR_FC_SPLITK");              // tuning, read when a plan is created: 0 = off
    const int env = es ? atoi(es) : -1;
    if (env == 0 || K < 8192) return 0;
    return env > 0 ? std::min(env, 32) : 8;
}

int launch_fc_splitk(bool in_f32, const PwArgs& a, float* partial, int S, hipStream_t st) {
    if (a.M <= 0 || a.K <= 0 || a.K % 8 || a.e.cout % 8 || a.KS != cdiv(a.K, 16) || a.NT != cdiv(a.e.cout, 32) || a.se || a.e.residual ||
        a.e.slope || !partial || S < 1 || S > 32 || a.e.cout % 4)
        FR_FAIL(FR_ERR_ARG, "fc (split K): inconsistent shapes");
    const long mt = (a.M + 63) / 64;
    if (mt > 0x7FFFFFFFL || a.NT > 65535) FR_FAIL(FR_ERR_ARG, "fc (split K): grid too large");
    dim3 grid((unsigned)mt, (unsigned)cdiv(a.NT, 2), (unsigned)S);
    if (in_f32) hipLaunchKernelGGL((fc_splitk_kernel<float>), grid, dim3(256), 0, st, a, partial, S);
    else hipLaunchKernelGGL((fc_splitk_kernel<bf16>), grid, dim3(256), 0, st, a, partial, S);
    FR_LAUNCH_CHECK("fc_splitk_kernel");
    const long n4 = a.M * (long)(a.e.cout >> 2);
    hipLaunchKernelGGL(fc_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)partial, a.e, a.M, S);
    FR_LAUNCH_CHECK("fc_splitk_reduce_kernel");
    return FR_OK;
}

void pw_pick_tiling(bool in_f32, int NT, int KS, int* ntb, int* chunk_ks) {
    const int ns = in_f32 ? 2 : 1;
    if (in_f32 && KS <= 8 && !(getenv("FR_PW_F32_STREAM") && atoi(getenv("FR_PW_F32_STREAM")) == 0)) {
        // fp32 plans, K <= 128: the streaming kernel with the whole hi+lo weight slab of the block's n-tiles resident
        // (n-tiles x k-steps x 2 KB), no k-chunk restaging
        int b = 1;
        // (b may exceed NT: 3 n-tiles run as ONE block of 4 rather than as column groups of 2 + 1 that both read the
        // activations -- 24 -> 72 channels on the 56x56 maps: 0.140 -> 0.126 ms per 512 images in f24 plans)
        while (b < NT && b * 2 <= 4 && b * 2 * KS * 2 <= 32) b *= 2;      // <= 32 KB of weights + 35 KB of output staging: two blocks per CU
        *ntb = b;
        *chunk_ks = KS;
        return;
    }
    // bf16, 128 < K <= 256: two n-tiles per block keep the whole weight slab (<= 32 KB) resident,
    // so the streaming kernel applies (no k-chunk restaging); the activations are re-read from L2
    // by the extra column groups.
    // measured (bf16, 512-image launches): with 3..8 k-steps four n-tiles per block lose 10-27 % against two (the
    // 40->120/240 and 80->184/200 expansions of mobilenetv3_large); with 1-2 k-steps whole rows per block still win
    int cap = (!in_f32 && KS > 2 && KS <= 16) ? 2 : 4;
    static const int cap_env = getenv("FR_PW_NTB") ? atoi(getenv("FR_PW_NTB")) : 0;        // tuning: cap on n-tiles per block
    if (cap_env > 0 && !in_f32) cap = std::min(cap, cap_env);
    int b = 1;
    while (b < NT && b < cap) b *= 2;
    int ck = 48 / (b * ns);           // <= 48 KB of LDS
    if (ck > KS) ck = KS;
    if (ck < 1) ck = 1;
    *ntb = b;
    *chunk_ks = ck;
}

template <typename TIn, typename TOut, int NTB, int SG>
static int launch_pw_stream_sg(const PwArgs& a, hipStream_t st) {
    constexpr int NS = kWide<TIn> ? 2 : 1;
    // pixel tiles per wave: 8 operand fragments per lane in flight for bf16, 4 (hi + lo each) for fp32 up to K = 64
    int mt = NS == 2 ? (a.KS <= 2 ? 2 : 1) : (a.KS <= 2 ? 4 : (a.KS <= 4 ? 2 : (a.KS <= 8 ? 1 : 0)));
    static const int mt_env = getenv("FR_PW_MT") ? atoi(getenv("FR_PW_MT")) : 0;           // tuning: cap on pixel tiles per wave
    if (mt_env > 0 && mt > mt_env) mt = mt_env;
    const long per_iter = 128L * (mt ? mt : 1);
    const long n_iter = (a.M + per_iter - 1) / per_iter;
    // ~2048 blocks in total: with many column groups each block walks several pixel tiles, so that the
    // weight slab it staged is amortised (one 128-pixel tile per block re-staged 20 KB of weights for
    // 16 KB of activations in the 160->960 layers)
    const int ng = cdiv(a.NT, NTB);
    static const int total_blocks = getenv("FR_PW_BLOCKS") ? atoi(getenv("FR_PW_BLOCKS")) : 2048;
    const long want = std::max<long>(8, cdiv(total_blocks, ng));
    const PwGrid pg{(int)std::min<long>(n_iter, want), ng};
    dim3 grid(pw_grid_blocks(pg.gx, pg.ng));
    size_t lds = (size_t)NTB * a.KS * NS * 1024 + (size_t)NTB * 256;
    lds += 4 * 32 * (size_t)stage_pitch(SG * 32 * kHiBytes<TOut>);
    if (kIsF24<TOut>) lds += 4 * 32 * (size_t)stage_pitch_lo(SG * 32);
#define FR_PW_GO1(MT_, KSM_, ACT_) do { FR_BIG_LDS((pw_stream_kernel<TIn, TOut, NTB, MT_, KSM_, SG, ACT_>), lds); \
        hipLaunchKernelGGL((pw_stream_kernel<TIn, TOut, NTB, MT_, KSM_, SG, ACT_>), grid, dim3(256), lds, st, a, pg); } while (0)
#define FR_PW_GO(MT_, KSM_) do { switch (a.e.act) { \
        case FR_ACT_NONE: FR_PW_GO1(MT_, KSM_, FR_ACT_NONE); break; \
        case FR_ACT_PRELU: FR_PW_GO1(MT_, KSM_, FR_ACT_PRELU); break; \
        case FR_ACT_HSWISH: FR_PW_GO1(MT_, KSM_, FR_ACT_HSWISH); break; \
        case FR_ACT_RELU: FR_PW_GO1(MT_, KSM_, FR_ACT_RELU); break; \
        default: FR_FAIL(FR_ERR_UNSUPPORTED, "pw: activation with no 1x1 kernel"); } } while (0)
    if constexpr (NS == 2) {
        if (mt == 2) FR_PW_GO(2, 2);
        else if (a.KS <= 4) FR_PW_GO(1, 4);
        else FR_PW_GO(1, 8);
    } else {
        if (mt == 4) FR_PW_GO(4, 2);
        else if (mt == 2) FR_PW_GO(2, 4);
        else if (mt == 1) FR_PW_GO(1, 8);
        else FR_PW_GO(1, 16);
    }
#undef FR_PW_GO
#undef FR_PW_GO1
    FR_LAUNCH_CHECK("pw_stream_kernel");
    return FR_OK;
}

template <typename TIn, typename TOut, int NTB>
static int launch_pw_stream(const PwArgs& a, hipStream_t st) {
    constexpr int SG2 = NTB < 2 ? NTB : 2;
    if constexpr (std::is_same<TOut, bf16>::value)       // (fp32 tiles of 4 n-tiles would take 67 KB of LDS: two at a time, 256-byte runs)
        if (NTB > 2 && a.NT <= NTB) return launch_pw_stream_sg<TIn, TOut, NTB, NTB>(a, st);     // whole rows in one block
    if constexpr (std::is_same<TOut, float>::value && NTB >= 2) {
        // fp32 outputs of a layer with two or more column groups (the expansion convs) are staged ONE n-tile (128-byte rows) at a time:
        // 16 KB less LDS per block -- three blocks per CU beside the hi + lo weight slab instead of two -- for half as long write-back
        // runs.  Measured per layer (mobilenetv3_large f32, 512 images): 112->672 0.1245 -> 0.105 ms, 80->480 0.073 -> 0.057, 80->200 0.040 ->
        // 0.032, 40->240 0.1045 -> 0.1016; layers whose block covers whole output rows lose (24->72 at 56x56: 0.120 -> 0.154, 120->40 with
        // squeeze-excite scale and residual: 0.084 -> 0.089) and keep two n-tiles.  FR_PW_F32_SG1: 0 never, 2 always (sweeps).
        static const int sg1 = getenv("FR_PW_F32_SG1") ? atoi(getenv("FR_PW_F32_SG1")) : 1;
        if (sg1 == 2 || (sg1 == 1 && cdiv(a.NT, NTB) >= 2)) return launch_pw_stream_sg<TIn, TOut, NTB, 1>(a, st);
    }
    return launch_pw_stream_sg<TIn, TOut, NTB, SG2>(a, st);
}

template <typename TIn, typename TOut>
static int launch_pw_t(const PwArgs& a, int ntb, hipStream_t st) {
    if constexpr (std::is_same<TIn, bf16>::value) {
        // (the streaming kernel is compiled per activation; a hard sigmoid behind a 1x1 conv -- no reference net has one -- takes pw_gemm_kernel)
        if (a.KS <= 16 && a.chunk_ks >= a.KS && (a.KS <= 8 || ntb <= 2) && a.e.act != FR_ACT_HSIGMOID) {
            switch (ntb) {
                case 1: return launch_pw_stream<TIn, TOut, 1>(a, st);
                case 2: return launch_pw_stream<TIn, TOut, 2>(a, st);
                case 4: return launch_pw_stream<TIn, TOut, 4>(a, st);
                default: break;
            }
        }
    } else if constexpr (kWide<TOut>) {
        if (a.KS <= 8 && a.chunk_ks >= a.KS && (a.e.act != FR_ACT_HSIGMOID || kIsF24<TIn>)) {
            switch (ntb) {
                case 1: return launch_pw_stream<TIn, TOut, 1>(a, st);
                case 2: return launch_pw_stream<TIn, TOut, 2>(a, st);
                case 4: return launch_pw_stream<TIn, TOut, 4>(a, st);
                default: break;
            }
        }
    }
    if constexpr (kIsF24<TIn> || kIsF24<TOut>) {
        FR_FAIL(FR_ERR_UNSUPPORTED, "pw: f24 plans have no k-chunked 1x1 kernel for this shape (K <= 128, or K > 128 with H*W >= 43 behind squeeze-excite)");
    } else {
    constexpr int NS = std::is_same<TIn, float>::value ? 2 : 1;
    const long n_super = (a.M + 127) / 128;
    const PwGrid pg{(int)std::min<long>(n_super, 2048), cdiv(a.NT, ntb)};
    dim3 grid(pw_grid_blocks(pg.gx, pg.ng));
    size_t lds = (size_t)std::min(ntb, a.NT) * a.chunk_ks * NS * 1024;
    PwArgs b = a;
    b.stage_off = (int)lds;
    if (std::is_same<TOut, bf16>::value) lds += 4 * 32 * (size_t)stage_pitch(std::min(ntb, 2) * 64);
    switch (ntb) {
        case 1: FR_BIG_LDS((pw_gemm_kernel<TIn, TOut, 1>), lds); hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 1>), grid, dim3(256), lds, st, b, pg); break;
        case 2: FR_BIG_LDS((pw_gemm_kernel<TIn, TOut, 2>), lds); hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 2>), grid, dim3(256), lds, st, b, pg); break;
        case 4: FR_BIG_LDS((pw_gemm_kernel<TIn, TOut, 4>), lds); hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 4>), grid, dim3(256), lds, st, b, pg); break;
        default: FR_FAIL(FR_ERR_ARG, "pw: bad n-tile blocking");
    }
    FR_LAUNCH_CHECK("pw_gemm_kernel");
    return FR_OK;
    }
}

// Which kernel launch_pw picks for a layer shape (diagnostics: fr_plan_layer_kernel, bench.py).
const char* pw_kernel_name(int in_fmt, int out_fmt, int K, int cout, bool se, int HW) {
    const bool in_w = in_fmt != FMT_BF16, out_w = out_fmt != FMT_BF16;
    const int KS = cdiv(K, 16), NT = cdiv(cout, 32);
    int ntb, ck;
    pw_pick_tiling(in_w, NT, KS, &ntb, &ck);
    if (in_w && out_w && KS > 8 && (!se || HW >= 43) && !(getenv("FR_PW_F32_PROJ") && atoi(getenv("FR_PW_F32_PROJ")) == 0)) return "pw_proj_kernel";
    if (in_w) return (out_w && KS <= 8 && ck >= KS) ? "pw_stream_kernel" : "pw_gemm_kernel";
    if (!out_w && KS > 16 && (!se || HW >= 43)) return "pw_proj_kernel";
    if (KS <= 16 && ck >= KS && (KS <= 8 || ntb <= 2)) return "pw_stream_kernel";
    return "pw_gemm_kernel";
}

int launch_pw(int in_fmt, int out_fmt, const PwArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.K <= 0 || a.K % 8 || a.e.cout % 8 || a.KS != cdiv(a.K, 16) || a.NT != cdiv(a.e.cout, 32))
        FR_FAIL(FR_ERR_ARG, "pw: inconsistent shapes (K and Cout must be multiples of 8)");
    const bool in_w = in_fmt != FMT_BF16;
    int ntb, ck;
    pw_pick_tiling(in_w, a.NT, a.KS, &ntb, &ck);
    if (ck != a.chunk_ks) FR_FAIL(FR_ERR_ARG, "pw: chunk_ks does not match pw_pick_tiling");
    static const bool proj32 = !(getenv("FR_PW_F32_PROJ") && atoi(getenv("FR_PW_F32_PROJ")) == 0);
    const bool proj_ok = !a.se || (a.HW >= 43 && a.M % a.HW == 0);
    if (in_fmt == FMT_F24 || out_fmt == FMT_F24) {
        if (in_fmt != out_fmt) FR_FAIL(FR_ERR_UNSUPPORTED, "pw: f24 input and output go together");
        if (!a.x_lo || !a.e.y_lo || (a.e.residual && !a.e.residual_lo)) FR_FAIL(FR_ERR_ARG, "pw: f24 tensors without their lo planes");
        if (a.KS > 8 && proj_ok) return launch_pw_proj<f24>(a, st);
        return launch_pw_t<f24, f24>(a, ntb, st);
    }
    const bool in_f32 = in_fmt == FMT_F32, out_f32 = out_fmt == FMT_F32;
    if (in_f32 && out_f32 && proj32 && a.KS > 8 && proj_ok) return launch_pw_proj<float>(a, st);
    if (in_f32 && out_f32) return launch_pw_t<float, float>(a, ntb, st);
    if (!in_f32 && !out_f32 && a.KS > 16 && proj_ok) return launch_pw_proj<bf16>(a, st);
    if (!in_f32 && !out_f32) return launch_pw_t<bf16, bf16>(a, ntb, st);
    if (!in_f32 && out_f32) return launch_pw_t<bf16, float>(a, ntb, st);
    FR_FAIL(FR_ERR_UNSUPPORTED, "pw: fp32 input with bf16 output is not instantiated");
}
