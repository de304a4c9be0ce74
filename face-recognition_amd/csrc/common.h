// Shared device/host helpers for libfr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <type_traits>

#include "../../include/fr_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ---- host side -------------------------------------------------------------------------------
void fr_set_error(const std::string& msg);
#define FR_FAIL(code, msg) do { fr_set_error(msg); return (code); } while (0)
#define FR_HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fr_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return FR_ERR_HIP; } } while (0)
#define FR_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    fr_set_error(std::string("launch ") + name + ": " + hipGetErrorString(e_)); return FR_ERR_HIP; } } while (0)

// Kernels that may take more than the default 64 KB of dynamic LDS: raise the kernel's limit ONCE per (kernel
// instantiation, device) to the hardware maximum (160 KB), under a mutex -- never per launch and never to the launch's own
// size (one instantiation serves layers with different sizes: a second host thread could lower it between another
// thread's set and launch).  The limit does not affect occupancy, the launch's actual size does.
int fr_big_lds_once(const void* kernel);
#define FR_BIG_LDS_ONCE(kernel, bytes) do { if ((size_t)(bytes) > 65536) { int rc_ = fr_big_lds_once((const void*)(kernel)); if (rc_) return rc_; } } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// host fp32 -> bf16 bits, round to nearest even
static inline uint16_t host_f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float host_bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- device side -----------------------------------------------------------------------------
#define FR_DEV __device__ __forceinline__

FR_DEV float bfbits_to_f(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }

// Activation element access: T = bf16 (2 B) or float (4 B).  "Vec8" = 8 consecutive channels.
struct Vec8f { float v[8]; };

FR_DEV Vec8f load8(const bf16* p) {
    u32x4 r = *reinterpret_cast<const u32x4*>(p);
    Vec8f o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o.v[2 * i] = __uint_as_float(r[i] << 16);
        o.v[2 * i + 1] = __uint_as_float(r[i] & 0xFFFF0000u);
    }
    return o;
}
FR_DEV Vec8f load8(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    Vec8f o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o.v[i] = a[i]; o.v[4 + i] = b[i]; }
    return o;
}
FR_DEV unsigned pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 t = { (bf16)lo, (bf16)hi };
    return __builtin_bit_cast(unsigned, t);
}
FR_DEV void store8(bf16* p, const Vec8f& x) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2bf(x.v[2 * i], x.v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = r;
}
FR_DEV void store8(float* p, const Vec8f& x) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = x.v[i]; b[i] = x.v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

// ---- "f24" activation storage (FR_PREC_F24): the top 24 bits of the fp32 pattern (sign, 8 exponent bits, 15 stored
// mantissa bits = 16 significant bits, what the split-bf16 GEMMs of the fp32 plans carry anyway), rounded to nearest, in
// TWO PLANES: hi = the top 16 bits [M][C] -- a bf16 tensor (truncated, not rounded) in exactly the layout of the bf16
// plans, and directly the hi operand of the split-bf16 MFMA products -- and lo = the next 8 bits as bytes
// [M][f24_lo_pitch(C)] (rows padded to 16 B so that every row starts 16-byte aligned for LDS-DMA and vector stores).
// 3 B per element instead of 4: the fp32 plans are byte-bound.  x - hi is exact in bf16 (8 bits).
struct f24 {};
template <typename T> struct act_hi { typedef T type; };
template <> struct act_hi<f24> { typedef bf16 type; };
template <typename T> constexpr bool kIsF24 = std::is_same<T, f24>::value;
template <typename T> constexpr bool kWide = std::is_same<T, f24>::value || std::is_same<T, float>::value;   // fp32-class arithmetic (split operands)
template <typename T> constexpr int kHiBytes = (int)sizeof(typename act_hi<T>::type);                         // bytes per element in the (hi) plane
__host__ __device__ constexpr int f24_lo_pitch(int C) { return (C + 15) & ~15; }
__host__ __device__ constexpr size_t f24_plane_bytes(size_t elems) { return elems * 2; }   // hi plane of a buffer; the lo plane follows it directly (elems % 8 == 0: 16-byte aligned)

FR_DEV float f24_round(float v) { return __uint_as_float((__float_as_uint(v) + 0x80u) & 0xFFFFFF00u); }
// 8 channels: hi = 4 dwords (8 x 16 bits), lo = 2 dwords (8 bytes)  ->  fp32 (exact)
FR_DEV Vec8f f24_decode(const u32x4& hi, unsigned lo0, unsigned lo1) {
    Vec8f o;
    o.v[0] = __uint_as_float(__builtin_amdgcn_perm(hi[0], lo0, 0x0504000cu));
    o.v[1] = __uint_as_float(__builtin_amdgcn_perm(hi[0], lo0, 0x0706010cu));
    o.v[2] = __uint_as_float(__builtin_amdgcn_perm(hi[1], lo0, 0x0504020cu));
    o.v[3] = __uint_as_float(__builtin_amdgcn_perm(hi[1], lo0, 0x0706030cu));
    o.v[4] = __uint_as_float(__builtin_amdgcn_perm(hi[2], lo1, 0x0504000cu));
    o.v[5] = __uint_as_float(__builtin_amdgcn_perm(hi[2], lo1, 0x0706010cu));
    o.v[6] = __uint_as_float(__builtin_amdgcn_perm(hi[3], lo1, 0x0504020cu));
    o.v[7] = __uint_as_float(__builtin_amdgcn_perm(hi[3], lo1, 0x0706030cu));
    return o;
}
// fp32 -> rounded f24 planes of 8 channels
FR_DEV void f24_encode(const Vec8f& x, u32x4& hi, unsigned& lo0, unsigned& lo1) {
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = __float_as_uint(x.v[i]) + 0x80u;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[i] = __builtin_amdgcn_perm(u[2 * i + 1], u[2 * i], 0x07060302u);
    lo0 = __builtin_amdgcn_perm(u[1], u[0], 0x0c0c0501u) | __builtin_amdgcn_perm(u[3], u[2], 0x05010c0cu);
    lo1 = __builtin_amdgcn_perm(u[5], u[4], 0x0c0c0501u) | __builtin_amdgcn_perm(u[7], u[6], 0x05010c0cu);
}
// MFMA operands of 8 stored channels: hi fragment = the hi plane as it is, lo = x - hi (exact in bf16)
FR_DEV void f24_frags(const u32x4& hi, unsigned lo0, unsigned lo1, bf16x8& fh, bf16x8& fl) {
    const Vec8f x = f24_decode(hi, lo0, lo1);
    u32x4 l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x.v[2 * i] - __uint_as_float(hi[i] << 16);
        const float b = x.v[2 * i + 1] - __uint_as_float(hi[i] & 0xffff0000u);
        l[i] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    }
    fh = __builtin_bit_cast(bf16x8, hi);
    fl = __builtin_bit_cast(bf16x8, l);
}
// the same split from fp32 values that are ALREADY f24-rounded (fused kernels: a result that the unfused plan would
// have stored and re-loaded): hi = truncation, lo = x - hi.  Bit-identical to f24_frags(f24_encode(x)).
FR_DEV void f24_split8(const Vec8f& x, bf16x8& fh, bf16x8& fl) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u0 = __float_as_uint(x.v[2 * i]), u1 = __float_as_uint(x.v[2 * i + 1]);
        h[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        const float a = x.v[2 * i] - __uint_as_float(u0 & 0xffff0000u);
        const float b = x.v[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
        l[i] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    }
    fh = __builtin_bit_cast(bf16x8, h);
    fl = __builtin_bit_cast(bf16x8, l);
}
// global-memory forms: hi plane element pointer (bf16 layout) + lo plane byte pointer of the same 8 channels
FR_DEV Vec8f f24_load8(const bf16* hi, const unsigned char* lo) {
    const u32x4 h = *reinterpret_cast<const u32x4*>(hi);
    const unsigned long long l = *reinterpret_cast<const unsigned long long*>(lo);
    return f24_decode(h, (unsigned)l, (unsigned)(l >> 32));
}
FR_DEV void f24_store8(bf16* hi, unsigned char* lo, const Vec8f& x) {
    u32x4 h; unsigned l0, l1;
    f24_encode(x, h, l0, l1);
    *reinterpret_cast<u32x4*>(hi) = h;
    *reinterpret_cast<unsigned long long*>(lo) = (unsigned long long)l0 | ((unsigned long long)l1 << 32);
}

FR_DEV float apply_act(float v, int act, float slope) {
    switch (act) {
        case FR_ACT_PRELU:    return v >= 0.f ? v : v * slope;
        case FR_ACT_HSWISH:   return v * fminf(fmaxf(fmaf(v, 1.f / 6.f, 0.5f), 0.f), 1.f);       // x * relu6(x + 3) / 6 as fma + med3 + mul (2 packed + 1 VALU per pair less)
        case FR_ACT_RELU:     return fmaxf(v, 0.f);
        case FR_ACT_HSIGMOID: return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
        default:              return v;
    }
}

// max(v, 0) as the single v_max_f32 it is (fmaxf also emits a canonicalising v_max v,v,v in front).
FR_DEV float relu1(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}

// Activation of 8 channels with ONE uniform branch per vector (apply_act per element makes the
// compiler emit a switch per element).
// PReLU with the slopes in registers.  Never write `v >= 0 ? v : v * slope[j]` with `slope` a pointer: the load
// is then conditional, and the compiler emits a compare, an exec-masked branch and a dependent load PER ELEMENT
// (measured: 10 k of 16 k cycles per block in the fused expand phase, 10-15 % of the 1x1 kernels with PReLU).
FR_DEV void prelu8(Vec8f& v, const f32x4& s0, const f32x4& s1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v.v[j] = v.v[j] >= 0.f ? v.v[j] : v.v[j] * s0[j];
        v.v[4 + j] = v.v[4 + j] >= 0.f ? v.v[4 + j] : v.v[4 + j] * s1[j];
    }
}

// slope8: 16-byte aligned, read as two vectors up front when act == FR_ACT_PRELU (unused otherwise).
FR_DEV void act8(Vec8f& v, int act, const float* slope8) {
    if (act == FR_ACT_PRELU) {
        prelu8(v, *reinterpret_cast<const f32x4*>(slope8), *reinterpret_cast<const f32x4*>(slope8 + 4));
    } else if (act == FR_ACT_HSWISH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = v.v[j] * fminf(fmaxf(fmaf(v.v[j], 1.f / 6.f, 0.5f), 0.f), 1.f);
    } else if (act == FR_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = relu1(v.v[j]);
    } else if (act == FR_ACT_HSIGMOID) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = fminf(fmaxf(v.v[j] + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
}

// n / d for n*d < 2^32 with a per-block precomputed multiplier (loops index LDS tiles with it).
struct FastDiv {
    unsigned long long mul; unsigned d;
    FR_DEV explicit FastDiv(unsigned dd) : mul((1ull << 32) / dd + 1ull), d(dd) {}
    FR_DEV unsigned div(unsigned n) const { return (unsigned)(((unsigned long long)n * mul) >> 32); }
    FR_DEV void divmod(unsigned n, unsigned& q, unsigned& r) const { q = div(n); r = n - q * d; }
};

// Split 8 fp32 values into bf16 hi and lo fragments: x ~= hi + lo with ~16 mantissa bits.
FR_DEV void split8(const Vec8f& x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bf16 h = (bf16)x.v[i];
        hi[i] = h;
        lo[i] = (bf16)(x.v[i] - (float)h);
    }
}
FR_DEV bf16x8 to_frag(const Vec8f& x) {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16)x.v[i];
    return f;
}

// MFMA 32x32x16 output-row permutation used by every GEMM-shaped kernel here: the weight
// ("A") row i of n-tile t holds output channel  32t + 16*((i>>2)&1) + (i&3) + 4*(i>>3), so that
// lane half h (lane>>5), accumulator register r holds channel 32t + 16h + r: 16 consecutive
// channels per lane, i.e. wide NHWC stores.
static inline int mfma_row_to_cout(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

// ---- coalesced output through a wave-private LDS tile -------------------------------------------
// The MFMA epilogue leaves each lane with 16 channels (32 B) of ONE pixel and its neighbour lane with
// another pixel, so a direct store instruction writes isolated 16-byte pieces; measured on MI355X
// that pattern tops out at 3.2-3.6 TB/s where line-contiguous stores reach 5.4 TB/s
// (tools/micro/storepat.hip, storepat2.hip; trading halves between lanes with DPP so that lane pairs
// write 32 B recovered only a third of the gap).  The bf16 kernels therefore put a
// [32 pixels][<= 64 channels] tile into LDS and write it back as contiguous 16-byte chunks.
__host__ __device__ constexpr int stage_pitch(int row_bytes) { return ((row_bytes + 127) & ~127) + 16; }   // conflict-free b128 rows
FR_DEV void stage_put(unsigned char* stage, int pitch, int row, int col, const Vec8f& v) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2bf(v.v[2 * i], v.v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(stage + row * pitch + col * 2) = r;
}
// Write rows [0,rows) x ncol channels of the staged tile to y (row stride ystride elements).
FR_DEV void stage_flush(const unsigned char* stage, int pitch, int lane, bf16* y, long ystride, int ncol, int rows) {
    const int cr = ncol >> 3;                                   // 16-byte chunks per row (<= 16)
    const unsigned magic = (65536u + cr - 1) / cr;              // c / cr for c < 2048
    const int total = rows * cr;
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < total; c += 64) {
        const int row = (int)(((unsigned)c * magic) >> 16), ch = c - row * cr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * pitch + ch * 16);
        *reinterpret_cast<u32x4*>(y + row * ystride + ch * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// fp32 outputs (fp32 plans) take the same route: a lane's 16 channels are 64 B, written as 16-byte pieces that sit
// cout * 4 bytes apart from the neighbour lane's -- the isolated-piece pattern again (measured: the fp32 1x1 convs
// wrote at 2.8 TB/s).  Tile rows are then ncol * 4 bytes.
FR_DEV void stage_put_f32(unsigned char* stage, int pitch, int row, int col, const Vec8f& v) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v.v[i]; b[i] = v.v[4 + i]; }
    *reinterpret_cast<f32x4*>(stage + row * pitch + col * 4) = a;
    *reinterpret_cast<f32x4*>(stage + row * pitch + col * 4 + 16) = b;
}
FR_DEV void stage_flush_f32(const unsigned char* stage, int pitch, int lane, float* y, long ystride, int ncol, int rows) {
    const int cr = ncol >> 2;                                   // 16-byte chunks per row (<= 32)
    const unsigned magic = (65536u + cr - 1) / cr;
    const int total = rows * cr;
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < total; c += 64) {
        const int row = (int)(((unsigned)c * magic) >> 16), ch = c - row * cr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * pitch + ch * 16);
        *reinterpret_cast<u32x4*>(y + row * ystride + ch * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// Common epilogue arguments of the GEMM-shaped conv kernels.
struct EpiArgs {
    const float* bias;      // [cout_pad32] folded bias (zeros when absent)
    const float* slope;     // [cout_pad32] PReLU slopes (or NULL)
    const void* residual;   // [M, cout] activation-typed, or NULL
    void* y;                // [M, cout] activation-typed
    int cout;
    int act;
    // f24 plans: the lo planes ([M][f24_lo_pitch(cout)] bytes) of y and of the residual
    unsigned char* y_lo;
    const unsigned char* residual_lo;
};

// Store one 32x32 accumulator tile.  Lane: pixel m (row of Y), channels c0 .. c0+15.
template <typename TOut>
FR_DEV void epilogue_store(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok) {
    if (!m_ok) return;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int c = c0 + 8 * g;
        if (c >= e.cout) continue;
        Vec8f v;
        f32x4 b0 = *reinterpret_cast<const f32x4*>(e.bias + c);
        f32x4 b1 = *reinterpret_cast<const f32x4*>(e.bias + c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v.v[i] = acc[8 * g + i] + b0[i]; v.v[4 + i] = acc[8 * g + 4 + i] + b1[i]; }
        if (e.act == FR_ACT_PRELU) {
            prelu8(v, *reinterpret_cast<const f32x4*>(e.slope + c), *reinterpret_cast<const f32x4*>(e.slope + c + 4));
        } else if (e.act != FR_ACT_NONE) {
            act8(v, e.act, nullptr);
        }
        long off = m * (long)e.cout + c;
        if constexpr (kIsF24<TOut>) {
            const long off_lo = m * (long)f24_lo_pitch(e.cout) + c;
            if (e.residual) {
                Vec8f r = f24_load8(reinterpret_cast<const bf16*>(e.residual) + off, e.residual_lo + off_lo);
#pragma unroll
                for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
            }
            f24_store8(reinterpret_cast<bf16*>(e.y) + off, e.y_lo + off_lo, v);
        } else {
            if (e.residual) {
                Vec8f r = load8(reinterpret_cast<const TOut*>(e.residual) + off);
#pragma unroll
                for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
            }
            store8(reinterpret_cast<TOut*>(e.y) + off, v);
        }
    }
}

// Same arithmetic as epilogue_store, but the bf16 result goes to the wave's LDS staging tile
// (row = pixel within the tile, col = channel within the staged group) instead of global memory.
FR_DEV void epilogue_stage(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok,
                           unsigned char* stage, int pitch, int row, int col0) {
    if (!m_ok) return;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int c = c0 + 8 * g;
        if (c >= e.cout) continue;
        Vec8f v;
        f32x4 b0 = *reinterpret_cast<const f32x4*>(e.bias + c);
        f32x4 b1 = *reinterpret_cast<const f32x4*>(e.bias + c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v.v[i] = acc[8 * g + i] + b0[i]; v.v[4 + i] = acc[8 * g + 4 + i] + b1[i]; }
        if (e.act == FR_ACT_PRELU) {
            prelu8(v, *reinterpret_cast<const f32x4*>(e.slope + c), *reinterpret_cast<const f32x4*>(e.slope + c + 4));
        } else if (e.act != FR_ACT_NONE) {
            act8(v, e.act, nullptr);
        }
        if (e.residual) {
            Vec8f r = load8(reinterpret_cast<const bf16*>(e.residual) + m * (long)e.cout + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
        }
        stage_put(stage, pitch, row, col0 + 8 * g, v);
    }
}

// epilogue_stage for fp32 outputs: same arithmetic as epilogue_store<float>, result into the wave's fp32 staging tile.
FR_DEV void epilogue_stage_f32(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok,
                               unsigned char* stage, int pitch, int row, int col0) {
    if (!m_ok) return;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int c = c0 + 8 * g;
        if (c >= e.cout) continue;
        Vec8f v;
        f32x4 b0 = *reinterpret_cast<const f32x4*>(e.bias + c);
        f32x4 b1 = *reinterpret_cast<const f32x4*>(e.bias + c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v.v[i] = acc[8 * g + i] + b0[i]; v.v[4 + i] = acc[8 * g + 4 + i] + b1[i]; }
        if (e.act == FR_ACT_PRELU) {
            prelu8(v, *reinterpret_cast<const f32x4*>(e.slope + c), *reinterpret_cast<const f32x4*>(e.slope + c + 4));
        } else if (e.act != FR_ACT_NONE) {
            act8(v, e.act, nullptr);
        }
        if (e.residual) {
            Vec8f r = load8(reinterpret_cast<const float*>(e.residual) + m * (long)e.cout + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
        }
        stage_put_f32(stage, pitch, row, col0 + 8 * g, v);
    }
}

// epilogue_stage for f24 outputs: same arithmetic, the result rounded to f24; the hi halves go to the wave's bf16-shaped
// staging tile (written back by stage_flush like a bf16 tile), the lo bytes of the lane's 16 channels (16 B) to a second
// tile at byte column col0 (written back by stage_flush_lo).  A lane whose pixel is out of range writes nothing.
FR_DEV void epilogue_stage_f24(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok,
                               unsigned char* stage, int pitch, unsigned char* stage_lo, int pitch_lo, int row, int col0) {
    if (!m_ok) return;
    u32x4 lo = {0u, 0u, 0u, 0u};
    const int LP = f24_lo_pitch(e.cout);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int c = c0 + 8 * g;
        if (c >= e.cout) continue;
        Vec8f v;
        f32x4 b0 = *reinterpret_cast<const f32x4*>(e.bias + c);
        f32x4 b1 = *reinterpret_cast<const f32x4*>(e.bias + c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v.v[i] = acc[8 * g + i] + b0[i]; v.v[4 + i] = acc[8 * g + 4 + i] + b1[i]; }
        if (e.act == FR_ACT_PRELU) {
            prelu8(v, *reinterpret_cast<const f32x4*>(e.slope + c), *reinterpret_cast<const f32x4*>(e.slope + c + 4));
        } else if (e.act != FR_ACT_NONE) {
            act8(v, e.act, nullptr);
        }
        if (e.residual) {
            Vec8f r = f24_load8(reinterpret_cast<const bf16*>(e.residual) + m * (long)e.cout + c, e.residual_lo + m * (long)LP + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
        }
        u32x4 h; unsigned l0, l1;
        f24_encode(v, h, l0, l1);
        *reinterpret_cast<u32x4*>(stage + row * pitch + (col0 + 8 * g) * 2) = h;
        lo[2 * g] = l0; lo[2 * g + 1] = l1;
    }
    *reinterpret_cast<u32x4*>(stage_lo + row * pitch_lo + col0) = lo;
}
// The arithmetic of every GEMM-shaped epilogue: v[g] = act(acc + bias) (+ residual) for the lane's channel groups
// c0 + 8 g .. + 7 (g = 0, 1) of pixel m; ok[g] says whether the group exists (pixel inside M, channels inside cout).
// Every load is issued up front and unconditionally, at an address clamped into the arrays (row 0, channel 0 for a
// lane that is out of range; bias and slopes are padded to 32 channels): loads behind per-lane bounds branches cannot
// be hoisted by the compiler and cost one dependent round trip each (bias, slope, residual, twice per tile).
// T: storage type of the residual (bf16, float, f24).
template <typename T>
FR_DEV void epilogue_values(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok, Vec8f (&v)[2], bool (&ok)[2]) {
    const long mm = m_ok ? m : 0;
    int cs[2];
    f32x4 b[2][2], sl[2][2];
    Vec8f r[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int c = c0 + 8 * g;
        ok[g] = m_ok && c < e.cout;
        cs[g] = c < e.cout ? c : 0;
        b[g][0] = *reinterpret_cast<const f32x4*>(e.bias + cs[g]);
        b[g][1] = *reinterpret_cast<const f32x4*>(e.bias + cs[g] + 4);
    }
    const bool prelu = e.act == FR_ACT_PRELU;
    if (prelu) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            sl[g][0] = *reinterpret_cast<const f32x4*>(e.slope + cs[g]);
            sl[g][1] = *reinterpret_cast<const f32x4*>(e.slope + cs[g] + 4);
        }
    }
    if (e.residual) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const long off = mm * (long)e.cout + cs[g];
            if constexpr (kIsF24<T>) r[g] = f24_load8(reinterpret_cast<const bf16*>(e.residual) + off, e.residual_lo + mm * (long)f24_lo_pitch(e.cout) + cs[g]);
            else r[g] = load8(reinterpret_cast<const T*>(e.residual) + off);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[g].v[i] = acc[8 * g + i] + b[g][0][i]; v[g].v[4 + i] = acc[8 * g + 4 + i] + b[g][1][i]; }
        if (prelu) prelu8(v[g], sl[g][0], sl[g][1]);
        else if (e.act != FR_ACT_NONE) act8(v[g], e.act, nullptr);
        if (e.residual) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[g].v[i] += r[g].v[i];
        }
    }
}

// ---- the *_h forms: the same functions on epilogue_values (all loads up front).  Measured per kernel: they pay in
// pw_proj_kernel and dwp_kernel (-1..2 %), cost occupancy in stem_kernel / pw_gemm_kernel (68 -> 106 VGPRs), so each
// call site chooses.
template <typename TOut>
FR_DEV void epilogue_store_h(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok) {
    Vec8f v[2];
    bool ok[2];
    epilogue_values<TOut>(acc, e, m, c0, m_ok, v, ok);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!ok[g]) continue;
        const int c = c0 + 8 * g;
        const long off = m * (long)e.cout + c;
        if constexpr (kIsF24<TOut>) f24_store8(reinterpret_cast<bf16*>(e.y) + off, e.y_lo + m * (long)f24_lo_pitch(e.cout) + c, v[g]);
        else store8(reinterpret_cast<TOut*>(e.y) + off, v[g]);
    }
}

// Same arithmetic as epilogue_store, but the bf16 result goes to the wave's LDS staging tile
// (row = pixel within the tile, col = channel within the staged group) instead of global memory.
FR_DEV void epilogue_stage_h(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok,
                           unsigned char* stage, int pitch, int row, int col0) {
    Vec8f v[2];
    bool ok[2];
    epilogue_values<bf16>(acc, e, m, c0, m_ok, v, ok);
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (ok[g]) stage_put(stage, pitch, row, col0 + 8 * g, v[g]);
}

// epilogue_stage for fp32 outputs: same arithmetic as epilogue_store_h<float>, result into the wave's fp32 staging tile.
FR_DEV void epilogue_stage_f32_h(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok,
                               unsigned char* stage, int pitch, int row, int col0) {
    Vec8f v[2];
    bool ok[2];
    epilogue_values<float>(acc, e, m, c0, m_ok, v, ok);
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (ok[g]) stage_put_f32(stage, pitch, row, col0 + 8 * g, v[g]);
}

// epilogue_stage for f24 outputs: same arithmetic, the result rounded to f24; the hi halves go to the wave's bf16-shaped
// staging tile (written back by stage_flush like a bf16 tile), the lo bytes of the lane's 16 channels (16 B) to a second
// tile at byte column col0 (written back by stage_flush_lo).  A lane whose pixel is out of range writes nothing.
FR_DEV void epilogue_stage_f24_h(const f32x16& acc, const EpiArgs& e, long m, int c0, bool m_ok,
                               unsigned char* stage, int pitch, unsigned char* stage_lo, int pitch_lo, int row, int col0) {
    Vec8f v[2];
    bool ok[2];
    epilogue_values<f24>(acc, e, m, c0, m_ok, v, ok);
    if (!m_ok) return;
    u32x4 lo = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!ok[g]) continue;
        u32x4 h; unsigned l0, l1;
        f24_encode(v[g], h, l0, l1);
        *reinterpret_cast<u32x4*>(stage + row * pitch + (col0 + 8 * g) * 2) = h;
        lo[2 * g] = l0; lo[2 * g + 1] = l1;
    }
    *reinterpret_cast<u32x4*>(stage_lo + row * pitch_lo + col0) = lo;
}
// rows [0,rows) x ncol channels of the lo staging tile -> the lo plane (row stride lp bytes), whole 16-byte chunks (a
// trailing half chunk lands in the padding of the 16-byte-padded lo rows)
__host__ __device__ constexpr int stage_pitch_lo(int row_bytes) { return row_bytes + 16; }
FR_DEV void stage_flush_lo(const unsigned char* stage_lo, int pitch_lo, int lane, unsigned char* y_lo, long lp, int ncol, int rows) {
    const int cr = (ncol + 15) >> 4;                            // 16-byte chunks per row (<= 8)
    const unsigned magic = (65536u + cr - 1) / cr;
    const int total = rows * cr;
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < total; c += 64) {
        const int row = (int)(((unsigned)c * magic) >> 16), ch = c - row * cr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage_lo + row * pitch_lo + ch * 16);
        *reinterpret_cast<u32x4*>(y_lo + row * lp + ch * 16) = v;
    }
    __builtin_amdgcn_wave_barrier();
}
