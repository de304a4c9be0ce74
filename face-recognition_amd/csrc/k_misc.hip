// Wavefront-reduction and layout kernels: weighted spatial reduction (SE average pool, GDC 7x7
// global depthwise conv), NCHW<->NHWC conversion, L2 normalisation, pairwise cosine + decision.
#include "kernels.h"

// out[n][c] = sum_p x[n][p][c] * w[p][c] + b[c]   (w == NULL: mean over p).
// Replaces AdaptiveAvgPool2d(1) (utils/layers.py:107,114) and the GDC depthwise 7x7 valid conv +
// BatchNorm2d (utils/layers.py:180, 156-173).  One block per (image, 32-channel-vector chunk);
// threads = (channel vector) x (pixel slice); partial sums meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void reduce_kernel(ReduceArgs a) {
    __shared__ float part[256 * 8];
    const int CV = a.C >> 3;
    const int nchunk = (CV + 31) / 32;
    const int n = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int cv0 = ch * 32, ncv = min(32, CV - cv0);
    const int PS = 256 / ncv;
    const int cv = threadIdx.x % ncv, ps = threadIdx.x / ncv;
    typedef typename act_hi<T>::type TH;
    const TH* X = reinterpret_cast<const TH*>(a.x) + (long)n * a.P * a.C + (cv0 + cv) * 8;
    const int LP = f24_lo_pitch(a.C);
    const unsigned char* XL = kIsF24<T> ? a.x_lo + (long)n * a.P * LP + (cv0 + cv) * 8 : nullptr;
    (void)XL;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ps < PS) {
        // four loads in flight per thread; the summation order (p ascending) is unchanged
        for (int p0 = ps; p0 < a.P; p0 += 4 * PS) {
            Vec8f v[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = p0 + q * PS;
                const int pc = p < a.P ? p : ps;
                if constexpr (kIsF24<T>) v[q] = f24_load8(X + (long)pc * a.C, XL + (long)pc * LP);
                else v[q] = load8(X + (long)pc * a.C);
                if (a.w) w[q] = load8(a.w + (long)pc * a.C + (cv0 + cv) * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (p0 + q * PS < a.P) {
                    if (a.w) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[j] = fmaf(v[q].v[j], w[q].v[j], s[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[j] += v[q].v[j];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[threadIdx.x * 8 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < ncv * 8) {
        const int c = threadIdx.x >> 3, j = threadIdx.x & 7;     // channel vector c, lane j
        float t = 0.f;
        for (int q = 0; q < PS; ++q) t += part[(q * ncv + c) * 8 + j];   // fixed order: deterministic
        const int cg = (cv0 + c) * 8 + j;
        if (!a.w) t *= a.scale != 0.f ? a.scale : 1.f / (float)a.P;
        if (a.bias) t += a.bias[cg];
        a.out[(long)n * a.C + cg] = t;
    }
}

int launch_reduce(int fmt, const ReduceArgs& a, hipStream_t st) {
    if (a.C % 8 || a.N <= 0 || a.P <= 0) FR_FAIL(FR_ERR_ARG, "reduce: C must be a multiple of 8");
    if (fmt == FMT_F24 && !a.x_lo) FR_FAIL(FR_ERR_ARG, "reduce: f24 input without its lo plane");
    const int nchunk = cdiv(a.C / 8, 32);
    dim3 grid((unsigned)((long)a.N * nchunk));
    if (fmt == FMT_F24) hipLaunchKernelGGL((reduce_kernel<f24>), grid, dim3(256), 0, st, a);
    else if (fmt == FMT_F32) hipLaunchKernelGGL((reduce_kernel<float>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((reduce_kernel<bf16>), grid, dim3(256), 0, st, a);
    FR_LAUNCH_CHECK("reduce_kernel");
    return FR_OK;
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, long total, int C, int HW) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C; const long r = i / C;
        const long p = r % HW, n = r / HW;
        y[i] = (T)x[(n * C + c) * HW + p];
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, long total, int C, int HW) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i % HW; const long r = i / HW;
        const int c = r % C; const long n = r / C;
        y[i] = (float)x[(n * HW + p) * C + c];
    }
}
// f24 forms: one thread per 8 channels of a pixel (C % 8 == 0)
__global__ void nchw_to_nhwc_f24_kernel(const float* x, bf16* y, unsigned char* y_lo, long total8, int C, int HW) {
    const int CV = C >> 3, LP = f24_lo_pitch(C);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const int cv = i % CV; const long m = i / CV;
        const long p = m % HW, n = m / HW;
        Vec8f v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = x[(n * C + cv * 8 + j) * HW + p];
        f24_store8(y + m * C + cv * 8, y_lo + m * LP + cv * 8, v);
    }
}
__global__ void nhwc_to_nchw_f24_kernel(const bf16* x, const unsigned char* x_lo, float* y, long total8, int C, int HW) {
    const int CV = C >> 3, LP = f24_lo_pitch(C);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const int cv = i % CV; const long m = i / CV;
        const long p = m % HW, n = m / HW;
        const Vec8f v = f24_load8(x + m * C + cv * 8, x_lo + m * LP + cv * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[(n * C + cv * 8 + j) * HW + p] = v.v[j];
    }
}
int launch_nchw_to_nhwc(const float* x, void* y, void* y_lo, int fmt, int N, int C, int H, int W, hipStream_t st) {
    const long total = (long)N * C * H * W;
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 65536);
    if (fmt == FMT_F24) {
        if (C % 8 || !y_lo) FR_FAIL(FR_ERR_UNSUPPORTED, "nchw_to_nhwc (f24): C must be a multiple of 8");
        hipLaunchKernelGGL(nchw_to_nhwc_f24_kernel, dim3(grid), dim3(256), 0, st, x, (bf16*)y, (unsigned char*)y_lo, total / 8, C, H * W);
    } else if (fmt == FMT_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid), dim3(256), 0, st, x, (float*)y, total, C, H * W);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16>), dim3(grid), dim3(256), 0, st, x, (bf16*)y, total, C, H * W);
    FR_LAUNCH_CHECK("nchw_to_nhwc_kernel");
    return FR_OK;
}
int launch_nhwc_to_nchw(const void* x, const void* x_lo, int fmt, float* y, int N, int C, int H, int W, hipStream_t st) {
    const long total = (long)N * C * H * W;
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 65536);
    if (fmt == FMT_F24) {
        if (C % 8 || !x_lo) FR_FAIL(FR_ERR_UNSUPPORTED, "nhwc_to_nchw (f24): C must be a multiple of 8");
        hipLaunchKernelGGL(nhwc_to_nchw_f24_kernel, dim3(grid), dim3(256), 0, st, (const bf16*)x, (const unsigned char*)x_lo, y, total / 8, C, H * W);
    } else if (fmt == FMT_F32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, y, total, C, H * W);
    else hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16>), dim3(grid), dim3(256), 0, st, (const bf16*)x, y, total, C, H * W);
    FR_LAUNCH_CHECK("nhwc_to_nchw_kernel");
    return FR_OK;
}

// ---- one wave per row reductions -------------------------------------------------------------
FR_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// y = x / ||x||  (rows with ||x|| == 0 stay 0).  One wave per row.
template <bool BF16OUT>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* x, void* y, float* norms, int n, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* xr = x + (long)row * d;
    float ss = 0.f;
    for (int i = lane; i < d; i += 64) ss = fmaf(xr[i], xr[i], ss);
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
    for (int i = lane; i < d; i += 64) {
        const float v = xr[i] * inv;
        if constexpr (BF16OUT) reinterpret_cast<bf16*>(y)[(long)row * d + i] = (bf16)v;
        else reinterpret_cast<float*>(y)[(long)row * d + i] = v;
    }
    if (norms && lane == 0) norms[row] = nrm;
}

// sim = a.b / (||a|| * ||b|| + eps), dec = (double)sim > thr.  One wave per pair.
// eps = 0: utils/face_utils.py:86; eps = 1e-5: evaluate.py:554; decision: evaluate.py:205-206.
__global__ __launch_bounds__(256) void cosine_pairs_kernel(const float* a, const float* b, int pairs, int d,
                                                           float eps, double thr, float* sim, uint8_t* dec) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= pairs) return;
    const float* ar = a + (long)row * d;
    const float* br = b + (long)row * d;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float u = ar[i], v = br[i];
        ab = fmaf(u, v, ab); aa = fmaf(u, u, aa); bb = fmaf(v, v, bb);
    }
    ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
    if (lane == 0) {
        const float s = ab / (sqrtf(aa) * sqrtf(bb) + eps);
        sim[row] = s;
        if (dec) dec[row] = (double)s > thr ? 1 : 0;
    }
}

extern "C" int fr_l2norm(const float* x, float* y, float* norms, int n, int d, void* stream) {
    if (n < 0 || d <= 0) FR_FAIL(FR_ERR_ARG, "fr_l2norm: bad shape");
    if (n == 0) return FR_OK;
    hipLaunchKernelGGL((l2norm_kernel<false>), dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, (void*)y, norms, n, d);
    FR_LAUNCH_CHECK("l2norm_kernel");
    return FR_OK;
}
extern "C" int fr_l2norm_bf16(const float* x, uint16_t* y, int n, int d, void* stream) {
    if (n < 0 || d <= 0) FR_FAIL(FR_ERR_ARG, "fr_l2norm_bf16: bad shape");
    if (n == 0) return FR_OK;
    hipLaunchKernelGGL((l2norm_kernel<true>), dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, (void*)y, (float*)nullptr, n, d);
    FR_LAUNCH_CHECK("l2norm_kernel");
    return FR_OK;
}
extern "C" int fr_cosine_pairs(const float* a, const float* b, int pairs, int d, float eps, double threshold,
                               float* out_sim, uint8_t* out_dec, void* stream) {
    if (pairs < 0 || d <= 0) FR_FAIL(FR_ERR_ARG, "fr_cosine_pairs: bad shape");
    if (pairs == 0) return FR_OK;
    hipLaunchKernelGGL(cosine_pairs_kernel, dim3(cdiv(pairs, 4)), dim3(256), 0, (hipStream_t)stream,
                       a, b, pairs, d, eps, threshold, out_sim, out_dec);
    FR_LAUNCH_CHECK("cosine_pairs_kernel");
    return FR_OK;
}
