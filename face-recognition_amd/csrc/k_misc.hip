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
    const T* X = reinterpret_cast<const T*>(a.x) + (long)n * a.P * a.C + (cv0 + cv) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ps < PS) {
        // four loads in flight per thread; the summation order (p ascending) is unchanged
        for (int p0 = ps; p0 < a.P; p0 += 4 * PS) {
            Vec8f v[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = p0 + q * PS;
                const int pc = p < a.P ? p : ps;
                v[q] = load8(X + (long)pc * a.C);
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

int launch_reduce(bool f32, const ReduceArgs& a, hipStream_t st) {
    if (a.C % 8 || a.N <= 0 || a.P <= 0) FR_FAIL(FR_ERR_ARG, "reduce: C must be a multiple of 8");
    const int nchunk = cdiv(a.C / 8, 32);
    dim3 grid((unsigned)((long)a.N * nchunk));
    if (f32) hipLaunchKernelGGL((reduce_kernel<float>), grid, dim3(256), 0, st, a);
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
int launch_nchw_to_nhwc(const float* x, void* y, bool y_f32, int N, int C, int H, int W, hipStream_t st) {
    const long total = (long)N * C * H * W;
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 65536);
    if (y_f32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid), dim3(256), 0, st, x, (float*)y, total, C, H * W);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16>), dim3(grid), dim3(256), 0, st, x, (bf16*)y, total, C, H * W);
    FR_LAUNCH_CHECK("nchw_to_nhwc_kernel");
    return FR_OK;
}
int launch_nhwc_to_nchw(const void* x, bool x_f32, float* y, int N, int C, int H, int W, hipStream_t st) {
    const long total = (long)N * C * H * W;
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 65536);
    if (x_f32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, y, total, C, H * W);
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
