// Batched face alignment warp: the reference's face_alignment
// (utils/face_utils.py:56-71) is cv2.warpAffine(image, M, (112, 112), borderValue=0.0) on the CPU, one face
// at a time (models/onnx_model.py:82).  This kernel restates OpenCV's INTER_LINEAR / BORDER_CONSTANT path
// (opencv-python 4.10, imgwarp.cpp) in the same integer arithmetic, one thread per output pixel:
//   X0(y) = round((m1*y + m2) * 1024) + 16,  adelta(x) = round(m0 * x * 1024)          (double, round-half-even)
//   X = (X0 + adelta) >> 5;  sx = X >> 5, fx = X & 31   (likewise Y with m3, m4, m5)
//   out = (sum of the 2x2 source pixels * {(32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx*fy} * 32 + 2^14) >> 15
// with 0 for source pixels outside the image.  m = the INVERTED 2x3 matrix (inverted on the host in
// double, as cv::warpAffine does before it starts).  The double products and sums are written with
// __dmul_rn / __dadd_rn so that the compiler cannot contract them into fused multiply-adds.
// Bit-exact against oracle/align.py; no reference run could pin either (cv2 is not in this image).
#include "kernels.h"

__global__ __launch_bounds__(256) void align_kernel(const unsigned char* const* images, const int* sizes_hw,
                                                    const double* inv, int n, int S, unsigned char* out) {
    const int f = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (f >= n || idx >= S * S) return;
    const int y = idx / S, x = idx - y * S;
    const double* m = inv + (size_t)f * 6;
    const int H = sizes_hw[2 * f], W = sizes_hw[2 * f + 1];
    const unsigned char* img = images[f];
    auto sat = [](double v) -> long {
        v = rint(v);
        return (long)fmin(fmax(v, -2147483648.0), 2147483647.0);
    };
    const long adelta = sat(__dmul_rn(__dmul_rn(m[0], (double)x), 1024.0));
    const long bdelta = sat(__dmul_rn(__dmul_rn(m[3], (double)x), 1024.0));
    const long X0 = sat(__dmul_rn(__dadd_rn(__dmul_rn(m[1], (double)y), m[2]), 1024.0)) + 16;
    const long Y0 = sat(__dmul_rn(__dadd_rn(__dmul_rn(m[4], (double)y), m[5]), 1024.0)) + 16;
    const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    long sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int acc[3] = {0, 0, 0};
    auto tap = [&](long yy, long xx, int w) {
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const unsigned char* p = img + ((size_t)yy * W + xx) * 3;
            acc[0] += p[0] * w; acc[1] += p[1] * w; acc[2] += p[2] * w;
        }
    };
    tap(sy, sx, w00); tap(sy, sx + 1, w01); tap(sy + 1, sx, w10); tap(sy + 1, sx + 1, w11);
    unsigned char* o = out + ((size_t)f * S * S + idx) * 3;
    o[0] = (unsigned char)((acc[0] + (1 << 14)) >> 15);
    o[1] = (unsigned char)((acc[1] + (1 << 14)) >> 15);
    o[2] = (unsigned char)((acc[2] + (1 << 14)) >> 15);
}

extern "C" int fr_align_faces(const unsigned char* const* images, const int* sizes_hw, const double* inv_matrices,
                              int n, int out_size, unsigned char* out, void* stream) {
    if (n < 0 || out_size <= 0 || out_size > 4096 || (n > 0 && (!images || !sizes_hw || !inv_matrices || !out)))
        FR_FAIL(FR_ERR_ARG, "fr_align_faces: bad arguments");
    if (n == 0) return FR_OK;
    if (n > 65535) FR_FAIL(FR_ERR_ARG, "fr_align_faces: at most 65535 faces per call");
    dim3 grid((unsigned)cdiv(out_size * out_size, 256), (unsigned)n);
    hipLaunchKernelGGL(align_kernel, grid, dim3(256), 0, (hipStream_t)stream, images, sizes_hw, inv_matrices, n, out_size, out);
    FR_LAUNCH_CHECK("align_kernel");
    return FR_OK;
}
