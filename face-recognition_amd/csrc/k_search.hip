// 1:N cosine search: scores = G . Q^T on bf16 MFMA with exact per-lane top-k selection, plus the
// candidate merge used for gallery splits / shards, plus the synthetic row generator.
//
// The reference has no 1:N search; the closest arithmetic is the eval branch
// of MarginCosineProduct (utils/metrics.py:18-28: normalised embeddings x normalised matrix^T) and
// calculate_accuracy's topk (utils/general.py:102-116).  Oracle: oracle/search.py.
//
// Kernel: a block owns 128 queries, kept in LDS for its whole life (128 x 512 bf16, rows padded
// by 16 B so ds_read_b128 is bank-conflict free), and streams its gallery split in 128-row tiles,
// 32 k-columns at a time, double-buffered through registers.  4 waves in a 2x2 arrangement; each
// wave computes a 64 (gallery) x 64 (query) tile as 2x2 v_mfma_f32_32x32x16_bf16 tiles with the
// gallery as the "A" operand, so that accumulator columns are queries: every lane owns ONE query
// per n-tile and sees 32 gallery scores for it, and keeps a private sorted top-k list in
// registers (strict '>' insert => ties stay with the lower index).  Lists of the lanes / waves /
// gallery splits that share a query are merged by topk_select_kernel.
#include "kernels.h"

#define SEARCH_D 512
#define SEARCH_KMAX 8
#define Q_STRIDE (SEARCH_D * 2 + 16)     // bytes per query row in LDS
#define G_STRIDE (32 * 2 + 16)           // bytes per gallery row slice in LDS
#define SEARCH_LISTS 4                   // lists per query per split: 2 wave rows x 2 lane halves

struct SearchArgs {
    const bf16* q; const bf16* g;
    long Q, G;
    int k, nsplit;
    long tiles_per_split;                // 128-row gallery tiles per split
    float* part_s; int* part_i;          // [nsplit][SEARCH_LISTS][Qpad][k]
    long Qpad;
};

template <int KK>
__global__ __launch_bounds__(256) void search_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qs = smem;                                  // 128 * Q_STRIDE
    unsigned char* Gs = smem + 128 * Q_STRIDE;                 // 2 * 128 * G_STRIDE
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wg = wave >> 1, wq = wave & 1;                   // wave row (gallery), wave col (query)
    const long q0 = (long)blockIdx.x * 128;
    const int split = blockIdx.y;

    // stage the query tile (rows beyond Q are clamped; their results are never read)
    for (int i = tid; i < 128 * 64; i += 256) {
        const int row = i >> 6, c = i & 63;
        const long qr = min(q0 + row, a.Q - 1);
        *reinterpret_cast<u32x4*>(Qs + row * Q_STRIDE + c * 16) =
            *reinterpret_cast<const u32x4*>(a.q + qr * SEARCH_D + c * 8);
    }

    float ts[2][KK]; int ti[2][KK];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < KK; ++e) { ts[j][e] = -INFINITY; ti[j][e] = 0x7FFFFFFF; }

    const long tile_begin = (long)split * a.tiles_per_split;
    const long ntiles_all = (a.G + 127) / 128;
    const long tile_end = min(tile_begin + a.tiles_per_split, ntiles_all);

    // staging role: thread -> (row = tid>>1, 32-byte half = tid&1) of a 128 x 32 slice (2 x 16 B)
    const int srow = tid >> 1, shalf = tid & 1;
    u32x4 pre0, pre1;
    auto prefetch = [&](long tile, int ks) {
        const long gr = min(tile * 128 + srow, a.G - 1);
        const bf16* src = a.g + gr * SEARCH_D + ks * 32 + shalf * 16;
        pre0 = *reinterpret_cast<const u32x4*>(src);
        pre1 = *reinterpret_cast<const u32x4*>(src + 8);
    };
    auto commit = [&](int buf) {
        unsigned char* dst = Gs + buf * 128 * G_STRIDE + srow * G_STRIDE + shalf * 32;
        *reinterpret_cast<u32x4*>(dst) = pre0;
        *reinterpret_cast<u32x4*>(dst + 16) = pre1;
    };

    if (tile_begin < tile_end) { prefetch(tile_begin, 0); commit(0); }
    __syncthreads();

    for (long tile = tile_begin; tile < tile_end; ++tile) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll 1
        for (int ks = 0; ks < 16; ++ks) {
            const int buf = ks & 1;
            const bool more = ks < 15 || tile + 1 < tile_end;
            if (more) { if (ks < 15) prefetch(tile, ks + 1); else prefetch(tile + 1, 0); }
            const unsigned char* gb = Gs + buf * 128 * G_STRIDE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(gb + (wg * 64 + i * 32 + p) * G_STRIDE + (s * 16 + 8 * h) * 2);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8*>(Qs + (wq * 64 + j * 32 + p) * Q_STRIDE + (ks * 32 + s * 16 + 8 * h) * 2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            if (more) commit(buf ^ 1);
            __syncthreads();
        }

        // selection: lane owns query column (wq*64 + j*32 + p); rows = gallery
        const long gbase = tile * 128 + wg * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long grow = gbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float s = acc[i][j][r];
                    if (grow < a.G && s > ts[j][KK - 1]) {
                        int gi = (int)grow;
#pragma unroll
                        for (int e = 0; e < KK; ++e) {
                            if (s > ts[j][e]) {
                                const float t = ts[j][e]; ts[j][e] = s; s = t;
                                const int u = ti[j][e]; ti[j][e] = gi; gi = u;
                            }
                        }
                    }
                }
            }
        }
    }

    // publish this lane's lists
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long qrow = q0 + wq * 64 + j * 32 + p;
        const int list = wg * 2 + h;
        const long base = (((long)split * SEARCH_LISTS + list) * a.Qpad + qrow) * a.k;
        for (int e = 0; e < KK; ++e)
            if (e < a.k) { a.part_s[base + e] = ts[j][e]; a.part_i[base + e] = ti[j][e]; }
    }
}

// out[q] = best k of nlists candidate lists (each k long, list stride `lstride` elements),
// ordered by (score desc, index asc).  One thread per query.
template <typename TI>
__global__ void topk_select_kernel(const float* s, const TI* idx, int nlists, long lstride, long Q, int k,
                                   long idx_offset, float* out_s, int64_t* out_i) {
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float bs[SEARCH_KMAX]; int64_t bi[SEARCH_KMAX];
#pragma unroll
    for (int e = 0; e < SEARCH_KMAX; ++e) { bs[e] = -INFINITY; bi[e] = INT64_MAX; }
    for (int l = 0; l < nlists; ++l) {
        for (int c = 0; c < k; ++c) {
            float v = s[l * lstride + q * k + c];
            int64_t vi = (int64_t)idx[l * lstride + q * k + c];
            if (vi == (int64_t)0x7FFFFFFF && sizeof(TI) == 4) continue;        // empty slot
#pragma unroll
            for (int e = 0; e < SEARCH_KMAX; ++e) {
                if (e < k && (v > bs[e] || (v == bs[e] && vi < bi[e]))) {
                    const float t = bs[e]; bs[e] = v; v = t;
                    const int64_t u = bi[e]; bi[e] = vi; vi = u;
                }
            }
        }
    }
    for (int e = 0; e < k; ++e) {
        out_s[q * k + e] = bs[e];
        out_i[q * k + e] = bi[e] == INT64_MAX ? -1 : bi[e] + idx_offset;
    }
}

static void search_geometry(long Q, long G, int* nsplit, long* tiles_per_split, long* Qpad) {
    const long qtiles = (Q + 127) / 128, gtiles = (G + 127) / 128;
    long ns = (1024 + qtiles - 1) / qtiles;        // aim for >= 1024 blocks
    if (ns > gtiles) ns = gtiles;
    if (ns < 1) ns = 1;
    const long tps = (gtiles + ns - 1) / ns;
    ns = (gtiles + tps - 1) / tps;
    *nsplit = (int)ns; *tiles_per_split = tps; *Qpad = qtiles * 128;
}

extern "C" int fr_topk_workspace_bytes(int64_t Q, int64_t G, int k, size_t* out_bytes) {
    if (Q <= 0 || G <= 0 || k < 1 || k > SEARCH_KMAX || !out_bytes) FR_FAIL(FR_ERR_ARG, "fr_topk_workspace_bytes: bad arguments");
    int ns; long tps, qpad;
    search_geometry(Q, G, &ns, &tps, &qpad);
    *out_bytes = (size_t)ns * SEARCH_LISTS * qpad * k * 8 + 512;
    return FR_OK;
}

extern "C" int fr_topk_search(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int d, int k,
                              int64_t idx_offset, float* out_scores, int64_t* out_idx,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !g || !out_scores || !out_idx || Q <= 0 || G <= 0) FR_FAIL(FR_ERR_ARG, "fr_topk_search: bad arguments");
    if (d != SEARCH_D) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search: d must be 512");
    if (k < 1 || k > SEARCH_KMAX) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search: k must be in 1..8");
    if (G > 0x7FFFFF00L) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search: at most 2^31 gallery rows per call (shard it)");
    size_t need = 0;
    fr_topk_workspace_bytes(Q, G, k, &need);
    if (!workspace || workspace_bytes < need) FR_FAIL(FR_ERR_WORKSPACE, "fr_topk_search: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    SearchArgs a{};
    a.q = reinterpret_cast<const bf16*>(q); a.g = reinterpret_cast<const bf16*>(g);
    a.Q = Q; a.G = G; a.k = k;
    search_geometry(Q, G, &a.nsplit, &a.tiles_per_split, &a.Qpad);
    unsigned char* ws = reinterpret_cast<unsigned char*>(align_up((size_t)workspace, 256));
    const size_t nel = (size_t)a.nsplit * SEARCH_LISTS * a.Qpad * k;
    a.part_s = reinterpret_cast<float*>(ws);
    a.part_i = reinterpret_cast<int*>(ws + nel * 4);
    const size_t lds = 128 * Q_STRIDE + 2 * 128 * G_STRIDE;
    dim3 grid((unsigned)(a.Qpad / 128), (unsigned)a.nsplit);
    static bool attr_set = false;
    if (!attr_set) {
        FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (k == 1) hipLaunchKernelGGL((search_kernel<1>), grid, dim3(256), lds, st, a);
    else if (k <= 5) hipLaunchKernelGGL((search_kernel<5>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((search_kernel<8>), grid, dim3(256), lds, st, a);
    FR_LAUNCH_CHECK("search_kernel");
    hipLaunchKernelGGL((topk_select_kernel<int>), dim3((unsigned)((Q + 127) / 128)), dim3(128), 0, st,
                       a.part_s, a.part_i, a.nsplit * SEARCH_LISTS, a.Qpad * (long)k, (long)Q, k, (long)idx_offset,
                       out_scores, out_idx);
    FR_LAUNCH_CHECK("topk_select_kernel");
    return FR_OK;
}

extern "C" int fr_topk_merge(const float* scores, const int64_t* idx, int P, int64_t Q, int k,
                             float* out_scores, int64_t* out_idx, void* stream) {
    if (!scores || !idx || P < 1 || Q <= 0 || k < 1 || k > SEARCH_KMAX) FR_FAIL(FR_ERR_ARG, "fr_topk_merge: bad arguments");
    hipLaunchKernelGGL((topk_select_kernel<int64_t>), dim3((unsigned)((Q + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       scores, idx, P, (long)Q * k, (long)Q, k, 0L, out_scores, out_idx);
    FR_LAUNCH_CHECK("topk_select_kernel");
    return FR_OK;
}

// ---- synthetic rows (bit-identical to oracle/search.py) -----------------------------------------
__device__ __forceinline__ long synth_int(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const long s = (long)(z & 0xFFFF) + (long)((z >> 16) & 0xFFFF) + (long)((z >> 32) & 0xFFFF) + (long)(z >> 48);
    return s - 2 * 65535;
}

__global__ __launch_bounds__(256) void synth_rows_kernel(bf16* out, uint64_t seed, long row0, long nrows, int dim,
                                                         uint64_t plant_seed, long n_plant, long n_total) {
    const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= nrows) return;
    const long row = row0 + r;
    long plant_q = -1;
    if (n_plant > 0) {
        const long step = n_total / n_plant;
        const long d = ((row - 17) % n_total + n_total) % n_total;
        if (d % step == 0 && d / step < n_plant) plant_q = d / step;
    }
    long v[8];
    long ss = 0;
    const int per = dim / 64;                       // dim = 512 -> 8 per lane
    for (int e = 0; e < per; ++e) {
        const int col = e * 64 + lane;
        long x = synth_int(seed, (uint64_t)row * (uint64_t)dim + col);
        if (plant_q >= 0) x += 3 * synth_int(plant_seed, (uint64_t)plant_q * (uint64_t)dim + col);
        v[e] = x;
        ss += x * x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float scale = (float)(1.0 / sqrt((double)ss));
    for (int e = 0; e < per; ++e) out[r * dim + e * 64 + lane] = (bf16)((float)v[e] * scale);
}

extern "C" int fr_synth_rows(uint16_t* out, uint64_t seed, int64_t row0, int64_t nrows, int dim,
                             uint64_t plant_seed, int64_t n_plant, int64_t n_gallery_total, void* stream) {
    if (!out || nrows < 0 || dim != SEARCH_D || n_plant < 0 || (n_plant > 0 && n_gallery_total < n_plant))
        FR_FAIL(FR_ERR_ARG, "fr_synth_rows: bad arguments (dim must be 512)");
    if (nrows == 0) return FR_OK;
    const long blocks = (nrows + 3) / 4;
    if (blocks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "fr_synth_rows: too many rows per call");
    hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<bf16*>(out), seed, (long)row0, (long)nrows, dim, plant_seed, (long)n_plant, (long)n_gallery_total);
    FR_LAUNCH_CHECK("synth_rows_kernel");
    return FR_OK;
}
