// 1:N cosine search: scores = G . Q^T on bf16 MFMA with exact per-lane top-k selection, plus the
// candidate merge used for gallery splits / shards, plus the synthetic row generator.
//
// The reference has no 1:N search; the closest arithmetic is the eval branch
// of MarginCosineProduct (utils/metrics.py:18-28: normalised embeddings x normalised matrix^T) and
// calculate_accuracy's topk (utils/general.py:102-116).  Oracle: oracle/search.py.
//
// Kernel: 4 waves in a 2x2 arrangement; each wave computes a 64 (gallery) x 64 (query) tile as 2x2
// v_mfma_f32_32x32x16_bf16 tiles with the gallery as the "A" operand, so that accumulator columns
// are queries: every lane owns ONE query per n-tile, sees 32 gallery scores for it per tile, and
// keeps a private sorted top-k list in registers (strict '>' insert => ties stay with the lower
// index).  Lists of the lanes / waves / gallery splits that share a query are merged by
// topk_select_kernel.
#include "kernels.h"

#define SEARCH_D 512
#define SEARCH_KMAX 8
#define SEARCH_LISTS 4                   // lists per query per split: 2 wave rows x 2 lane halves
#define S_BK 64                          // K slice (bf16 elements) staged per barrier
#define S_ROWB (S_BK * 2)
#define S_TILE_BYTES (128 * S_ROWB)      // one operand tile: 128 rows x 128 B
#define S_BUF (2 * S_TILE_BYTES)
#define S_KSL (SEARCH_D / S_BK)          // slices per gallery tile
#define S_PAD 1024                       // unused LDS bytes in front of the tiles (see lbase)

struct SearchArgs {
    const bf16* q; const bf16* g;
    long Q, G;
    int k, nsplit;
    long tiles_per_split;                // 128-row gallery tiles per split
    float* part_s; int* part_i;          // [nsplit][SEARCH_LISTS][Qpad][k]
    long Qpad;
    int walk;                            // resident-query kernels: 1 = strided tile walk (rq_tile_stride), 0 = in index order
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// A block owns 128 queries and one gallery split; it walks the split in 128-row tiles.  Both operand
// tiles of every 64-wide K slice go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane; the
// query slices are re-read from L2, which is what keeps LDS at 64 KB and two blocks per CU), with the
// bank swizzle (chunk ^ ((row>>1)&7)) applied on the source address and undone on the fragment read.
// The (tile, slice) sequence is flattened so the DMA of the next tile's first slice overlaps the last
// MFMAs and the top-k selection of the current tile.
// VALU diet (the first version of this loop issued 20 VALU instructions per MFMA and was VALU-bound):
// the 8 slices of a tile are unrolled so LDS buffer offsets and the per-slice global offset are
// immediates; row pointers are recomputed once per tile; the top-k selection first reduces each
// 32x32 accumulator tile to a per-lane maximum (v_max3) and skips the insertion code unless some
// lane of the wave beats its current k-th score.
template <int KK>
__global__ __launch_bounds__(256, 2) void search_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wg = wave >> 1, wq = wave & 1;                   // wave row (gallery), wave col (query)
    const long q0 = (long)blockIdx.x * 128;
    const int split = blockIdx.y;
    const long tile_begin = (long)split * a.tiles_per_split;
    const long ntiles_all = (a.G + 127) / 128;
    const long tile_end = min(tile_begin + a.tiles_per_split, ntiles_all);

    float ts[2][KK]; int ti[2][KK];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < KK; ++e) { ts[j][e] = -INFINITY; ti[j][e] = 0x7FFFFFFF; }

    // staging role: one wave instruction = 8 rows x 128 B; this lane: row sub, physical chunk pc
    const int sub = lane >> 3, pc = lane & 7;
    const unsigned char* qsrc[4];
    int grow[4], lcs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wave + 4 * j) + sub;
        const int lc = pc ^ ((row >> 1) & 7);
        lcs[j] = lc * 16;
        grow[j] = row;
        const long qr = min(q0 + row, a.Q - 1);
        qsrc[j] = reinterpret_cast<const unsigned char*>(a.q + qr * SEARCH_D) + lc * 16;
    }
    const unsigned char* gbase = reinterpret_cast<const unsigned char*>(a.g);
    const unsigned char* gsrc[4];
    auto tile_ptrs = [&](long tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long gr = min(tile * 128 + grow[j], a.G - 1);
            gsrc[j] = gbase + gr * (SEARCH_D * 2) + lcs[j];
        }
    };
    // The instruction offset of global_load_lds is added to BOTH the global and the LDS address, so the
    // LDS base passed for slice KSI is pre-decremented by KSI*S_ROWB; the tiles start S_PAD bytes into
    // the allocation so that this never goes below zero.
    unsigned char* const lbase = smem + S_PAD + wave * 1024;

    // per-lane fragment offsets for the 4 k-steps of a slice
    int aoff[2][4], boff[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wg * 64 + i * 32 + p;
#pragma unroll
        for (int s = 0; s < 4; ++s) aoff[i][s] = row * S_ROWB + (((h ^ ((row >> 1) & 7)) << 4) ^ (s << 5));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wq * 64 + j * 32 + p;
#pragma unroll
        for (int s = 0; s < 4; ++s) boff[j][s] = S_TILE_BYTES + row * S_ROWB + (((h ^ ((row >> 1) & 7)) << 4) ^ (s << 5));
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define S_STAGE(BUFI, KSI)                                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                           \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)gsrc[j_], (lds_ptr_t)(lbase + (BUFI) * S_BUF + j_ * 4096 - (KSI) * S_ROWB), \
                                             16, (KSI) * S_ROWB, 0);                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                           \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)qsrc[j_],                                                  \
                                             (lds_ptr_t)(lbase + (BUFI) * S_BUF + S_TILE_BYTES + j_ * 4096 - (KSI) * S_ROWB), 16, \
                                             (KSI) * S_ROWB, 0);                                                   \
    } while (0)

    if (tile_begin < tile_end) {
        tile_ptrs(tile_begin);
        S_STAGE(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

#pragma unroll 1
    for (long tile = tile_begin; tile < tile_end; ++tile) {
        const bool more = tile + 1 < tile_end;
        auto slice = [&](auto KSC) {
            constexpr int ks = decltype(KSC)::value;
            constexpr int buf = ks & 1;
            if constexpr (ks + 1 < S_KSL) {
                S_STAGE(buf ^ 1, ks + 1);
            } else {
                if (more) {
                    tile_ptrs(tile + 1);
                    S_STAGE(buf ^ 1, 0);
                }
            }
            const unsigned char* bb = smem + S_PAD + buf * S_BUF;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(bb + aoff[i][s]);
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(bb + boff[j][s]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            if constexpr (ks == S_KSL - 1) {
                // ---- selection: lane owns query column (wq*64 + j*32 + p); accumulator rows = gallery
                const long gb = tile * 128 + wg * 64;
                const bool partial = gb + 64 > a.G;                 // only the last tile can hold rows >= G
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (partial) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (gb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= a.G) acc[i][j][r] = -INFINITY;
                        }
                        float mx = acc[i][j][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
                        if (__any(mx > ts[j][KK - 1])) {
                            const int g0 = (int)gb + i * 32 + 4 * h;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float s = acc[i][j][r];
                                if (s > ts[j][KK - 1]) {
                                    int gi = g0 + (r & 3) + 8 * (r >> 2);
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
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        static_assert(S_KSL == 8, "slice unroll below assumes 8 slices per tile");
        slice(std::integral_constant<int, 0>{}); slice(std::integral_constant<int, 1>{});
        slice(std::integral_constant<int, 2>{}); slice(std::integral_constant<int, 3>{});
        slice(std::integral_constant<int, 4>{}); slice(std::integral_constant<int, 5>{});
        slice(std::integral_constant<int, 6>{}); slice(std::integral_constant<int, 7>{});
    }
#undef S_STAGE

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
__global__ void topk_select_kernel(const float* s, const TI* idx, int nlists, long lstride, long lstride_i, long Q, int k,
                                   long idx_offset, float* out_s, int64_t* out_i) {
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float bs[SEARCH_KMAX]; int64_t bi[SEARCH_KMAX];
#pragma unroll
    for (int e = 0; e < SEARCH_KMAX; ++e) { bs[e] = -INFINITY; bi[e] = INT64_MAX; }
    for (int l = 0; l < nlists; ++l) {
        for (int c = 0; c < k; ++c) {
            float v = s[l * lstride + q * k + c];
            int64_t vi = (int64_t)idx[l * lstride_i + q * k + c];
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


// ---- resident-query form -------------------------------------------------------------------------------------
// search_kernel stages a 128-query slice next to every 128-row gallery slice: 1 filled LDS byte per 64 flop, and
// its L2 -> LDS fill sits at the ~28 B/clk/CU that the micro-architecture guide gives as the ceiling for row
// gathers from L2 ("bound by the L2 -> LDS fill").  Here the queries never pass through LDS: a wave
// owns 32 queries and keeps their MFMA B fragments for the whole K = 512 in registers (32 x bf16x8 = 128 VGPRs,
// loaded once per block); a block is 8 waves = 256 queries; the gallery goes through LDS as whole 64-row tiles
// (64 KB, double-buffered, one LDS-DMA instruction per 1-KB row, 16-byte pieces XOR-swizzled with row & 15 on the
// source side), shared by the 8 waves: 1 filled byte per 256 flop, one LDS read per MFMA, and ONE wait + barrier
// per tile (64 MFMAs per wave) instead of one per K slice (16).
// Selection as in search_kernel: a lane owns one query and sees 16 of the 32 rows of each accumulator tile (its
// half h), so there are 2 candidate lists per query and split.
#define RQ_WALK_GRP 16                   // tiles per group of the strided walk (below)

// Tile visiting order of the resident-query kernels: tile j of a split is tile_begin + (j * S mod T) with S ~ 0.618 T,
// coprime with T (a low-discrepancy walk over the split).  On a gallery whose scores TREND with the row index -- e.g.
// rows appended in order of quality, a gallery sorted by similarity to a cluster centre -- an in-order walk makes every
// tile beat the running top-k of every lane (the insertion code then runs on every tile: measured 2.0x slower,
// profiles/r02_search_adversarial_10M_x_65536.json); the strided walk samples the whole range in its first tiles, so the
// thresholds are high from the start and insertions die out as on random data.  The per-lane lists are then ordered by
// (score desc, index asc) explicitly, because "first seen wins" no longer means "lower index wins".
FR_DEV long rq_tile_stride(long T) {
    if (T < 3) return 1;
    long s = (long)((double)T * 0.6180339887498949);
    if (s < 1) s = 1;
    for (;; ++s) {
        long x = s, y = T;
        while (y) { const long t = x % y; x = y; y = t; }
        if (x == 1) return s;
    }
}
#define RQ_QB 256                        // queries per block
#define RQ_GT 64                         // gallery rows per tile
#define RQ_TILE_BYTES (RQ_GT * SEARCH_D * 2)
#define RQ_LISTS 2
#define RQ_GRP 2                         // k-steps per group of A fragment reads (one group in flight ahead of the MFMAs)

template <int KK>
__global__ __launch_bounds__(512, 1) void search_rq_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const long q0 = (long)blockIdx.x * RQ_QB;
    const int split = blockIdx.y;
    const long tile_begin = (long)split * a.tiles_per_split;
    const long ntiles_all = (a.G + RQ_GT - 1) / RQ_GT;
    const long tile_end = min(tile_begin + a.tiles_per_split, ntiles_all);

    float ts[KK]; int ti[KK];
#pragma unroll
    for (int e = 0; e < KK; ++e) { ts[e] = -INFINITY; ti[e] = 0x7FFFFFFF; }

    // gallery tile -> LDS: this wave loads rows wave, wave + 8, ... of a tile; lane l writes physical piece l of
    // the row and fetches logical piece l ^ (row & 15)
    const unsigned char* gbase = reinterpret_cast<const unsigned char*>(a.g);
    const int off_even = (lane ^ wave) << 4, off_odd = (lane ^ (wave | 8)) << 4;
    auto stage = [&](long tile, int buf) {
#pragma unroll
        for (int j = 0; j < RQ_GT / 8; ++j) {
            const int r = wave + 8 * j;
            const long gr = min(tile * RQ_GT + r, a.G - 1);
            const unsigned char* src = gbase + gr * (SEARCH_D * 2) + ((j & 1) ? off_odd : off_even);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + buf * RQ_TILE_BYTES + r * 1024), 16, 0, 0);
        }
    };
    const int T = (int)(tile_end - tile_begin);
    const int NG = (T + RQ_WALK_GRP - 1) / RQ_WALK_GRP;
    const int S = a.walk ? (int)rq_tile_stride(NG) : 1;
    auto walk_next = [&](int p_) {                          // see search_rq16_kernel
        int g = p_ / RQ_WALK_GRP, w = p_ % RQ_WALK_GRP + 1;
        if (w == RQ_WALK_GRP || g * RQ_WALK_GRP + w >= T) { w = 0; g += S; if (g >= NG) g -= NG; }
        return g * RQ_WALK_GRP + w;
    };
    int p0 = 0, p1 = T > 1 ? walk_next(0) : 0, p2 = T > 2 ? walk_next(p1) : 0;
    if (tile_begin < tile_end) stage(tile_begin + p0, 0);

    // this lane's query (column p of the wave's 32) as B fragments of the 32 k-steps
    bf16x8 bq[32];
    {
        const long qr = min(q0 + wave * 32 + p, a.Q - 1);
        const bf16x8* src = reinterpret_cast<const bf16x8*>(a.q + qr * SEARCH_D + 8 * h);
#pragma unroll
        for (int s = 0; s < 32; ++s) bq[s] = src[2 * s];
    }
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = (i * 32 + p) * 1024 + ((h ^ (p & 15)) << 4);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;      // 32-bit LDS address of the tiles
    u32x4 af[2][RQ_GRP][2];
    unsigned ka[8];                                        // LDS address of k-step s: ka[s & 7] + (s >> 3) * 256 (+ 32768 for i = 1)
    auto rd = [](auto SC, unsigned addr, u32x4& d0, u32x4& d1) {
        constexpr int s_ = decltype(SC)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d0) : "v"(addr), "n"((s_ >> 3) * 256) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d1) : "v"(addr), "n"((s_ >> 3) * 256 + 32768) : "memory");
    };
    // Order of a tile: MFMAs -> wait for the next tile's DMA + barrier -> request the tile after next into the
    // buffer just finished -> request the first A fragments of the next tile -> selection of this tile (its VALU
    // work covers the barrier skew and the latency of those first reads) -> next tile.
    if (T > 1) stage(tile_begin + p1, 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) ka[j] = lds0 + (abase[0] ^ (j << 5));
    if (tile_begin < tile_end) {
        rd(std::integral_constant<int, 0>{}, ka[0], af[0][0][0], af[0][0][1]);
        rd(std::integral_constant<int, 1>{}, ka[1], af[0][1][0], af[0][1][1]);
    }
    int buf = 0;
#pragma unroll 1
    for (int tj = 0; tj < T; ++tj) {
        const long tile = tile_begin + p0;
        // A fragments are read in groups of RQ_GRP k-steps, one group ahead of the MFMAs that use them.  The reads
        // and their waits are inline asm: through the compiler, every second group waited with lgkmcnt(0) right
        // after the NEXT group's reads had been issued (a full LDS round trip exposed per 8 MFMAs).  LDS reads
        // return in order, so lgkmcnt(2 * RQ_GRP) = "all but the group just requested"; the wait names the
        // fragments it releases as in/out operands, which orders the MFMAs that consume them after it.
        auto group = [&](auto GC) {
            constexpr int g = decltype(GC)::value;
            if constexpr (g + 1 < 32 / RQ_GRP) {
                rd(std::integral_constant<int, (g + 1) * RQ_GRP>{}, ka[((g + 1) * RQ_GRP) & 7], af[(g + 1) & 1][0][0], af[(g + 1) & 1][0][1]);
                rd(std::integral_constant<int, (g + 1) * RQ_GRP + 1>{}, ka[((g + 1) * RQ_GRP + 1) & 7], af[(g + 1) & 1][1][0], af[(g + 1) & 1][1][1]);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[g & 1][0][0]), "+v"(af[g & 1][0][1]), "+v"(af[g & 1][1][0]), "+v"(af[g & 1][1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[g & 1][0][0]), "+v"(af[g & 1][0][1]), "+v"(af[g & 1][1][0]), "+v"(af[g & 1][1][1]));
            }
#pragma unroll
            for (int u = 0; u < RQ_GRP; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[g & 1][u][i]), bq[g * RQ_GRP + u], acc[i], 0, 0, 0);
        };
        static_assert(RQ_GRP == 2, "the group reads below are written for two k-steps per group");
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
        group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{}); group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
        group(std::integral_constant<int, 8>{}); group(std::integral_constant<int, 9>{}); group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
        group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{}); group(std::integral_constant<int, 14>{}); group(std::integral_constant<int, 15>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's rows of the next tile have landed
        __syncthreads();                                         // ... everyone's; and everyone is done reading this tile
        if (tj + 2 < T) stage(tile_begin + p2, buf);
        if (tj + 1 < T) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ka[j] = lds0 + (buf ^ 1) * RQ_TILE_BYTES + (abase[0] ^ (j << 5));
            rd(std::integral_constant<int, 0>{}, ka[0], af[0][0][0], af[0][0][1]);
            rd(std::integral_constant<int, 1>{}, ka[1], af[0][1][0], af[0][1][1]);
        }
        p0 = p1; p1 = p2; p2 = walk_next(p2);
        // ---- selection: accumulator rows = gallery rows tile*64 + i*32 + (r&3) + 8*(r>>2) + 4h
        const long gb = tile * RQ_GT;
        const bool partial = gb + RQ_GT > a.G;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (partial) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (gb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= a.G) acc[i][r] = __builtin_nanf("");
            }
            float mx = acc[i][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[i][r]);
            if (__any(mx >= ts[KK - 1])) {
                const int g0 = (int)gb + i * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sc = acc[i][r];
                    if (sc >= ts[KK - 1]) {                  // (score desc, index asc): see search_rq16_kernel; a masked row is NaN
                        int gi = g0 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                        for (int e = 0; e < KK; ++e) {
                            if (sc > ts[e] || (sc == ts[e] && gi < ti[e])) {
                                const float t = ts[e]; ts[e] = sc; sc = t;
                                const int u = ti[e]; ti[e] = gi; gi = u;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        buf ^= 1;
    }

    const long qrow = q0 + wave * 32 + p;
    const long base = (((long)split * RQ_LISTS + h) * a.Qpad + qrow) * a.k;
    for (int e = 0; e < KK; ++e)
        if (e < a.k) { a.part_s[base + e] = ts[e]; a.part_i[base + e] = ti[e]; }
}

// ---- resident queries on v_mfma_f32_16x16x32_bf16 -------------------------------------------------------------
// Same structure as search_rq_kernel (queries as B fragments in registers, 64-row gallery tiles through LDS, one
// wait + barrier per tile), on the 16x16x32 MFMA shape: the micro-architecture guide measures bf16 loops on random
// data at ~1.12-1.15x the FLOP/s of the 32x32x16 shape at equal cycles per FLOP, because the chip holds a higher
// clock under that instruction.  Per wave and tile: 16 k-steps x (4 row blocks of 16 gallery rows) x (2 column
// blocks of 16 queries) = 128 MFMAs of 16 cycles; 64 A-fragment reads (ds_read_b128), as before; 8 accumulators
// of 4 registers.  Accumulator layout: lane l holds column (query) l & 15 and rows 4 * (l >> 4) + r, so a query's
// 64 scores of a tile are spread over 4 lanes (16 each): 4 candidate lists per query and split, and every lane
// keeps two lists (one per column block).
#define RQ16_LISTS 4
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int KK>
__global__ __launch_bounds__(512, 1) void search_rq16_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, kg = lane >> 4;
    const long q0 = (long)blockIdx.x * RQ_QB;
    const int split = blockIdx.y;
    const long tile_begin = (long)split * a.tiles_per_split;
    const long ntiles_all = (a.G + RQ_GT - 1) / RQ_GT;
    const long tile_end = min(tile_begin + a.tiles_per_split, ntiles_all);

    float ts[2][KK]; int ti[2][KK];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < KK; ++e) { ts[cb][e] = -INFINITY; ti[cb][e] = 0x7FFFFFFF; }

    // gallery tile -> LDS exactly as in search_rq_kernel: physical 16-byte piece l of row r holds logical piece l ^ (r & 15)
    const unsigned char* gbase = reinterpret_cast<const unsigned char*>(a.g);
    const int off_even = (lane ^ wave) << 4, off_odd = (lane ^ (wave | 8)) << 4;
    auto stage = [&](long tile, int buf) {
#pragma unroll
        for (int j = 0; j < RQ_GT / 8; ++j) {
            const int r = wave + 8 * j;
            const long gr = min(tile * RQ_GT + r, a.G - 1);
            const unsigned char* src = gbase + gr * (SEARCH_D * 2) + ((j & 1) ? off_odd : off_even);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + buf * RQ_TILE_BYTES + r * 1024), 16, 0, 0);
        }
    };
    // the walk strides over GROUPS of RQ_WALK_GRP consecutive tiles (1 MB of gallery): a tile-by-tile stride cost 3 % on
    // random data (every 64-KB tile in another page), groups cost nothing and sample a trend just as well
    const int T = (int)(tile_end - tile_begin);
    const int NG = (T + RQ_WALK_GRP - 1) / RQ_WALK_GRP;
    const int S = a.walk ? (int)rq_tile_stride(NG) : 1;
    auto walk_next = [&](int p) {                           // p = group * RQ_WALK_GRP + tile within the group
        int g = p / RQ_WALK_GRP, w = p % RQ_WALK_GRP + 1;
        if (w == RQ_WALK_GRP || g * RQ_WALK_GRP + w >= T) { w = 0; g += S; if (g >= NG) g -= NG; }
        return g * RQ_WALK_GRP + w;
    };
    int p0 = 0, p1 = T > 1 ? walk_next(0) : 0, p2 = T > 2 ? walk_next(p1) : 0;   // current, next and after-next tile
    if (tile_begin < tile_end) stage(tile_begin + p0, 0);

    // B fragments of this lane's two queries (column c16 of column blocks 0 and 1): k = 32 s + 8 kg + j
    bf16x8 bq[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const long qr = min(q0 + wave * 32 + cb * 16 + c16, a.Q - 1);
        const bf16x8* src = reinterpret_cast<const bf16x8*>(a.q + qr * SEARCH_D + 8 * kg);
#pragma unroll
        for (int s = 0; s < 16; ++s) bq[cb][s] = src[4 * s];
    }
    // A fragment of row block rb, k-step s: row rb*16 + c16, logical piece 4 s + kg -> physical (4 s + kg) ^ c16
    const int abase = c16 * 1024 + ((kg ^ c16) << 4);

    f32x4_t acc[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    u32x4 af[2][4];
    unsigned ka[4];                                        // LDS address of k-step s: ka[s & 3] + (s >> 2) * 256 (+ rb * 16384)
    auto rd4 = [](auto SC, unsigned addr, u32x4& d0, u32x4& d1, u32x4& d2, u32x4& d3) {
        constexpr int s_ = decltype(SC)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d0) : "v"(addr), "n"((s_ >> 2) * 256) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d1) : "v"(addr), "n"((s_ >> 2) * 256 + 16384) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d2) : "v"(addr), "n"((s_ >> 2) * 256 + 32768) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d3) : "v"(addr), "n"((s_ >> 2) * 256 + 49152) : "memory");
    };
    if (T > 1) stage(tile_begin + p1, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) ka[j] = lds0 + (abase ^ (j << 6));
    if (tile_begin < tile_end) rd4(std::integral_constant<int, 0>{}, ka[0], af[0][0], af[0][1], af[0][2], af[0][3]);
    int buf = 0;
#pragma unroll 1
    for (int tj = 0; tj < T; ++tj) {
        const long tile = tile_begin + p0;
        auto step = [&](auto SC) {
            constexpr int s = decltype(SC)::value;
            if constexpr (s + 1 < 16) {
                rd4(std::integral_constant<int, s + 1>{}, ka[(s + 1) & 3], af[(s + 1) & 1][0], af[(s + 1) & 1][1], af[(s + 1) & 1][2], af[(s + 1) & 1][3]);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[s & 1][0]), "+v"(af[s & 1][1]), "+v"(af[s & 1][2]), "+v"(af[s & 1][3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[s & 1][0]), "+v"(af[s & 1][1]), "+v"(af[s & 1][2]), "+v"(af[s & 1][3]));
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[s & 1][rb]), bq[cb][s], acc[rb][cb], 0, 0, 0);
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's rows of the next tile have landed
        __syncthreads();                                         // ... everyone's; and everyone is done reading this tile
        if (tj + 2 < T) stage(tile_begin + p2, buf);
        if (tj + 1 < T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ka[j] = lds0 + (buf ^ 1) * RQ_TILE_BYTES + (abase ^ (j << 6));
            rd4(std::integral_constant<int, 0>{}, ka[0], af[0][0], af[0][1], af[0][2], af[0][3]);
        }
        p0 = p1; p1 = p2; p2 = walk_next(p2);
        // ---- selection: accumulator rows = gallery rows tile*64 + rb*16 + 4*kg + r, column = this lane's query of block cb
        const long gb = tile * RQ_GT;
        const bool partial = gb + RQ_GT > a.G;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (partial) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gb + rb * 16 + 4 * kg + r >= a.G) acc[rb][cb][r] = __builtin_nanf("");
            }
            float mx = acc[0][cb][0];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[rb][cb][r]);
            if (__any(mx >= ts[cb][KK - 1])) {
                const int g0 = (int)gb + 4 * kg;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sc = acc[rb][cb][r];
                        // (score desc, index asc): tiles are not visited in index order.  A masked row (NaN) never enters.
                        if (sc >= ts[cb][KK - 1]) {
                            int gi = g0 + rb * 16 + r;
#pragma unroll
                            for (int e = 0; e < KK; ++e) {
                                if (sc > ts[cb][e] || (sc == ts[cb][e] && gi < ti[cb][e])) {
                                    const float t = ts[cb][e]; ts[cb][e] = sc; sc = t;
                                    const int u = ti[cb][e]; ti[cb][e] = gi; gi = u;
                                }
                            }
                        }
                    }
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        buf ^= 1;
    }

#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const long qrow = q0 + wave * 32 + cb * 16 + c16;
        const long base = (((long)split * RQ16_LISTS + kg) * a.Qpad + qrow) * a.k;
        for (int e = 0; e < KK; ++e)
            if (e < a.k) { a.part_s[base + e] = ts[cb][e]; a.part_i[base + e] = ti[cb][e]; }
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once for each device a
// search runs on (a process-wide flag made the first search on a second GPU fail at launch), under a lock so that
// host threads searching on different streams do not race on the flags.
#include <mutex>
static int search_set_lds_limits() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    FR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && done[dev]) return FR_OK;
    const int rq = 2 * RQ_TILE_BYTES, sl = 2 * S_BUF + S_PAD;
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_rq_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, rq));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_rq_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, rq));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_rq_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, rq));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_rq16_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, rq));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_rq16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, rq));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
    FR_HIP_CHECK(hipFuncSetAttribute((const void*)search_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, sl));
    if (dev >= 0 && dev < 64) done[dev] = true;
    return FR_OK;
}

static void search_geometry_rq(long Q, long G, int* nsplit, long* tiles_per_split, long* Qpad) {
    const long qtiles = (Q + RQ_QB - 1) / RQ_QB, gtiles = (G + RQ_GT - 1) / RQ_GT;
    static const long want = [] { const char* e = getenv("FR_SEARCH_BLOCKS"); return e && atol(e) > 0 ? atol(e) : 256L; }();   // tuning
    long ns = (want + qtiles - 1) / qtiles;        // >= 256 blocks: one 8-wave block per CU (measured: 256 / 512 / 1024 -> 1324 / 1316 / 1240 TFLOP/s)
    if (ns > gtiles) ns = gtiles;
    if (ns < 1) ns = 1;
    const long tps = (gtiles + ns - 1) / ns;
    ns = (gtiles + tps - 1) / tps;
    *nsplit = (int)ns; *tiles_per_split = tps; *Qpad = qtiles * RQ_QB;
}

static void search_geometry(long Q, long G, int* nsplit, long* tiles_per_split, long* Qpad) {
    const long qtiles = (Q + 127) / 128, gtiles = (G + 127) / 128;
    long ns = (512 + qtiles - 1) / qtiles;         // aim for >= 512 blocks (2 per CU); long splits keep top-k inserts rare
    if (ns > gtiles) ns = gtiles;
    if (ns < 1) ns = 1;
    const long tps = (gtiles + ns - 1) / ns;
    ns = (gtiles + tps - 1) / tps;
    *nsplit = (int)ns; *tiles_per_split = tps; *Qpad = qtiles * 128;
}

// Queries are processed in chunks of at most SEARCH_QCHUNK per launch: with more query tiles than
// resident blocks every block walks the whole gallery on its own schedule and the gallery is
// streamed from HBM once per block instead of once per XCD (65536 queries in one launch: 632
// TFLOP/s; in chunks of 16384: 997).
// read per call (two getenv per search, negligible next to a launch) so that a test can run both kernels in one process
static bool search_use_rq() {
    const char* s = getenv("FR_SEARCH_RQ");              // default on; 0 = search_kernel
    return !s || atoi(s) != 0;
}
static bool search_use_mfma16() {
    const char* s = getenv("FR_SEARCH_MFMA");            // default 16: v_mfma_f32_16x16x32_bf16; 32 = the 32x32x16 form
    return !s || atoi(s) != 32;
}
static long search_qchunk() {
    const char* e = getenv("FR_SEARCH_QCHUNK");          // tuning
    return e && atol(e) >= 256 ? atol(e) / 256 * 256 : (search_use_rq() ? 32768L : 16384L);
}
#define SEARCH_QCHUNK search_qchunk()

extern "C" int fr_topk_workspace_bytes(int64_t Q, int64_t G, int k, size_t* out_bytes) {
    if (Q <= 0 || G <= 0 || k < 1 || k > SEARCH_KMAX || !out_bytes) FR_FAIL(FR_ERR_ARG, "fr_topk_workspace_bytes: bad arguments");
    // the chunks of one call share the workspace: a full chunk and the (shorter) last one have different split
    // counts -- fewer query tiles mean more gallery splits -- so the size is the larger of the two
    size_t need = 0;
    const int64_t sizes[2] = {std::min<int64_t>(Q, SEARCH_QCHUNK), Q > SEARCH_QCHUNK ? Q % SEARCH_QCHUNK : 0};
    for (int64_t qc : sizes) {
        if (qc <= 0) continue;
        int ns; long tps, qpad;
        search_geometry(qc, G, &ns, &tps, &qpad);
        need = std::max(need, (size_t)ns * SEARCH_LISTS * qpad * k * 8 + 512);
        search_geometry_rq(qc, G, &ns, &tps, &qpad);            // either form may run (FR_SEARCH_RQ)
        need = std::max(need, (size_t)ns * RQ16_LISTS * qpad * k * 8 + 512);     // RQ16_LISTS >= RQ_LISTS
    }
    *out_bytes = need;
    return FR_OK;
}

static int search_check_args(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int d, int k,
                             const float* out_scores, const int64_t* out_idx) {
    if (!q || !g || !out_scores || !out_idx || Q <= 0 || G <= 0) FR_FAIL(FR_ERR_ARG, "fr_topk_search: bad arguments");
    if (d != SEARCH_D) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search: d must be 512");
    if (k < 1 || k > SEARCH_KMAX) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search: k must be in 1..8");
    if (G > 0x7FFFFF00L) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search: at most 2^31 gallery rows per call (shard it)");
    return FR_OK;
}

// One search with the partial top-k lists in `workspace` (already checked to hold fr_topk_workspace_bytes(Q,G,k)).
static int search_run(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int k,
                      int64_t idx_offset, float* out_scores, int64_t* out_idx, void* workspace, void* stream) {
    if (Q > SEARCH_QCHUNK) {                  // chunks run back to back on the stream and share the workspace
        for (int64_t c0 = 0; c0 < Q; c0 += SEARCH_QCHUNK) {
            const int64_t qc = std::min<int64_t>(SEARCH_QCHUNK, Q - c0);
            const int rc = search_run(q + c0 * SEARCH_D, g, qc, G, k, idx_offset, out_scores + c0 * k, out_idx + c0 * k,
                                      workspace, stream);
            if (rc) return rc;
        }
        return FR_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    SearchArgs a{};
    a.q = reinterpret_cast<const bf16*>(q); a.g = reinterpret_cast<const bf16*>(g);
    a.Q = Q; a.G = G; a.k = k;
    const bool rq = search_use_rq();
    { const char* w = getenv("FR_SEARCH_WALK"); a.walk = !(w && atoi(w) == 0); }         // tuning / comparison: 0 = tiles in index order
    if (rq) search_geometry_rq(Q, G, &a.nsplit, &a.tiles_per_split, &a.Qpad);
    else search_geometry(Q, G, &a.nsplit, &a.tiles_per_split, &a.Qpad);
    const bool m16 = rq && k <= 5 && search_use_mfma16();      // k = 6..8 would spill registers in the 16x16x32 form
    const int lists = rq ? (m16 ? RQ16_LISTS : RQ_LISTS) : SEARCH_LISTS;
    unsigned char* ws = reinterpret_cast<unsigned char*>(align_up((size_t)workspace, 256));
    const size_t nel = (size_t)a.nsplit * lists * a.Qpad * k;
    a.part_s = reinterpret_cast<float*>(ws);
    a.part_i = reinterpret_cast<int*>(ws + nel * 4);
    if (rq) {
        const size_t lds = 2 * RQ_TILE_BYTES;
        dim3 grid((unsigned)(a.Qpad / RQ_QB), (unsigned)a.nsplit);
        { const int rc = search_set_lds_limits(); if (rc) return rc; }
        if (m16) {
            if (k == 1) hipLaunchKernelGGL((search_rq16_kernel<1>), grid, dim3(512), lds, st, a);
            else hipLaunchKernelGGL((search_rq16_kernel<5>), grid, dim3(512), lds, st, a);
        } else {
            if (k == 1) hipLaunchKernelGGL((search_rq_kernel<1>), grid, dim3(512), lds, st, a);
            else if (k <= 5) hipLaunchKernelGGL((search_rq_kernel<5>), grid, dim3(512), lds, st, a);
            else hipLaunchKernelGGL((search_rq_kernel<8>), grid, dim3(512), lds, st, a);
        }
        FR_LAUNCH_CHECK("search_rq_kernel");
        hipLaunchKernelGGL((topk_select_kernel<int>), dim3((unsigned)((Q + 127) / 128)), dim3(128), 0, st,
                           a.part_s, a.part_i, a.nsplit * lists, a.Qpad * (long)k, a.Qpad * (long)k, (long)Q, k, (long)idx_offset,
                           out_scores, out_idx);
        FR_LAUNCH_CHECK("topk_select_kernel");
        return FR_OK;
    }
    const size_t lds = 2 * S_BUF + S_PAD;
    dim3 grid((unsigned)(a.Qpad / 128), (unsigned)a.nsplit);
    { const int rc = search_set_lds_limits(); if (rc) return rc; }
    if (k == 1) hipLaunchKernelGGL((search_kernel<1>), grid, dim3(256), lds, st, a);
    else if (k <= 5) hipLaunchKernelGGL((search_kernel<5>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((search_kernel<8>), grid, dim3(256), lds, st, a);
    FR_LAUNCH_CHECK("search_kernel");
    hipLaunchKernelGGL((topk_select_kernel<int>), dim3((unsigned)((Q + 127) / 128)), dim3(128), 0, st,
                       a.part_s, a.part_i, a.nsplit * SEARCH_LISTS, a.Qpad * (long)k, a.Qpad * (long)k, (long)Q, k, (long)idx_offset,
                       out_scores, out_idx);
    FR_LAUNCH_CHECK("topk_select_kernel");
    return FR_OK;
}

extern "C" int fr_topk_search(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int d, int k,
                              int64_t idx_offset, float* out_scores, int64_t* out_idx,
                              void* workspace, size_t workspace_bytes, void* stream) {
    { const int rc = search_check_args(q, g, Q, G, d, k, out_scores, out_idx); if (rc) return rc; }
    size_t need = 0;
    fr_topk_workspace_bytes(Q, G, k, &need);
    if (!workspace || workspace_bytes < need) FR_FAIL(FR_ERR_WORKSPACE, "fr_topk_search: workspace too small");
    return search_run(q, g, Q, G, k, idx_offset, out_scores, out_idx, workspace, stream);
}

// ---- the entry point that allocates nothing ------------------------------------------------------------------
// fr_topk_search_noalloc keeps the partial top-k lists (the only scratch buffer of the search: what
// fr_topk_search takes as `workspace`) in this file-scope device array.  In host code the NAME `scratch` is only a
// stand-in for the symbol, not a device address: the address of each GPU's copy is asked for once per device with
// hipGetSymbolAddress and kept in scratch_ptr[]; kernels, like everything else on the host side, only ever see
// that pointer.
#define FR_STATIC_SCRATCH_BYTES (32u << 20)
static __device__ __align__(256) unsigned char scratch[FR_STATIC_SCRATCH_BYTES];

static int search_static_scratch(void** out) {
    static std::mutex mu;
    static void* scratch_ptr[64] = {nullptr};
    int dev = 0;
    FR_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) FR_FAIL(FR_ERR_UNSUPPORTED, "fr_topk_search_noalloc: device index beyond 63");
    std::lock_guard<std::mutex> lock(mu);
    if (!scratch_ptr[dev]) {
        void* p = nullptr;
        FR_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));       // the current device's copy of the array
        if (!p) FR_FAIL(FR_ERR_HIP, "fr_topk_search_noalloc: hipGetSymbolAddress returned a null address");
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return FR_OK;
}

extern "C" int fr_topk_search_noalloc(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int d, int k,
                                      int64_t idx_offset, float* out_scores, int64_t* out_idx, void* stream) {
    void* ws = nullptr;
    { const int rc = search_static_scratch(&ws); if (rc) return rc; }      // first call on a device: looks the array up
    { const int rc = search_set_lds_limits(); if (rc) return rc; }
    if (Q == 0) return FR_OK;                                              // set-up only (e.g. before a graph capture)
    { const int rc = search_check_args(q, g, Q, G, d, k, out_scores, out_idx); if (rc) return rc; }
    size_t need = 0;
    fr_topk_workspace_bytes(Q, G, k, &need);
    if (need > FR_STATIC_SCRATCH_BYTES) {
        char msg[200];
        snprintf(msg, sizeof msg, "fr_topk_search_noalloc: this search needs %zu bytes of partial lists, the static scratch holds %u "
                 "(use fr_topk_search with a workspace)", need, FR_STATIC_SCRATCH_BYTES);
        FR_FAIL(FR_ERR_WORKSPACE, msg);
    }
    return search_run(q, g, Q, G, k, idx_offset, out_scores, out_idx, ws, stream);
}

extern "C" int fr_topk_merge(const float* scores, const int64_t* idx, int P, int64_t Q, int k,
                             float* out_scores, int64_t* out_idx, void* stream) {
    if (!scores || !idx || P < 1 || Q <= 0 || k < 1 || k > SEARCH_KMAX) FR_FAIL(FR_ERR_ARG, "fr_topk_merge: bad arguments");
    hipLaunchKernelGGL((topk_select_kernel<int64_t>), dim3((unsigned)((Q + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       scores, idx, P, (long)Q * k, (long)Q * k, (long)Q, k, 0L, out_scores, out_idx);
    FR_LAUNCH_CHECK("topk_select_kernel");
    return FR_OK;
}

extern "C" int fr_topk_merge_packed(const void* packed, int P, int64_t Q, int k,
                                    float* out_scores, int64_t* out_idx, void* stream) {
    if (!packed || !out_scores || !out_idx || P < 1 || Q <= 0 || k < 1 || k > SEARCH_KMAX) FR_FAIL(FR_ERR_ARG, "fr_topk_merge_packed: bad arguments");
    if ((size_t)packed % 8) FR_FAIL(FR_ERR_ARG, "fr_topk_merge_packed: the receive buffer must be 8-byte aligned");
    const size_t s_bytes = ((size_t)Q * k * 4 + 7) / 8 * 8, total = s_bytes + (size_t)Q * k * 8;     // one rank's message
    const unsigned char* base = static_cast<const unsigned char*>(packed);
    hipLaunchKernelGGL((topk_select_kernel<int64_t>), dim3((unsigned)((Q + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(base), reinterpret_cast<const int64_t*>(base + s_bytes), P,
                       (long)(total / 4), (long)(total / 8), (long)Q, k, 0L, out_scores, out_idx);
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
