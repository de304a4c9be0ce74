// Plan compiler + executor: turns the reference-level layer list (fr_layer_desc, host fp32 tensors
// straight from a state_dict) into folded/packed device weights and a launch schedule over the
// HIP kernels.  Replaces nn.Module construction / load_state_dict / forward (inference.py:55-96,
// :121-122; evaluate.py:179-194).
#include <math.h>
#include <atomic>
#include <mutex>
#include <set>
#include <map>
#include <array>
#include <memory>
#include <vector>

#include "kernels.h"
#include <cstdlib>

static thread_local std::string g_last_error;
void fr_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* fr_last_error(void) { return g_last_error.c_str(); }

int fr_big_lds_once(const void* kernel) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    FR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return FR_OK;
    FR_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({kernel, dev});
    return FR_OK;
}

namespace {

enum Kind { K_STEM, K_PW, K_DW, K_GDW /*global depthwise (weighted reduce)*/, K_SE, K_LINEAR, K_CONV3 };

struct Layer {
    int kind = 0;
    int src = -1, residual = -1;
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0, act = 0;
    int inC = 0, inH = 0, inW = 0;        // input dims
    int C = 0, H = 0, W = 0;              // output dims
    bool out_f32 = false;                 // fp32-class values in the output buffer (fp32 or f24): split GEMM operands downstream
    bool in_f32 = false;
    int out_fmt = FMT_BF16, in_fmt = FMT_BF16;   // storage format of the output / input buffer (FMT_*; f24: hi + lo planes)
    int slot = -1;                        // output slot
    int slot_pool = -1, slot_hidden = -1; // SE extras (fp32 [N,C], [N,S]); slot = scale [N,C]
    int slot_partial = -1, pool_tiles = 0; // depthwise feeding an SE block: per-tile channel sums [N,pool_tiles,C] fp32
    int conv_spb = 0;                     // conv_halo16_kernel: K slices per barrier (0 = default)
    int splitk = 0;                       // long-K Linear: K slices of fc_splitk_kernel (partial sums in slot_partial); 0 = fc_kernel
    // device parameters
    bf16* w_packed = nullptr;             // GEMM-shaped kernels
    bf16* w_packed2 = nullptr;            // SE fc2
    float* w_f32 = nullptr;               // dw / gdw tap-major weights
    float* bias = nullptr;
    float* bias2 = nullptr;               // SE fc2 bias
    float* slope = nullptr;
    int K = 0, KS = 0, NT = 0, chunk_ks = 0;       // GEMM geometry
    int K2 = 0, KS2 = 0, NT2 = 0, chunk_ks2 = 0;   // SE fc2 geometry
    int conv_bk = 32;                     // K-slice width the conv weights were packed with
    bool conv_halo = false;               // dense 3x3 stride 1: conv_halo_kernel (FR_CONV_HALO, read when the plan is created)
    bool conv_halo16 = false;             // ... in its 16x16x32 MFMA form (weights packed in that kernel's row order)
    bool via_conv = false;                // Linear routed through the conv implicit-GEMM kernel
    bool fused_next = false;              // (reserved: a layer computed inside the next layer's kernel; none today)
    int fuse_src = -1;                    // 1x1 conv: index of the depthwise layer computed in its kernel (dwp_kernel)
    int fuse_src2 = -1;                   // ... and of the expand 1x1 conv in front of that depthwise layer
    // stem -> depthwise -> 1x1 in one launch: on the 1x1 conv the index of the stem layer, on the stem layer the index of that
    // 1x1 conv.  Decided per call: a uint8 input (fr_embed_u8) or a tap on the stem run the stem kernel and the two-layer fusion
    // instead, so the stem keeps its output slot.
    int fuse_stem = -1, stem_fused_by = -1;
};

}  // namespace

struct fr_plan {
    std::vector<Layer> layers;
    std::vector<size_t> slot_bytes;       // per image
    std::vector<void*> allocs;
    int prec = 0, in_c = 0, in_h = 0, in_w = 0;
    int input_slot = -1;                  // NHWC copy of the network input (non-stem first layers)
    int device = 0;
    int launches = 0;
    void* zeros = nullptr;                // 256 B of zeros (padding source of the LDS-DMA conv path)
    float* u8_lut = nullptr;              // ((b / 255) - 0.5) / 0.5 for b = 0..255 (fr_embed_u8)
    // fr_embed runs the two halves of a large batch on the caller's stream and on this auxiliary
    // stream (fork/join by events): each half's kernel tails and launch gaps are filled by the other half
    // The streams are shared by all plans of a device (aux_streams below) and live as long as the process: a GPU process has few
    // hardware queues (4 here) onto which HIP deals its streams in creation order, and a plan whose own auxiliary stream landed on
    // the caller's queue ran its two halves one after the other (measured: the third plan of a process 236 k instead of 257 k img/s
    // in bf16; with GPU_MAX_HW_QUEUES=8 the second one instead).  Events stay per plan.
    static constexpr int MAX_AUX = 3;
    hipStream_t aux[MAX_AUX] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_AUX] = {nullptr, nullptr, nullptr};
    bool act_f32() const { return prec != FR_PREC_BF16; }            // fp32-class arithmetic (FR_PREC_F32 and FR_PREC_F24)
    int act_fmt() const { return prec == FR_PREC_F32 ? FMT_F32 : prec == FR_PREC_F24 ? FMT_F24 : FMT_BF16; }
    ~fr_plan() {
        for (int i = 0; i < MAX_AUX; ++i)
            if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (void* p : allocs) (void)hipFree(p);
    }
};

namespace {

template <typename T>
int upload(fr_plan* pl, const std::vector<T>& host, T** dev) {
    void* p = nullptr;
    FR_HIP_CHECK(hipMalloc(&p, std::max<size_t>(host.size() * sizeof(T), 16)));
    pl->allocs.push_back(p);
    FR_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T*>(p);
    return FR_OK;
}

// Fold BatchNorm into per-output-channel (scale, shift):  y = conv(x)*scale + shift.
void fold_bn(const fr_layer_desc& d, int cout, std::vector<double>& scale, std::vector<double>& shift) {
    scale.assign(cout, 1.0);
    shift.assign(cout, 0.0);
    for (int c = 0; c < cout; ++c) {
        const double b = d.bias ? (double)d.bias[c] : 0.0;
        if (d.bn_weight) {
            const double s = (double)d.bn_weight[c] / sqrt((double)d.bn_var[c] + (double)d.bn_eps);
            scale[c] = s;
            shift[c] = (b - (double)d.bn_mean[c]) * s + (double)d.bn_bias[c];
        } else {
            shift[c] = b;
        }
    }
}

// Pack W[cout][K] (row-major, already folded, double) into MFMA "A" fragments
// [NT][KS][NS][64 lanes][8]: lane (i = lane&31, h = lane>>5), element j = W[32t + perm(i)][16s+8h+j].
// NS = 2 stores hi = bf16(w) and lo = bf16(w - hi).
void pack_gemm_weights(const std::vector<double>& W, int cout, int K, int ns, std::vector<uint16_t>& out,
                       int* KS_out, int* NT_out) {
    const int KS = cdiv(K, 16), NT = cdiv(cout, 32);
    out.assign((size_t)NT * KS * ns * 64 * 8, 0);
    for (int t = 0; t < NT; ++t)
        for (int s = 0; s < KS; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 32 * t + mfma_row_to_cout(lane & 31);
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * (lane >> 5) + j;
                    double w = (co < cout && k < K) ? W[(size_t)co * K + k] : 0.0;
                    const float wf = (float)w;
                    const uint16_t hi = host_f2bf(wf);
                    const size_t base = ((((size_t)t * KS + s) * ns) * 64 + lane) * 8 + j;
                    out[base] = hi;
                    if (ns == 2) out[base + 64 * 8] = host_f2bf((float)(w - (double)host_bf2f(hi)));
                }
            }
    *KS_out = KS;
    *NT_out = NT;
}

// Pack W(cout, tap, c) into the LDS image of conv_gemm_kernel: [cout tile][ks][NS][BN rows][32].
// row16: the row order of conv_halo16_kernel (16x16x32 MFMA): row rr of a wave's 64 couts holds cout
// 16 * ((rr % 16) / 4) + 4 * (rr / 16) + rr % 4, so that lane group kg collects couts 16 kg .. 16 kg + 15.
template <typename F>
void pack_conv_weights(F&& wfn, int cout, int out_hw, bool halo, int cin, int taps, int ns, int bk, std::vector<uint16_t>& out, bool row16 = false) {
    const int bn = conv_block_couts(cout, out_hw, halo), ntiles = cdiv(cout, bn), chunks = cin / bk, KS = taps * chunks;
    out.assign((size_t)ntiles * KS * ns * bn * bk, 0);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ks = 0; ks < KS; ++ks) {
            // K order: channel chunk outer, tap inner -- the 9 taps of one 64-channel chunk re-read the same
            // 128-byte pixel segments in consecutive K slices, which keeps them in L2 (tap-outer order had the
            // per-XCD footprint above the 4 MB L2: 74 % hit rate on the 256-channel SphereFace layers)
            const int tap = ks % taps, c0 = (ks / taps) * bk;
            for (int r = 0; r < bn; ++r) {
                const int rr = r % 64;
                const int co = row16 ? nt * bn + (r / 64) * 64 + 16 * ((rr % 16) / 4) + 4 * (rr / 16) + rr % 4
                                     : nt * bn + (r / 32) * 32 + mfma_row_to_cout(r % 32);
                if (co >= cout) continue;
                for (int kk = 0; kk < bk; ++kk) {
                    const double w = wfn(co, tap, c0 + kk);
                    const uint16_t hi = host_f2bf((float)w);
                    const size_t base = ((((size_t)nt * KS + ks) * ns) * bn + r) * bk + kk;
                    out[base] = hi;
                    if (ns == 2) out[base + (size_t)bn * bk] = host_f2bf((float)(w - (double)host_bf2f(hi)));
                }
            }
        }
}

std::vector<float> padded(const std::vector<double>& v, int n) {
    std::vector<float> o(n, 0.f);
    for (size_t i = 0; i < v.size(); ++i) o[i] = (float)v[i];
    return o;
}

int alloc_slot(std::vector<size_t>& sizes, std::vector<char>& busy, size_t need) {
    int best = -1;
    for (size_t i = 0; i < sizes.size(); ++i)
        if (!busy[i] && (best < 0 || sizes[i] > sizes[best])) best = (int)i;
    if (best < 0) { sizes.push_back(need); busy.push_back(1); return (int)sizes.size() - 1; }
    if (sizes[best] < need) sizes[best] = need;
    busy[best] = 1;
    return best;
}

int build_layer(fr_plan* pl, const fr_layer_desc& d, int idx, Layer& L) {
    const bool af32 = pl->act_f32();
    char where[64];
    snprintf(where, sizeof where, "layer %d: ", idx);
    L.src = d.src; L.residual = d.residual; L.act = d.act;
    if (d.src >= idx || d.residual >= idx || d.src < -1 || d.residual < -1)
        FR_FAIL(FR_ERR_ARG, std::string(where) + "src/residual must refer to earlier layers");
    const int afmt = pl->act_fmt();
    if (d.src < 0) { L.inC = pl->in_c; L.inH = pl->in_h; L.inW = pl->in_w; L.in_f32 = af32; L.in_fmt = afmt; }
    else { const Layer& s = pl->layers[d.src]; L.inC = s.C; L.inH = s.H; L.inW = s.W; L.in_f32 = s.out_f32; L.in_fmt = s.out_fmt; }
    if (d.act == FR_ACT_PRELU && !d.prelu) FR_FAIL(FR_ERR_ARG, std::string(where) + "PReLU without slopes");
    if (!d.weight) FR_FAIL(FR_ERR_ARG, std::string(where) + "missing weight");

    if (d.op == FR_OP_CONV) {
        L.cin = d.cin; L.cout = d.cout; L.k = d.kernel; L.stride = d.stride; L.pad = d.pad;
        if (d.cin != L.inC) FR_FAIL(FR_ERR_ARG, std::string(where) + "cin does not match the source layer");
        if (d.kernel < 1 || d.stride < 1 || d.pad < 0) FR_FAIL(FR_ERR_ARG, std::string(where) + "bad conv geometry");
        L.C = d.cout;
        L.H = (L.inH + 2 * d.pad - d.kernel) / d.stride + 1;
        L.W = (L.inW + 2 * d.pad - d.kernel) / d.stride + 1;
        if (L.H < 1 || L.W < 1) FR_FAIL(FR_ERR_ARG, std::string(where) + "empty output");
        if (d.cout % 8) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "cout must be a multiple of 8");
        std::vector<double> scale, shift;
        fold_bn(d, d.cout, scale, shift);
        std::vector<double> slope_d;
        if (d.act == FR_ACT_PRELU) slope_d.assign(d.prelu, d.prelu + d.cout);
        const int cpad = cdiv(d.cout, 32) * 32;
        L.out_f32 = af32; L.out_fmt = afmt;

        if (d.groups == 1 && d.kernel == 1) {
            if (d.stride != 1 || d.pad != 0) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "1x1 conv must be stride 1, pad 0");
            if (d.cin % 8) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "1x1 conv cin must be a multiple of 8");
            L.kind = K_PW; L.K = d.cin;
            std::vector<double> W((size_t)d.cout * d.cin);
            for (int co = 0; co < d.cout; ++co)
                for (int k = 0; k < d.cin; ++k) W[(size_t)co * d.cin + k] = (double)d.weight[(size_t)co * d.cin + k] * scale[co];
            std::vector<uint16_t> packed;
            pack_gemm_weights(W, d.cout, d.cin, L.in_f32 ? 2 : 1, packed, &L.KS, &L.NT);
            int rc = upload(pl, packed, reinterpret_cast<uint16_t**>(&L.w_packed)); if (rc) return rc;
            int ntb; pw_pick_tiling(L.in_f32, L.NT, L.KS, &ntb, &L.chunk_ks);
        } else if (d.groups == 1 && d.kernel == 3 && d.cin == 3) {
            if (d.src != -1) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "3-channel 3x3 conv must read the network input");
            if (d.pad != 1 || (d.stride != 1 && d.stride != 2)) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "stem conv needs pad 1, stride 1|2");
            L.kind = K_STEM; L.K = 27;
            std::vector<double> W((size_t)d.cout * 27);
            for (int co = 0; co < d.cout; ++co)
                for (int k = 0; k < 27; ++k) W[(size_t)co * 27 + k] = (double)d.weight[(size_t)co * 27 + k] * scale[co];
            std::vector<uint16_t> packed;
            pack_gemm_weights(W, d.cout, 27, af32 ? 2 : 1, packed, &L.KS, &L.NT);
            int rc = upload(pl, packed, reinterpret_cast<uint16_t**>(&L.w_packed)); if (rc) return rc;
        } else if (d.groups == d.cin && d.cin == d.cout) {
            if (d.cin % 8) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "depthwise channels must be a multiple of 8");
            const int kk = d.kernel * d.kernel;
            std::vector<float> wt((size_t)kk * d.cout);
            for (int c = 0; c < d.cout; ++c)
                for (int t = 0; t < kk; ++t) wt[(size_t)t * d.cout + c] = (float)((double)d.weight[(size_t)c * kk + t] * scale[c]);
            int rc = upload(pl, wt, &L.w_f32); if (rc) return rc;
            if (d.pad == 0 && d.kernel == L.inH && d.kernel == L.inW) {
                if (d.act != FR_ACT_NONE) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "global depthwise conv with activation");
                L.kind = K_GDW; L.out_f32 = true; L.out_fmt = FMT_F32;
            } else {
                if ((d.kernel != 3 && d.kernel != 5) || (d.stride != 1 && d.stride != 2) || d.pad != (d.kernel - 1) / 2)
                    FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "depthwise conv: k in {3,5}, stride in {1,2}, same padding");
                L.kind = K_DW;
            }
        } else if (d.groups == 1 && d.kernel == 3) {
            if (d.pad != 1 || (d.stride != 1 && d.stride != 2) || d.cin % 32)
                FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "dense 3x3 conv needs pad 1, stride 1|2, cin % 32 == 0");
            L.kind = K_CONV3;
            if (afmt == FMT_F24) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "dense 3x3 convs have no f24 kernel (FR_PREC_F24 is the depthwise-separable families' mode; use FR_PREC_F32 or FR_PREC_BF16)");
            std::vector<uint16_t> packed;
            const int cin = d.cin;
            const int bk64_min = getenv("FR_CONV_BK64_MIN") ? atoi(getenv("FR_CONV_BK64_MIN")) : 64;   // tuning: smallest cin that uses 64-channel K slices
            const bool halo_on = !getenv("FR_CONV_HALO") || atoi(getenv("FR_CONV_HALO")) != 0;
            L.conv_bk = (!L.in_f32 && d.cin % 64 == 0 && d.cin >= bk64_min) ? 64 : 32;
            L.conv_halo = halo_on && !L.in_f32 && conv_halo_rows(3, d.stride, d.pad, L.inW, 32) > 0;
            if (L.conv_halo) L.conv_bk = 32;                  // the shared-pixel kernel walks 32-channel chunks
            // FR_CONV_MFMA=32 keeps the shared-pixel kernel on the 32x32x16 MFMA shape (default: 16x16x32)
            L.conv_halo16 = L.conv_halo && !af32 && !(getenv("FR_CONV_MFMA") && atoi(getenv("FR_CONV_MFMA")) == 32);
            L.conv_spb = getenv("FR_CONV_SPB") ? atoi(getenv("FR_CONV_SPB")) : 0;      // K slices per barrier of conv_halo16_kernel (tuning; 0 = default)
            pack_conv_weights([&](int co, int tap, int c) { return (double)d.weight[((size_t)co * cin + c) * 9 + tap] * scale[co]; },
                              d.cout, L.H * L.W, L.conv_halo, d.cin, 9, L.in_f32 ? 2 : 1, L.conv_bk, packed, L.conv_halo16);
            int rc = upload(pl, packed, reinterpret_cast<uint16_t**>(&L.w_packed)); if (rc) return rc;
        } else {
            FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "conv shape with no kernel");
        }
        int rc = upload(pl, padded(shift, cpad), &L.bias); if (rc) return rc;
        if (d.act == FR_ACT_PRELU) { rc = upload(pl, padded(slope_d, cpad), &L.slope); if (rc) return rc; }
        if (L.residual >= 0) {
            const Layer& r = pl->layers[L.residual];
            if (r.C != L.C || r.H != L.H || r.W != L.W || r.out_fmt != L.out_fmt || r.kind == K_SE)
                FR_FAIL(FR_ERR_ARG, std::string(where) + "residual shape mismatch");
            if (L.kind != K_PW && L.kind != K_CONV3) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "residual only on 1x1 / 3x3 dense convs");
        }
        if (d.src >= 0 && pl->layers[d.src].kind == K_SE && L.kind != K_PW)
            FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "the consumer of an SE block must be a 1x1 conv");
    } else if (d.op == FR_OP_SE) {
        if (d.src < 0 || !d.weight2 || !d.bias || !d.bias2) FR_FAIL(FR_ERR_ARG, std::string(where) + "SE needs src, fc1/fc2 weights and biases");
        if (d.cin != L.inC || d.cin % 8 || d.cout % 8) FR_FAIL(FR_ERR_ARG, std::string(where) + "SE channel mismatch");
        if (pl->layers[d.src].kind == K_SE) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "SE of SE");
        L.kind = K_SE; L.cin = d.cin; L.cout = d.cout;
        L.C = L.inC; L.H = L.inH; L.W = L.inW; L.out_f32 = L.in_f32; L.out_fmt = L.in_fmt;
        const int C = d.cin, S = d.cout;
        std::vector<double> W1(d.weight, d.weight + (size_t)S * C), W2(d.weight2, d.weight2 + (size_t)C * S);
        std::vector<uint16_t> p1, p2;
        pack_gemm_weights(W1, S, C, 2, p1, &L.KS, &L.NT);  L.K = C;
        pack_gemm_weights(W2, C, S, 2, p2, &L.KS2, &L.NT2); L.K2 = S;
        int rc = upload(pl, p1, reinterpret_cast<uint16_t**>(&L.w_packed)); if (rc) return rc;
        rc = upload(pl, p2, reinterpret_cast<uint16_t**>(&L.w_packed2)); if (rc) return rc;
        int ntb;
        pw_pick_tiling(true, L.NT, L.KS, &ntb, &L.chunk_ks);
        pw_pick_tiling(true, L.NT2, L.KS2, &ntb, &L.chunk_ks2);
        std::vector<double> b1(d.bias, d.bias + S), b2(d.bias2, d.bias2 + C);
        rc = upload(pl, padded(b1, cdiv(S, 32) * 32), &L.bias); if (rc) return rc;
        rc = upload(pl, padded(b2, cdiv(C, 32) * 32), &L.bias2); if (rc) return rc;
    } else if (d.op == FR_OP_LINEAR) {
        if (d.src < 0) FR_FAIL(FR_ERR_ARG, std::string(where) + "Linear needs a source layer");
        const Layer& s = pl->layers[d.src];
        if (s.kind == K_SE) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "Linear after SE");
        if (s.out_fmt == FMT_F24) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "Linear on an f24 activation (only behind the global depthwise conv in f24 plans)");
        const int HW = s.H * s.W, K = s.C * HW;
        if (d.cin != K) FR_FAIL(FR_ERR_ARG, std::string(where) + "in_features does not match the flattened source");
        if (K % 8 || d.cout % 8) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "Linear features must be multiples of 8");
        L.kind = K_LINEAR; L.cin = K; L.cout = d.cout; L.K = K;
        L.C = d.cout; L.H = 1; L.W = 1; L.out_f32 = true; L.out_fmt = FMT_F32;
        std::vector<double> scale, shift;
        fold_bn(d, d.cout, scale, shift);
        // NCHW flatten index c*HW + p  ->  NHWC index p*C + c   (models/sphereface.py:126-127)
        std::vector<double> W((size_t)d.cout * K);
        for (int o = 0; o < d.cout; ++o)
            for (int c = 0; c < s.C; ++c)
                for (int p = 0; p < HW; ++p)
                    W[(size_t)o * K + (size_t)p * s.C + c] = (double)d.weight[(size_t)o * K + (size_t)c * HW + p] * scale[o];
        std::vector<uint16_t> packed;
        int rc;
        if (false && K >= 4096 && K % 32 == 0) {  // (the implicit-GEMM route launched 16 blocks for the SphereFace fc; fc_kernel launches 256)
            L.via_conv = true;
            L.conv_bk = (!L.in_f32 && K % 64 == 0) ? 64 : 32;
            pack_conv_weights([&](int co, int, int c) { return W[(size_t)co * K + c]; }, d.cout, 1, false, K, 1, L.in_f32 ? 2 : 1, L.conv_bk, packed);
        } else {
            pack_gemm_weights(W, d.cout, K, L.in_f32 ? 2 : 1, packed, &L.KS, &L.NT);
            int ntb; pw_pick_tiling(L.in_f32, L.NT, L.KS, &ntb, &L.chunk_ks);
            L.splitk = fc_splitk_slices(K);
        }
        rc = upload(pl, packed, reinterpret_cast<uint16_t**>(&L.w_packed)); if (rc) return rc;
        rc = upload(pl, padded(shift, cdiv(d.cout, 128) * 128), &L.bias); if (rc) return rc;
        if (d.act == FR_ACT_PRELU) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "PReLU after Linear");
        if (d.residual >= 0) FR_FAIL(FR_ERR_UNSUPPORTED, std::string(where) + "residual after Linear");
    } else {
        FR_FAIL(FR_ERR_ARG, std::string(where) + "unknown op");
    }
    return FR_OK;
}

size_t act_bytes_per_image(int fmt, int C, int H, int W) {
    if (fmt == FMT_F24) return (size_t)H * W * ((size_t)C * 2 + f24_lo_pitch(C));     // hi plane + lo plane (of the whole batch, back to back)
    return (size_t)C * H * W * (fmt == FMT_F32 ? 4 : 2);
}
size_t layer_out_bytes(const Layer& L) {           // per image
    return act_bytes_per_image(L.out_fmt, L.C, L.H, L.W);
}

}  // namespace

// The auxiliary streams of a device (created with that device current, on the first plan; never destroyed).  Plans that run
// concurrently from several host threads share them: their second halves queue behind each other on these streams (ordering only;
// every plan forks and joins with its own events).
static int aux_streams(int device, hipStream_t (&out)[fr_plan::MAX_AUX]) {
    static std::mutex mu;
    static std::map<int, std::array<hipStream_t, fr_plan::MAX_AUX>> pool;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find(device);
    if (it == pool.end()) {
        std::array<hipStream_t, fr_plan::MAX_AUX> s{};
        for (int i = 0; i < fr_plan::MAX_AUX; ++i) FR_HIP_CHECK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        it = pool.emplace(device, s).first;
    }
    for (int i = 0; i < fr_plan::MAX_AUX; ++i) out[i] = it->second[i];
    return FR_OK;
}

extern "C" int fr_plan_create(const fr_layer_desc* descs, int n, int in_c, int in_h, int in_w,
                              int precision, fr_plan** out_plan) {
    if (!descs || n <= 0 || !out_plan || in_c <= 0 || in_h <= 0 || in_w <= 0)
        FR_FAIL(FR_ERR_ARG, "fr_plan_create: bad arguments");
    const bool no_fuse = (precision & FR_PLAN_NO_FUSE) != 0;
    precision &= ~FR_PLAN_NO_FUSE;
    if (precision != FR_PREC_BF16 && precision != FR_PREC_F32 && precision != FR_PREC_F24) FR_FAIL(FR_ERR_ARG, "fr_plan_create: bad precision");
    std::unique_ptr<fr_plan> pl(new fr_plan);
    pl->prec = precision; pl->in_c = in_c; pl->in_h = in_h; pl->in_w = in_w;
    FR_HIP_CHECK(hipGetDevice(&pl->device));
    {
        std::vector<uint32_t> z(64, 0u);
        uint32_t* zp = nullptr;
        int rc = upload(pl.get(), z, &zp); if (rc) return rc;
        pl->zeros = zp;
        // ToTensor (b / 255) then Normalize(0.5, 0.5) ((t - 0.5) / 0.5) of inference.py:102-106, in the same
        // fp32 operations as torchvision performs them, tabulated once
        std::vector<float> lut(256);
        for (int b = 0; b < 256; ++b) {
            volatile float t = (float)b / 255.0f;
            volatile float u = t - 0.5f;
            lut[b] = u / 0.5f;
        }
        rc = upload(pl.get(), lut, &pl->u8_lut); if (rc) return rc;
    }
    {
        int rc = aux_streams(pl->device, pl->aux);
        if (rc) return rc;
    }
    for (int i = 0; i < fr_plan::MAX_AUX; ++i)
        FR_HIP_CHECK(hipEventCreateWithFlags(&pl->ev_join[i], hipEventDisableTiming));
    FR_HIP_CHECK(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
    pl->layers.reserve(n);
    for (int i = 0; i < n; ++i) {
        Layer L;
        int rc = build_layer(pl.get(), descs[i], i, L);
        if (rc) return rc;
        pl->layers.push_back(L);
    }
    // ---- buffer slots by liveness ----
    std::vector<int> last_use(n, -1);
    bool need_input_copy = false;
    int input_last_use = -1;
    for (int i = 0; i < n; ++i) {
        const Layer& L = pl->layers[i];
        auto touch = [&](int j) {
            if (j < 0) return;
            last_use[j] = std::max(last_use[j], i);
            if (pl->layers[j].kind == K_SE) last_use[pl->layers[j].src] = std::max(last_use[pl->layers[j].src], i);
        };
        touch(L.src); touch(L.residual);
        if (L.src < 0 && L.kind != K_STEM) { need_input_copy = true; input_last_use = i; }
    }
    last_use[n - 1] = n;   // final output stays live
    // depthwise -> 1x1 fusion (C <= 128): the depthwise output stays in LDS (k_dw.hip, dwp_kernel).
    for (int i = 0; !no_fuse && i + 1 < n; ++i) {
        Layer& D = pl->layers[i];
        Layer& P = pl->layers[i + 1];
        const int af = pl->act_fmt();
        if (!(D.kind == K_DW && D.in_fmt == af && P.kind == K_PW && P.src == i && last_use[i] == i + 1 && P.out_fmt == af)) continue;
        // the expand 1x1 conv in front of the depthwise conv as well (expand -> depthwise -> project in one
        // launch, the 4x wider expanded activation never leaves LDS); FR_FUSE3=0 keeps the expand conv separate
        const bool fuse3 = !(getenv("FR_FUSE3") && atoi(getenv("FR_FUSE3")) == 0);
        bool with_expand = false;
        if (fuse3 && D.src >= 0 && D.src == i - 1) {
            const Layer& E = pl->layers[i - 1];
            const bool se_in = E.src >= 0 && pl->layers[E.src].kind == K_SE;
            with_expand = E.kind == K_PW && E.in_fmt == af && E.out_fmt == af && E.residual < 0 && !se_in && last_use[i - 1] == i &&
                          E.fuse_src < 0 && dwp_supported(af, D.k, D.stride, D.C, D.H, D.W, P.cout, E.K);
        }
        if (!with_expand && !dwp_supported(af, D.k, D.stride, D.C, D.H, D.W, P.cout, 0)) continue;
        // the 3x3 stem conv in front of the depthwise conv as well (stem -> depthwise -> 1x1 in one launch: the stem output, which
        // is also the residual of mobilenetv3_large's first block, is never written or read); FR_FUSE_STEM=0 keeps the stem separate
        const bool fuse_stem = !(getenv("FR_FUSE_STEM") && atoi(getenv("FR_FUSE_STEM")) == 0);
        if (fuse_stem && !with_expand && D.src == i - 1 && i >= 1) {
            Layer& S0 = pl->layers[i - 1];
            const bool res_stem = P.residual == i - 1;
            // measured per shape (tools/dwp_tile_check.sh, 512 images, stem + depthwise + 1x1 together, against stem_kernel + dwp_kernel at its
            // best tile): 32 channels bf16 0.308 vs 0.375 ms, f32 0.486 vs 0.698; 16 channels bf16 0.323 vs 0.270, f32 0.480 vs 0.429 -- the
            // stem phase costs a block the same time whatever the channel count, what it saves grows with the bytes of the stem output:
            // 16-channel plans keep the two launches
            const bool pays = D.C >= 32;
            if (pays && S0.kind == K_STEM && S0.stride == 1 && S0.out_fmt == af && S0.residual < 0 && D.pad == 1 && D.inH == S0.H && D.inW == S0.W &&
                S0.inH == S0.H && S0.inW == S0.W && (P.residual < 0 || res_stem) && last_use[i - 1] <= i + 1 &&
                dwp_supported(af, D.k, D.stride, D.C, D.H, D.W, P.cout, res_stem ? DWP_STEM_RES : DWP_STEM)) {
                P.fuse_stem = i - 1;
                S0.stem_fused_by = i + 1;
            }
        }
        D.fused_next = true;
        P.fuse_src = i;
        if (D.src >= 0) last_use[D.src] = std::max(last_use[D.src], i + 1);
        else if (need_input_copy) input_last_use = std::max(input_last_use, i + 1);
        if (with_expand) {
            Layer& E = pl->layers[i - 1];
            E.fused_next = true;
            P.fuse_src2 = i - 1;
            if (E.src >= 0) last_use[E.src] = std::max(last_use[E.src], i + 1);
            else if (need_input_copy) input_last_use = std::max(input_last_use, i + 1);
        }
    }
    std::vector<size_t>& sizes = pl->slot_bytes;
    std::vector<char> busy;
    if (need_input_copy) pl->input_slot = alloc_slot(sizes, busy, act_bytes_per_image(pl->act_fmt(), in_c, in_h, in_w));
    int launches = need_input_copy ? 1 : 0;
    for (int i = 0; i < n; ++i) {
        Layer& L = pl->layers[i];
        if (L.fused_next) {
            L.slot = -1;
        } else if (L.kind == K_SE) {
            L.slot_pool = alloc_slot(sizes, busy, (size_t)L.cin * 4);
            L.slot_hidden = alloc_slot(sizes, busy, (size_t)L.cout * 4);
            L.slot = alloc_slot(sizes, busy, (size_t)L.cin * 4);
            launches += 3;
        } else {
            L.slot = alloc_slot(sizes, busy, layer_out_bytes(L));
            launches += 1;
            if (L.kind == K_LINEAR && L.splitk > 0) {       // partial sums of the K slices, free again after the layer
                L.slot_partial = alloc_slot(sizes, busy, (size_t)L.splitk * L.cout * 4);
                busy[L.slot_partial] = 0;
                launches += 1;
            }
            // depthwise -> squeeze-excite: the depthwise kernel also emits per-tile channel sums, so the
            // SE block never re-reads the activation (freed again once the SE layer has run)
            static const bool se_fuse = !(getenv("FR_SE_FUSE") && atoi(getenv("FR_SE_FUSE")) == 0);
            if (se_fuse && L.kind == K_DW && L.fuse_src < 0 && i + 1 < n && pl->layers[i + 1].kind == K_SE && pl->layers[i + 1].src == i) {
                const int T = dw_tiles_per_image(L.in_fmt, L.k, L.stride, L.C, L.H, L.W);
                if (T > 0) { L.pool_tiles = T; L.slot_partial = alloc_slot(sizes, busy, (size_t)T * L.C * 4); if (T <= 8) launches -= 1; }   // <= 8 tiles: fc1 folds them, no reduce launch
            }
        }
        if (L.kind == K_SE) {
            busy[L.slot_pool] = 0; busy[L.slot_hidden] = 0;
            if (pl->layers[L.src].slot_partial >= 0) busy[pl->layers[L.src].slot_partial] = 0;
        }
        for (int j = 0; j < i; ++j)
            if (last_use[j] == i && pl->layers[j].slot >= 0) busy[pl->layers[j].slot] = 0;
        if (last_use[i] < 0 && L.slot >= 0) busy[L.slot] = 0;   // dead output (only reachable through taps)
        if (need_input_copy && input_last_use == i) busy[pl->input_slot] = 0;
    }
    for (const Layer& L : pl->layers) if (L.stem_fused_by >= 0) launches -= 1;
    pl->launches = launches + (pl->layers.back().kind == K_LINEAR ? 0 : 1);
    *out_plan = pl.release();
    return FR_OK;
}

extern "C" int fr_plan_destroy(fr_plan* plan) {
    delete plan;
    return FR_OK;
}

extern "C" int fr_plan_workspace_bytes(const fr_plan* plan, int batch, size_t* out_bytes) {
    if (!plan || batch <= 0 || !out_bytes) FR_FAIL(FR_ERR_ARG, "fr_plan_workspace_bytes: bad arguments");
    auto bytes = [&](int b) {
        size_t total = 0;
        for (size_t s : plan->slot_bytes) total += align_up(s * (size_t)b, 256);
        return total + 256;
    };
    size_t worst = bytes(batch);
    for (int parts = 2; parts <= fr_plan::MAX_AUX + 1 && parts <= batch; ++parts) {
        size_t t = 0;
        for (int i = 0; i < parts; ++i) t += align_up(bytes((int)(((long)batch * (i + 1)) / parts - ((long)batch * i) / parts)), 256);
        worst = std::max(worst, t);
    }
    *out_bytes = worst;
    return FR_OK;
}

static size_t plan_ws_exact(const fr_plan* plan, int batch) {
    size_t total = 0;
    for (size_t s : plan->slot_bytes) total += align_up(s * (size_t)batch, 256);
    return total + 256;
}

extern "C" int fr_plan_layer_dims(const fr_plan* plan, int layer, int* c, int* h, int* w) {
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) FR_FAIL(FR_ERR_ARG, "fr_plan_layer_dims: bad layer");
    const Layer& L = plan->layers[layer];
    if (c) *c = L.C; if (h) *h = L.H; if (w) *w = L.W;
    return FR_OK;
}

extern "C" int fr_plan_num_launches(const fr_plan* plan) { return plan ? plan->launches : FR_ERR_ARG; }

// x_u8: `xv` is uint8 NHWC [batch,in_h,in_w,3] (fr_embed_u8) instead of fp32 NCHW.
static int embed_impl(const fr_plan* plan, const void* xv, bool x_u8, int batch, float* out,
                      void* workspace, size_t workspace_bytes, void* stream,
                      int tap_layer, float* tap_out, hipEvent_t* events, int inject_fail_layer = -1) {
    if (!plan || !xv || !out || batch <= 0) FR_FAIL(FR_ERR_ARG, "fr_embed: bad arguments");
    if (x_u8 && (plan->in_c != 3 || plan->input_slot >= 0 || plan->layers.empty() || plan->layers[0].kind != K_STEM))
        FR_FAIL(FR_ERR_UNSUPPORTED, "fr_embed_u8: the plan's first layer is not a 3-channel 3x3 stem");
    const float* x = x_u8 ? nullptr : static_cast<const float*>(xv);
    (void)events;
    const size_t need = plan_ws_exact(plan, batch);
    if (!workspace || workspace_bytes < need) FR_FAIL(FR_ERR_WORKSPACE, "fr_embed: workspace too small");
    if (tap_layer >= (int)plan->layers.size() || (tap_layer >= 0 && !tap_out)) FR_FAIL(FR_ERR_ARG, "fr_embed: bad tap");
    hipStream_t st = (hipStream_t)stream;
    const int N = batch;

    std::vector<unsigned char*> base(plan->slot_bytes.size());
    {
        unsigned char* p = reinterpret_cast<unsigned char*>(align_up((size_t)workspace, 256));
        for (size_t i = 0; i < base.size(); ++i) { base[i] = p; p += align_up(plan->slot_bytes[i] * (size_t)N, 256); }
    }
    const int afmt = plan->act_fmt();
    // f24 buffers: the lo plane of a buffer follows the hi plane of the WHOLE batch
    auto lo_of = [&](const Layer& B) -> unsigned char* {
        return B.out_fmt == FMT_F24 && B.slot >= 0 ? base[B.slot] + f24_plane_bytes((size_t)N * B.H * B.W * B.C) : nullptr;
    };
    const void* in_nhwc = nullptr;
    const unsigned char* in_nhwc_lo = nullptr;
    if (plan->input_slot >= 0) {
        unsigned char* ilo = afmt == FMT_F24 ? base[plan->input_slot] + f24_plane_bytes((size_t)N * plan->in_h * plan->in_w * plan->in_c) : nullptr;
        int rc = launch_nchw_to_nhwc(x, base[plan->input_slot], ilo, afmt, N, plan->in_c, plan->in_h, plan->in_w, st);
        if (rc) return rc;
        in_nhwc = base[plan->input_slot];
        in_nhwc_lo = ilo;
    }
    const int nl = (int)plan->layers.size();
    if (events) FR_HIP_CHECK(hipEventRecord(events[0], st));
    for (int i = 0; i < nl; ++i) {
        const Layer& L = plan->layers[i];
        const bool last = (i == nl - 1);
        if (i == inject_fail_layer) FR_FAIL(FR_ERR_HIP, "fr_embed: injected failure (fr_debug_fail_after)");
        if (L.fused_next) {
            if (i == tap_layer) FR_FAIL(FR_ERR_ARG, "fr_embed_tap: this layer is fused into the next one (create the plan with FR_PLAN_NO_FUSE)");
            if (events) FR_HIP_CHECK(hipEventRecord(events[i + 1], st));
            continue;
        }
        // stem computed inside the kernel of the following depthwise + 1x1 pair (fp32 NCHW input, no tap on the stem)
        auto stem_fused_now = [&](int stem_layer) { return stem_layer >= 0 && !x_u8 && tap_layer != stem_layer; };
        if (L.kind == K_STEM && L.stem_fused_by >= 0 && stem_fused_now(i)) {
            if (events) FR_HIP_CHECK(hipEventRecord(events[i + 1], st));
            continue;
        }
        void* y = (last && L.kind == K_LINEAR) ? (void*)out : (void*)base[L.slot];
        unsigned char* y_lo = lo_of(L);
        const void* xin = nullptr;
        const unsigned char* xin_lo = nullptr;
        const float* se = nullptr;
        if (L.src < 0) { xin = in_nhwc; xin_lo = in_nhwc_lo; }
        else if (plan->layers[L.src].kind == K_SE) {
            const Layer& S = plan->layers[L.src];
            xin = base[plan->layers[S.src].slot];
            xin_lo = lo_of(plan->layers[S.src]);
            se = reinterpret_cast<const float*>(base[S.slot]);
        } else { xin = base[plan->layers[L.src].slot]; xin_lo = lo_of(plan->layers[L.src]); }
        const void* res = L.residual >= 0 ? base[plan->layers[L.residual].slot] : nullptr;
        const unsigned char* res_lo = L.residual >= 0 ? lo_of(plan->layers[L.residual]) : nullptr;
        int rc = FR_OK;
        switch (L.kind) {
            case K_STEM: {
                StemArgs a{};
                a.x = x; a.w = L.w_packed;
                a.e = EpiArgs{L.bias, L.slope, res, y, L.cout, L.act, y_lo, res_lo};
                a.N = N; a.H = L.inH; a.W = L.inW; a.Ho = L.H; a.Wo = L.W; a.stride = L.stride; a.NT = L.NT;
                a.zeros = plan->zeros;
                if (x_u8) { a.x8 = static_cast<const unsigned char*>(xv); a.lut = plan->u8_lut; }
                rc = launch_stem(L.out_fmt, a, st);
                break;
            }
            case K_CONV3: {
                ConvArgs a{};
                a.x = xin; a.w = L.w_packed;
                a.e = EpiArgs{L.bias, L.slope, res, y, L.cout, L.act};
                a.M = (long)N * L.H * L.W; a.H = L.inH; a.W = L.inW; a.Ho = L.H; a.Wo = L.W;
                a.Cin = L.cin; a.ksize = 3; a.stride = L.stride; a.pad = L.pad; a.zeros = plan->zeros; a.bk = L.conv_bk; a.halo = L.conv_halo16 ? 2 : (L.conv_halo ? 1 : 0);
                a.spb = L.conv_spb;
                rc = launch_conv(L.in_f32, L.out_f32, a, st);
                break;
            }
            case K_PW: case K_LINEAR: {
                if (L.kind == K_PW && L.fuse_src >= 0) {            // depthwise + this 1x1 conv in one kernel
                    const Layer& D = plan->layers[L.fuse_src];
                    DwArgs a{};
                    a.x = L.fuse_src2 >= 0 ? nullptr : (D.src < 0 ? in_nhwc : (const void*)base[plan->layers[D.src].slot]);
                    a.x_lo = L.fuse_src2 >= 0 ? nullptr : (D.src < 0 ? in_nhwc_lo : (const unsigned char*)lo_of(plan->layers[D.src]));
                    a.w = D.w_f32; a.bias = D.bias; a.slope = D.slope; a.y = nullptr;
                    a.N = N; a.H = D.inH; a.W = D.inW; a.C = D.C; a.Ho = D.H; a.Wo = D.W; a.pad = D.pad; a.act = D.act;
                    a.zeros = plan->zeros;
                    DwpExpand ex{};
                    if (L.fuse_src2 >= 0) {
                        const Layer& E = plan->layers[L.fuse_src2];
                        ex.xin = E.src < 0 ? in_nhwc : (const void*)base[plan->layers[E.src].slot];
                        ex.xin_lo = E.src < 0 ? in_nhwc_lo : (const unsigned char*)lo_of(plan->layers[E.src]);
                        ex.we = E.w_packed; ex.bias = E.bias; ex.slope = E.slope; ex.act = E.act; ex.Cin = E.K;
                    }
                    const bool with_stem = stem_fused_now(L.fuse_stem);
                    if (with_stem) {
                        const Layer& S0 = plan->layers[L.fuse_stem];
                        ex.stem = 1; ex.res_lds = L.residual == L.fuse_stem; ex.xs = x;
                        ex.we = S0.w_packed; ex.bias = S0.bias; ex.slope = S0.slope; ex.act = S0.act; ex.Cin = 3;
                        a.x = nullptr; a.x_lo = nullptr;
                        if (ex.res_lds) { res = nullptr; res_lo = nullptr; }
                    }
                    rc = launch_dwp(afmt, D.k, D.stride, a, L.w_packed, EpiArgs{L.bias, L.slope, res, y, L.cout, L.act, y_lo, res_lo},
                                    (L.fuse_src2 >= 0 || with_stem) ? &ex : nullptr, st);
                    break;
                }
                if (L.via_conv) {
                    ConvArgs a{};
                    a.x = xin; a.w = L.w_packed;
                    a.e = EpiArgs{L.bias, nullptr, nullptr, y, L.cout, L.act};
                    a.M = N; a.H = a.W = a.Ho = a.Wo = 1; a.Cin = L.K; a.ksize = 1; a.stride = 1; a.pad = 0; a.zeros = plan->zeros; a.bk = L.conv_bk;
                    rc = launch_conv(L.in_f32, true, a, st);
                    break;
                }
                PwArgs a{};
                a.x = xin; a.x_lo = xin_lo; a.w = L.w_packed; a.se = se;
                a.e = EpiArgs{L.bias, L.slope, res, y, L.cout, L.act, y_lo, res_lo};
                a.M = L.kind == K_PW ? (long)N * L.H * L.W : (long)N;
                a.K = L.K; a.KS = L.KS; a.NT = L.NT; a.HW = L.H * L.W; a.chunk_ks = L.chunk_ks;
                if (L.kind == K_LINEAR && L.splitk > 0) rc = launch_fc_splitk(L.in_f32, a, reinterpret_cast<float*>(base[L.slot_partial]), L.splitk, st);
                else rc = L.kind == K_LINEAR ? launch_fc(L.in_f32, a, st) : launch_pw(L.in_fmt, L.out_fmt, a, st);
                break;
            }
            case K_DW: {
                DwArgs a{};
                a.x = xin; a.x_lo = xin_lo; a.y_lo = y_lo; a.w = L.w_f32; a.bias = L.bias; a.slope = L.slope; a.y = y;
                a.N = N; a.H = L.inH; a.W = L.inW; a.C = L.C; a.Ho = L.H; a.Wo = L.W; a.pad = L.pad; a.act = L.act;
                a.zeros = plan->zeros;
                if (L.slot_partial >= 0) { a.pool = reinterpret_cast<float*>(base[L.slot_partial]); a.pool_tiles = L.pool_tiles; }
                rc = launch_dw(L.in_fmt, L.k, L.stride, a, st);
                break;
            }
            case K_GDW: {
                ReduceArgs a{xin, xin_lo, L.w_f32, L.bias, reinterpret_cast<float*>(y), N, L.inH * L.inW, L.C};
                rc = launch_reduce(L.in_fmt, a, st);
                break;
            }
            case K_SE: {
                float* pool = reinterpret_cast<float*>(base[L.slot_pool]);
                float* hid = reinterpret_cast<float*>(base[L.slot_hidden]);
                const Layer& D = plan->layers[L.src];
                // pooled sums came with the depthwise output: fc1 folds up to 8 tiles per image itself (one launch
                // fewer); more tiles than that are folded by a reduce launch first (a serial fold inside the GEMM's
                // operand load was slower there: 0.030 vs 0.015 ms for the 28x28 layer with 49 tiles)
                const bool partials = D.slot_partial >= 0 && D.pool_tiles <= 8;
                if (D.slot_partial >= 0 && !partials) {
                    ReduceArgs r{base[D.slot_partial], nullptr, nullptr, nullptr, pool, N, D.pool_tiles, L.C, 1.f / (float)(L.H * L.W)};
                    rc = launch_reduce(FMT_F32, r, st);
                } else if (!partials) {
                    ReduceArgs r{xin, xin_lo, nullptr, nullptr, pool, N, L.H * L.W, L.C, 0.f};
                    rc = launch_reduce(L.in_fmt, r, st);
                }
                if (rc) break;
                PwArgs a{};
                a.x = pool; a.w = L.w_packed; a.se = nullptr;
                if (partials) { a.x = base[D.slot_partial]; a.x_tiles = D.pool_tiles; a.x_scale = 1.f / (float)(L.H * L.W); }
                a.e = EpiArgs{L.bias, nullptr, nullptr, hid, L.cout, FR_ACT_RELU};
                a.M = N; a.K = L.K; a.KS = L.KS; a.NT = L.NT; a.HW = 1; a.chunk_ks = L.chunk_ks;
                rc = launch_fc(true, a, st);
                if (rc) break;
                PwArgs b{};
                b.x = hid; b.w = L.w_packed2; b.se = nullptr;
                b.e = EpiArgs{L.bias2, nullptr, nullptr, y, L.cin, FR_ACT_HSIGMOID};
                b.M = N; b.K = L.K2; b.KS = L.KS2; b.NT = L.NT2; b.HW = 1; b.chunk_ks = L.chunk_ks2;
                rc = launch_fc(true, b, st);
                break;
            }
            default: FR_FAIL(FR_ERR_UNSUPPORTED, "fr_embed: kernel kind not built");
        }
        if (rc) return rc;
        if (i == tap_layer) {
            if (L.kind == K_SE) {           // tap of an SE block = its per-image scale [N,C]
                FR_HIP_CHECK(hipMemcpyAsync(tap_out, y, (size_t)N * L.C * 4, hipMemcpyDeviceToDevice, st));
            } else if (L.kind == K_LINEAR || L.kind == K_GDW) {
                if (y != (void*)tap_out) FR_HIP_CHECK(hipMemcpyAsync(tap_out, y, (size_t)N * L.C * 4, hipMemcpyDeviceToDevice, st));
            } else {
                rc = launch_nhwc_to_nchw(y, y_lo, L.out_fmt, tap_out, N, L.C, L.H, L.W, st);
                if (rc) return rc;
            }
        }
        if (last && L.kind != K_LINEAR) {
            if (L.kind == K_GDW || L.kind == K_SE) FR_HIP_CHECK(hipMemcpyAsync(out, y, (size_t)N * L.C * 4, hipMemcpyDeviceToDevice, st));
            else { rc = launch_nhwc_to_nchw(y, y_lo, L.out_fmt, out, N, L.C, L.H, L.W, st); if (rc) return rc; }
        }
        if (events) FR_HIP_CHECK(hipEventRecord(events[i + 1], st));
    }
    return FR_OK;
}

extern "C" int fr_embed_tap(const fr_plan* plan, const float* x, int batch, float* out,
                            void* workspace, size_t workspace_bytes, void* stream,
                            int tap_layer, float* tap_out) {
    return embed_impl(plan, x, false, batch, out, workspace, workspace_bytes, stream, tap_layer, tap_out, nullptr);
}

extern "C" int fr_embed_profile(const fr_plan* plan, const float* x, int batch, float* out,
                                void* workspace, size_t workspace_bytes, void* stream, float* layer_ms) {
    if (!plan || !layer_ms) FR_FAIL(FR_ERR_ARG, "fr_embed_profile: bad arguments");
    const int nl = (int)plan->layers.size();
    std::vector<hipEvent_t> ev(nl + 1, nullptr);
    int rc = FR_OK;
    for (auto& e : ev) {                                  // a failed create must not leak the events made before it
        hipError_t he = hipEventCreate(&e);
        if (he != hipSuccess) { e = nullptr; fr_set_error(std::string("hipEventCreate: ") + hipGetErrorString(he)); rc = FR_ERR_HIP; break; }
    }
    if (rc == FR_OK) rc = embed_impl(plan, x, false, batch, out, workspace, workspace_bytes, stream, -1, nullptr, ev.data());
    if (rc == FR_OK) {
        hipError_t he = hipEventSynchronize(ev[nl]);
        if (he != hipSuccess) { fr_set_error(std::string("hipEventSynchronize: ") + hipGetErrorString(he)); rc = FR_ERR_HIP; }
        for (int i = 0; i < nl && rc == FR_OK; ++i)
            if (hipEventElapsedTime(&layer_ms[i], ev[i], ev[i + 1]) != hipSuccess) { fr_set_error("hipEventElapsedTime failed"); rc = FR_ERR_HIP; }
    }
    const std::string kept = rc ? g_last_error : std::string();
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (rc) fr_set_error(kept);
    return rc;
}

extern "C" int fr_plan_layer_kernel(const fr_plan* plan, int layer, const char** name) {
    if (!plan || layer < 0 || layer >= (int)plan->layers.size() || !name) FR_FAIL(FR_ERR_ARG, "fr_plan_layer_kernel: bad layer");
    static const char* names[] = {"stem_kernel", "pw_gemm_kernel", "dw_kernel", "reduce_kernel", "se(reduce_kernel+2xfc_kernel)",
                                  "fc_kernel", "conv_glds_kernel"};
    const Layer& L = plan->layers[layer];
    *name = names[L.kind];
    if (L.kind == K_SE && L.src >= 0 && plan->layers[L.src].slot_partial >= 0 && plan->layers[L.src].pool_tiles <= 8)
        *name = "se(2xfc_kernel)";                    // fc1 folds the depthwise kernel's pooling partials itself
    if (L.fused_next || L.stem_fused_by >= 0) {
        *name = "(fused into the next layer)";
    } else if (L.kind == K_PW && L.fuse_src >= 0) {
        *name = "dwp_kernel";
        if (L.fuse_src2 >= 0) {
            const Layer& D = plan->layers[L.fuse_src];
            if (dwp_is_chunked(plan->act_fmt(), D.k, D.stride, D.C, D.H, D.W, L.cout, plan->layers[L.fuse_src2].K)) *name = "dwc_kernel";
        }
    } else if (L.kind == K_PW) {
        const bool se = L.src >= 0 && plan->layers[L.src].kind == K_SE;
        *name = pw_kernel_name(L.in_fmt, L.out_fmt, L.K, L.cout, se, L.H * L.W);
    } else if (L.kind == K_CONV3 && L.in_f32) {
        *name = "conv_gemm_kernel";
    } else if (L.kind == K_CONV3 && L.conv_halo) {
        *name = L.conv_halo16 ? "conv_halo16_kernel" : "conv_halo_kernel";
    } else if (L.kind == K_LINEAR && L.via_conv) {
        *name = L.in_f32 ? "conv_gemm_kernel" : "conv_glds_kernel";
    } else if (L.kind == K_LINEAR && L.splitk > 0) {
        *name = "fc_splitk_kernel(+reduce)";
    }
    return FR_OK;
}

static std::atomic<int> g_fail_after{-1};
extern "C" int fr_debug_fail_after(int layer) { g_fail_after.store(layer < 0 ? -1 : layer); return FR_OK; }

static int embed_split(const fr_plan* plan, const void* xv, bool x_u8, int batch, float* out,
                       void* workspace, size_t workspace_bytes, void* stream) {
    const unsigned char* x = static_cast<const unsigned char*>(xv);
    // Large batches are cut into `parts` independent sub-batches, the first on the caller's stream and
    // the others on the plan's auxiliary streams (fork/join by events, still asynchronous): kernel
    // tails, launch gaps and the last partially filled wave of workgroups of one part are filled by
    // the others.  Measured: mobilenetv3_large batch 1024 +4 %, sphere64 batch 512 +11 % with 2 parts.
    static const int split_min = getenv("FR_SPLIT_MIN") ? atoi(getenv("FR_SPLIT_MIN")) : 128;
    static const int split_parts = getenv("FR_SPLIT_PARTS") ? atoi(getenv("FR_SPLIT_PARTS")) : 2;
    const int parts = std::min(split_parts, fr_plan::MAX_AUX + 1);
    if (plan && x && out && workspace && split_min > 0 && parts > 1 && batch >= split_min && batch >= parts) {
        size_t off[fr_plan::MAX_AUX + 2] = {0};
        int n0[fr_plan::MAX_AUX + 2] = {0};
        for (int i = 0; i < parts; ++i) {
            n0[i + 1] = (int)(((long)batch * (i + 1)) / parts);
            off[i + 1] = off[i] + align_up(plan_ws_exact(plan, n0[i + 1] - n0[i]), 256);
        }
        if (workspace_bytes >= off[parts]) {
            hipStream_t st = (hipStream_t)stream;
            const Layer& last = plan->layers.back();
            const size_t in_el = (size_t)plan->in_c * plan->in_h * plan->in_w * (x_u8 ? 1 : sizeof(float));   // bytes per image
            const size_t out_el = (size_t)last.C * last.H * last.W;
            FR_HIP_CHECK(hipEventRecord(plan->ev_fork, st));
            // Every exit path joins what it forked: a part that fails half way has already put kernels on its
            // auxiliary stream that write the caller's workspace and output, so `st` must wait for them before the
            // caller (who sees the error) frees or reuses those buffers.  The first error is the one reported.
            int rc = FR_OK;
            std::string first_error;
            auto note = [&](int r) { if (r && !rc) { rc = r; first_error = g_last_error; } };
            auto hip_rc = [&](hipError_t e, const char* what) {
                if (e == hipSuccess) return (int)FR_OK;
                fr_set_error(std::string(what) + ": " + hipGetErrorString(e));
                return (int)FR_ERR_HIP;
            };
            // test hook (tests/test_gpu_parity.py::test_split_failure_joins_auxiliary_streams): part 1 stops with an
            // error after this many layers have been enqueued on its auxiliary stream
            // (set through fr_debug_fail_after, not the environment: no getenv on the hot path, and nothing a stray
            // environment variable of a production server can switch on)
            const int inj = g_fail_after.load(std::memory_order_relaxed);
            for (int i = 0; i < parts && !rc; ++i) {
                hipStream_t si = i == 0 ? st : plan->aux[i - 1];
                if (i > 0) {
                    note(hip_rc(hipStreamWaitEvent(si, plan->ev_fork, 0), "hipStreamWaitEvent(fork)"));
                    if (rc) break;                       // nothing was put on this auxiliary stream
                }
                note(embed_impl(plan, x + (size_t)n0[i] * in_el, x_u8, n0[i + 1] - n0[i], out + (size_t)n0[i] * out_el,
                                reinterpret_cast<unsigned char*>(workspace) + off[i], off[i + 1] - off[i], si, -1, nullptr, nullptr,
                                (inj >= 0 && i == 1) ? inj : -1));
                if (i > 0) {
                    const int j = hip_rc(hipEventRecord(plan->ev_join[i - 1], si), "hipEventRecord(join)");
                    note(j);
                    if (j == FR_OK) note(hip_rc(hipStreamWaitEvent(st, plan->ev_join[i - 1], 0), "hipStreamWaitEvent(join)"));
                    else (void)hipStreamSynchronize(si);  // could not record the join: wait on the host rather than leave it running
                }
            }
            if (rc) fr_set_error(first_error);
            return rc;
        }
    }
    return embed_impl(plan, xv, x_u8, batch, out, workspace, workspace_bytes, stream, -1, nullptr, nullptr);
}

// ---- hipGraph replay of one plan at one batch size (small batches: a forward is ~60 launches, launch-latency bound) ----
struct fr_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int device = 0, nodes = 0;
};

extern "C" int fr_embed_graph_destroy(fr_graph* g) {
    if (!g) return FR_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return FR_OK;
}

extern "C" int fr_embed_graph_create(const fr_plan* plan, const void* x, int x_is_u8, int batch, float* out,
                                     void* workspace, size_t workspace_bytes, fr_graph** out_graph) {
    if (!plan || !x || !out || !workspace || !out_graph || batch <= 0) FR_FAIL(FR_ERR_ARG, "fr_embed_graph_create: bad arguments");
    int dev = 0;
    FR_HIP_CHECK(hipGetDevice(&dev));
    if (dev != plan->device) FR_FAIL(FR_ERR_ARG, "fr_embed_graph_create: the plan belongs to another device");
    hipStream_t cs = nullptr;
    FR_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    std::unique_ptr<fr_graph> g(new fr_graph);
    g->device = dev;
    // eager once: code objects loaded, hipFuncSetAttribute done, errors of the shape reported outside a capture
    int rc = embed_impl(plan, x, x_is_u8 != 0, batch, out, workspace, workspace_bytes, cs, -1, nullptr, nullptr);
    hipError_t he = hipStreamSynchronize(cs);
    if (rc == FR_OK && he != hipSuccess) { fr_set_error(std::string("fr_embed_graph_create: ") + hipGetErrorString(he)); rc = FR_ERR_HIP; }
    if (rc == FR_OK) {
        he = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
        if (he != hipSuccess) { fr_set_error(std::string("hipStreamBeginCapture: ") + hipGetErrorString(he)); rc = FR_ERR_HIP; }
        else {
            rc = embed_impl(plan, x, x_is_u8 != 0, batch, out, workspace, workspace_bytes, cs, -1, nullptr, nullptr);
            const std::string first = rc ? g_last_error : std::string();
            he = hipStreamEndCapture(cs, &g->graph);            // always: the stream must leave capture mode
            if (rc) fr_set_error(first);
            else if (he != hipSuccess || !g->graph) { fr_set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(he)); rc = FR_ERR_HIP; }
        }
    }
    if (rc == FR_OK) {
        he = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (he != hipSuccess) { fr_set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(he)); rc = FR_ERR_HIP; }
    }
    if (rc == FR_OK) {
        size_t n = 0;                                       // kernel nodes only (memcpy / memset nodes of taps are not launches)
        if (hipGraphGetNodes(g->graph, nullptr, &n) == hipSuccess && n) {
            std::vector<hipGraphNode_t> nodes(n);
            if (hipGraphGetNodes(g->graph, nodes.data(), &n) == hipSuccess)
                for (size_t i = 0; i < n; ++i) {
                    hipGraphNodeType t;
                    if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++g->nodes;
                }
        }
    }
    (void)hipStreamDestroy(cs);
    if (rc) { fr_embed_graph_destroy(g.release()); return rc; }
    *out_graph = g.release();
    return FR_OK;
}

extern "C" int fr_embed_graph_launch(fr_graph* g, void* stream) {
    if (!g || !g->exec) FR_FAIL(FR_ERR_ARG, "fr_embed_graph_launch: no graph");
    int dev = -1;
    FR_HIP_CHECK(hipGetDevice(&dev));
    if (dev != g->device) FR_FAIL(FR_ERR_ARG, "fr_embed_graph_launch: the graph belongs to another device than the current one");
    FR_HIP_CHECK(hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
    return FR_OK;
}

extern "C" int fr_embed_graph_nodes(const fr_graph* g) { return g ? g->nodes : FR_ERR_ARG; }

extern "C" int fr_embed(const fr_plan* plan, const float* x, int batch, float* out,
                        void* workspace, size_t workspace_bytes, void* stream) {
    return embed_split(plan, x, false, batch, out, workspace, workspace_bytes, stream);
}

extern "C" int fr_embed_u8(const fr_plan* plan, const unsigned char* x_nhwc, int batch, float* out,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return embed_split(plan, x_nhwc, true, batch, out, workspace, workspace_bytes, stream);
}
