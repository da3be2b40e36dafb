// Torch bindings of the gfx950 kernels: validate operands on the host (a faulting kernel can
// take the whole node down, so every shape / dtype / layout / size assumption of a kernel is
// checked here), allocate outputs through the caching allocator, launch on the current stream.
#include <algorithm>
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "../kernels/kernels.h"
#include <cstdlib>
#include <string>

#include "binding.h"

namespace pdt {
namespace {

using at::Tensor;

hipStream_t cur_stream(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

// The convolution / pooling kernels address one operand through a 32-bit buffer descriptor (range-checked loads are
// what gives them zero padding and ragged tiles for free), so ONE LAUNCH sees at most 2 GiB of each operand.  A
// batch whose activations are larger (56x56x256 bf16 is 1.6 MB per image: 2 GiB at B = 1337) is cut into batch
// chunks HERE: NHWC keeps a sub-range of images contiguous, every kernel is independent per image, and the
// per-launch partial rows (BatchNorm statistics, split-K slabs) of the chunks are simply stacked.  The BatchNorm,
// GEMM and elementwise kernels use 64-bit addressing and take the whole tensor.
long g_window_bytes = (1L << 31) - 4096;
void set_activation_window(int64_t bytes) {
    TORCH_CHECK(bytes >= 4096 && bytes < (1L << 31), "activation window: 4 KiB .. 2 GiB");
    g_window_bytes = bytes;
}
// images per launch such that the largest per-image operand (bytes) stays inside the window
long chunk_images(long B, long bytes_per_image) {
    TORCH_CHECK(bytes_per_image <= g_window_bytes, "one image of ", bytes_per_image,
                " bytes exceeds the buffer-descriptor window of ", g_window_bytes, " bytes");
    const long fit = std::max<long>(1, g_window_bytes / std::max<long>(bytes_per_image, 1));
    if (fit >= B) return std::max<long>(1, B);
    // equal chunks: 2048 images of a layer that fits 1337 per window run as 1024 + 1024, not 1337 + 711 (a short
    // last chunk pays the per-launch ramp and tail of a full one)
    const long chunks = (B + fit - 1) / fit;
    return (B + chunks - 1) / chunks;
}
inline const char* at_image(const void* p, long b0, long bytes_per_image) {
    return p ? static_cast<const char*>(p) + b0 * bytes_per_image : nullptr;
}
inline char* at_image(void* p, long b0, long bytes_per_image) {
    return p ? static_cast<char*>(p) + b0 * bytes_per_image : nullptr;
}

void check_nhwc_bf16(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, ": expected a GPU tensor");
    TORCH_CHECK(t.scalar_type() == at::kBFloat16, name, ": expected bf16, got ", t.scalar_type());
    TORCH_CHECK(t.dim() == 4, name, ": expected 4-D [B,C,H,W]");
    TORCH_CHECK(t.is_contiguous(at::MemoryFormat::ChannelsLast), name, ": expected channels_last (NHWC) memory");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, ": data pointer must be 16-B aligned");
}
void check_vec_f32(const Tensor& t, long n, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.numel() == n, name,
                ": expected contiguous fp32 GPU tensor with ", n, " elements");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, ": data pointer must be 16-B aligned");
}
bool bn_channels_ok(long C) { return C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0; }

Tensor empty_nhwc(long B, long C, long H, long W, const Tensor& like) {
    return at::empty({B, C, H, W}, like.options().memory_format(at::MemoryFormat::ChannelsLast));
}

Tensor reduce_slab(const Tensor& slab, int nparts, long n) {
    Tensor out = at::empty({n}, slab.options());
    Tensor tmp;
    float* tmp_ptr = nullptr;
    // two levels when the column blocks alone (256 floats each) do not fill the chip
    if (nparts > 64 && n < 512 * 256) {
        tmp = at::empty({kReduceChunks, n}, slab.options());
        tmp_ptr = tmp.data_ptr<float>();
    }
    launch_reduce_slab(slab.data_ptr<float>(), nparts, n, (int)n, out.data_ptr<float>(), tmp_ptr, cur_stream(slab));
    return out;
}

constexpr int kMaxFinalizeRows = 512;  // rows the consumer kernels (bn_finalize) sum themselves

// [nparts][n] -> [<= max_rows][n]: first reduction level only (the consumer kernel fuses the second)
Tensor shrink_rows(const Tensor& slab, int nparts, long n, int max_rows) {
    if (nparts <= max_rows) return slab;
    const float* src = slab.data_ptr<float>();
    Tensor mid;
    // narrow rows, many of them (BN statistics of a 56x56 layer at batch 1024: 25 088 rows of 128 floats): 32
    // chunk blocks would read the 12 MB alone, so a first pass with ~1024 blocks brings the row count down
    const int colblocks = (int)((n + 255) / 256);
    if (nparts >= 2048 && colblocks * kReduceChunks < 512) {
        const int chunks = std::min(1024 / colblocks, nparts / 8);
        mid = at::empty({chunks, n}, slab.options());
        nparts = launch_reduce_rows_l1(src, nparts, n, (int)n, mid.data_ptr<float>(), chunks, cur_stream(slab));
        src = mid.data_ptr<float>();
    }
    Tensor out = at::empty({kReduceChunks, n}, slab.options());
    const int rows = launch_reduce_rows_l1(src, nparts, n, (int)n, out.data_ptr<float>(), kReduceChunks,
                                           cur_stream(slab));
    return out.narrow(0, 0, rows);
}

// ------------------------------------------------------------------------------------- conv
void check_conv_weight(const Tensor& w, const char* name) {
    TORCH_CHECK(w.is_cuda() && w.scalar_type() == at::kBFloat16 && w.dim() == 4, name, ": expected 4-D bf16 GPU weight");
    TORCH_CHECK(w.is_contiguous(at::MemoryFormat::ChannelsLast), name,
                ": expected channels_last weight (physical [Cout][kh][kw][Cin])");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(w.data_ptr()) % 16 == 0, name, ": data pointer must be 16-B aligned");
}

// cfg selecting the GEMM core (gemm.hip) instead of an implicit-GEMM build; 0 = pick by shape
constexpr int64_t kCfgGemmCore = 100;
// cfg selecting the 64-channel 3x3 window kernel (conv_win64.hip); 0 picks it whenever the layer fits
constexpr int64_t kCfgWin64 = 101;

// A pointwise stride-1 convolution in NHWC is C[M,N] = A[M,K] . W[N,K]^T with dense rows.  Measured at B=1024 on
// every pointwise shape of ResNet-50 (bench/micro/gemm_conv_shapes.py): with Cout a multiple of the 256-column
// tile the GEMM core is 0-35 % faster than the implicit-GEMM kernel (64->256 @56: 388 vs 572 us with the
// statistics epilogue), with 64 / 128 output channels it loses (half-empty tiles).  PDT_CONV_GEMM=0 turns it off.
// the 64-channel window kernel is persistent too (one workgroup per CU, tiles of 4 image rows): two tiles per CU at least
bool win64_enough(long B, long H) { return B * ((H + 3) / 4) >= 512; }

// The persistent 256x256 kernels want the chip full: below ~3/4 of the CUs' worth of tiles (the CIFAR-shaped step:
// 32 tiles per layer) the small-tile implicit-GEMM kernel wins (measured on the 3x32x32 hipGraph step: 4.10 ms with
// the round-1 kernels, 5.22 ms with the new ones on every layer, 4.1 again with this rule).
bool enough_gemm_tiles(long M, long N) { return ((M + 255) / 256) * ((N + 255) / 256) >= 192; }

// Output widths that are multiples of 128 but not of 256 on the GEMM core's 128-wide tile build (PDT_CONV_GEMM_N128=1)
bool conv_gemm_n128() {
    static const bool on = [] {
        const char* e = std::getenv("PDT_CONV_GEMM_N128");
        return e && std::string(e) == "1";
    }();
    return on;
}

bool conv_gemm_core_ok(long C, long N, long KH, long KW, long stride, long pad, int64_t cfg) {
    static const bool enabled = [] {
        const char* e = std::getenv("PDT_CONV_GEMM");
        return !(e && std::string(e) == "0");
    }();
    const bool eligible = KH == 1 && KW == 1 && stride == 1 && pad == 0 && C % 64 == 0 && N % 8 == 0;
    if (cfg == kCfgGemmCore) {
        TORCH_CHECK(eligible, "the GEMM core takes pointwise stride-1 convolutions with Cin % 64 == 0");
        return true;
    }
    return cfg == 0 && enabled && eligible && (N % 256 == 0 || (conv_gemm_n128() && N % 128 == 0));  // (+ enough_gemm_tiles at the call sites, which know M)
}

// 3x3 / stride 1 / pad 1 on the GEMM core's spatial-filter mode (the taps go into the DMA's scalar offset, the zero
// padding into per-row validity bits).  PDT_CONV_GEMM3=0 turns it off.
bool conv_gemm_taps_ok(long B, long C, long N, long H, long W, long KH, long KW, long stride, long pad, int64_t cfg) {
    static const bool enabled = [] {
        const char* e = std::getenv("PDT_CONV_GEMM3");
        return !(e && std::string(e) == "0");
    }();
    const bool eligible = KH == 3 && KW == 3 && stride == 1 && pad == 1 && C % 64 == 0 && N % 8 == 0 && W <= 126 &&
                          B * H * W < (1L << 31);
    if (cfg == kCfgGemmCore && eligible) return true;
    if (cfg == 0 && enabled && eligible && conv_gemm_n128() && N % 128 == 0 && C >= 128 && enough_gemm_tiles(B * H * W, 256)) return true;
    return cfg == 0 && enabled && eligible && N % 256 == 0 && C >= 256 && enough_gemm_tiles(B * H * W, N);
}
// forward taps (dh, dw) = (r - pad, q - pad) with weight slot r * KW + q; data gradient: (pad - r, pad - q)
void set_gemm_taps(GemmArgs& ga, long H, long W, long KH, long KW, long pad, bool dgrad) {
    ga.ntaps = (int)(KH * KW);
    ga.H = (int)H;
    ga.W = (int)W;
    for (int r = 0; r < KH; ++r)
        for (int q = 0; q < KW; ++q) {
            const int t = r * (int)KW + q;
            const int dh = dgrad ? (int)pad - r : r - (int)pad, dw = dgrad ? (int)pad - q : q - (int)pad;
            ga.tap[t] = (dh & 0xff) | ((dw & 0xff) << 8) | (t << 16);
        }
}

std::tuple<Tensor, Tensor> conv2d_fwd(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, bool want_stats,
                                      int64_t cfg) {
    check_nhwc_bf16(x, "x");
    check_conv_weight(w, "w");
    const long B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    const long N = w.size(0), KH = w.size(2), KW = w.size(3);
    TORCH_CHECK(w.size(1) == C, "weight Cin ", w.size(1), " != input C ", C);
    TORCH_CHECK(C % 64 == 0 && N % 64 == 0, "conv2d_fwd: Cin and Cout must be multiples of 64 (got ", C, ", ", N, ")");
    TORCH_CHECK(KH * KW <= kMaxTaps && stride >= 1 && pad >= 0 && pad < 64, "unsupported kernel geometry");
    const long Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    TORCH_CHECK(Ho > 0 && Wo > 0, "empty output");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor y = empty_nhwc(B, N, Ho, Wo, x);
    if ((cfg == kCfgWin64 || (cfg == 0 && win64_enough(B, H))) &&
        conv_win64_ok((int)C, (int)N, (int)KH, (int)KW, (int)stride, (int)pad, (int)H, (int)W)) {
        const int grid = conv_win64_grid((int)B, (int)H);
        Tensor slab, stats;
        if (want_stats) slab = at::empty({grid, 2, N}, x.options().dtype(at::kFloat));
        launch_conv_win64(x.data_ptr(), w.data_ptr(), y.data_ptr(), want_stats ? slab.data_ptr<float>() : nullptr, (int)B,
                          (int)H, (int)W, 0, nullptr, nullptr, nullptr, nullptr, cur_stream(x));
        return {y, slab};
    }
    TORCH_CHECK(cfg != kCfgWin64, "the window kernel takes 3x3 / stride 1 / pad 1 convolutions with 64 -> 64 channels");
    const bool taps = conv_gemm_taps_ok(B, C, N, H, W, KH, KW, stride, pad, cfg);
    if (taps || (conv_gemm_core_ok(C, N, KH, KW, stride, pad, cfg) && (cfg == kCfgGemmCore || enough_gemm_tiles(B * H * W, N)))) {
        GemmArgs ga;
        std::memset(&ga, 0, sizeof(ga));
        ga.A = x.data_ptr(); ga.B = w.data_ptr(); ga.C = y.data_ptr();
        ga.M = B * H * W; ga.N = (int)N; ga.K = (int)C;
        ga.lda = (int)C; ga.ldb = (int)(C * KH * KW); ga.ldc = (int)N;
        if (taps) set_gemm_taps(ga, H, W, KH, KW, pad, false);
        Tensor slab, stats;
        const int m_tiles = gemm_m_tiles(ga.M);
        if (want_stats) {
            slab = at::empty({m_tiles, 2 * N}, x.options().dtype(at::kFloat));
            ga.conv_mode = 1;
            ga.stats_slab = slab.data_ptr<float>();
        }
        launch_gemm_nt(ga, cur_stream(x));
        if (want_stats) stats = shrink_rows(slab, m_tiles, 2 * N, kMaxFinalizeRows).view({-1, 2, N});
        return {y, stats};
    }
    ConvGeom g;
    std::memset(&g, 0, sizeof(g));
    g.B = B; g.Hin = H; g.Win = W; g.C = C; g.Hout = Ho; g.Wout = Wo; g.N = N;
    g.nOH = Ho; g.nOW = Wo; g.os = 1; g.ooh = 0; g.oow = 0; g.is = stride;
    g.ntaps = KH * KW; g.wtaps = KH * KW;
    for (int r = 0; r < KH; ++r)
        for (int s = 0; s < KW; ++s) {
            const int t = r * KW + s;
            g.set_tap(t, r - (int)pad, s - (int)pad, t);
        }
    const long x_img = H * W * C * 2, y_img = Ho * Wo * N * 2;
    const long per = chunk_images(B, std::max(x_img, y_img));
    std::vector<int> tiles;
    int m_tiles = 0;
    for (long b0 = 0; b0 < B; b0 += per) {
        g.B = (int)std::min(per, B - b0);
        tiles.push_back(conv_igemm_m_tiles(g, nullptr, (int)cfg));
        m_tiles += tiles.back();
    }
    Tensor stats, slab;
    float* slab_ptr = nullptr;
    if (want_stats) {
        slab = at::empty({m_tiles, 2 * N}, x.options().dtype(at::kFloat));
        slab_ptr = slab.data_ptr<float>();
    }
    long row = 0, ci = 0;
    for (long b0 = 0; b0 < B; b0 += per, ++ci) {
        g.B = (int)std::min(per, B - b0);
        launch_conv_igemm(at_image(x.data_ptr(), b0, x_img), w.data_ptr(), at_image(y.data_ptr(), b0, y_img), g,
                          slab_ptr ? slab_ptr + row * 2 * N : nullptr, cur_stream(x), (int)cfg);
        row += tiles[ci];
    }
    if (want_stats) stats = shrink_rows(slab, m_tiles, 2 * N, kMaxFinalizeRows).view({-1, 2, N});
    return {y, stats};
}

// wt: [Cin][KH*KW][Cout] bf16 contiguous (transpose_weight(w))
// bn_*: when given (stride 1 only), dx is the gradient w.r.t. the OUTPUT of a BN+ReLU with input bn_x / activity
// mask bn_mask / statistics bn_mean, bn_invstd, and the second result holds the BN-backward partial sums
// [P<=32, 2, Cin] computed in the epilogue (what bn_bwd_reduce would return for this dx).
// bn2_*: a SECOND BatchNorm that receives this same dx through the same ReLU gate (bottleneck bn3 + the BatchNorm of
// the downsample branch): when the launch goes to the GEMM core (pointwise, wide output) the third result holds that
// BatchNorm's partial sums [P, 2, Cin] as well (row 0 = sum g, shared); otherwise it is undefined and the caller
// reduces on its own.
std::tuple<Tensor, Tensor, Tensor> conv2d_dgrad(const Tensor& dy, const Tensor& wt, std::vector<int64_t> x_shape, int64_t KH,
                                                int64_t KW, int64_t stride, int64_t pad, const c10::optional<Tensor>& addend,
                                                const c10::optional<Tensor>& bn_x, const c10::optional<Tensor>& bn_mask,
                                                const c10::optional<Tensor>& bn_mean, const c10::optional<Tensor>& bn_invstd,
                                                const c10::optional<Tensor>& addend_mask, int64_t addend_stride, int64_t cfg,
                                                const c10::optional<Tensor>& bn2_x, const c10::optional<Tensor>& bn2_mean,
                                                const c10::optional<Tensor>& bn2_invstd) {
    check_nhwc_bf16(dy, "dy");
    const bool bnr2 = bn2_x.has_value() && bn2_x->defined();
    TORCH_CHECK(addend_stride == 1 || addend_stride == 2, "addend_stride must be 1 or 2");
    const bool bnr = bn_x.has_value() && bn_x->defined();
    const void* addend_ptr = nullptr;
    const uint8_t* addend_mask_ptr = nullptr;
    if (addend.has_value() && addend->defined()) {
        check_nhwc_bf16(*addend, "addend");
        if (addend_stride == 2) {
            TORCH_CHECK(addend->size(0) == x_shape[0] && addend->size(1) == x_shape[1] &&
                            addend->size(2) == (x_shape[2] + 1) / 2 && addend->size(3) == (x_shape[3] + 1) / 2,
                        "compact stride-2 addend must be [B, C, ceil(H/2), ceil(W/2)]");
            TORCH_CHECK(!(addend_mask.has_value() && addend_mask->defined()), "compact addend takes no mask");
        } else {
            TORCH_CHECK(addend->sizes().vec() == x_shape, "addend must have the shape of dx");
        }
        TORCH_CHECK(stride == 1, "addend fusion is implemented for stride-1 data gradients");
        addend_ptr = addend->data_ptr();
        if (addend_mask.has_value() && addend_mask->defined()) {
            TORCH_CHECK(addend_mask->is_cuda() && addend_mask->scalar_type() == at::kByte &&
                            addend_mask->is_contiguous() && addend_mask->numel() == addend->numel() / 8,
                        "addend_mask: expected contiguous uint8 [pixels, C/8]");
            addend_mask_ptr = addend_mask->data_ptr<uint8_t>();
        }
    } else {
        TORCH_CHECK(!(addend_mask.has_value() && addend_mask->defined()), "addend_mask without addend");
    }
    TORCH_CHECK(wt.is_cuda() && wt.scalar_type() == at::kBFloat16 && wt.dim() == 3 && wt.is_contiguous(),
                "wt: expected contiguous bf16 [Cin][taps][Cout]");
    TORCH_CHECK(x_shape.size() == 4, "x_shape must be [B,C,H,W]");
    const long B = x_shape[0], C = x_shape[1], H = x_shape[2], W = x_shape[3];
    const long N = dy.size(1), Ho = dy.size(2), Wo = dy.size(3);
    TORCH_CHECK(dy.size(0) == B && wt.size(0) == C && wt.size(1) == KH * KW && wt.size(2) == N, "dgrad shape mismatch");
    TORCH_CHECK(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1, "dgrad geometry mismatch");
    TORCH_CHECK(C % 64 == 0 && N % 64 == 0, "conv2d_dgrad: channels must be multiples of 64");
    TORCH_CHECK(stride == 1 || stride == 2, "conv2d_dgrad: stride must be 1 or 2");
    TORCH_CHECK(KH * KW <= kMaxTaps && pad < 64, "unsupported kernel geometry");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dy.device());
    hipStream_t s = cur_stream(dy);

    auto base_geom = [&]() {
        ConvGeom g;
        std::memset(&g, 0, sizeof(g));
        g.B = B; g.Hin = Ho; g.Win = Wo; g.C = N;      // gather from dY, reduce over Cout
        g.Hout = H; g.Wout = W; g.N = C;               // write dX, produce Cin
        g.is = 1; g.wtaps = KH * KW;
        return g;
    };
    if ((cfg == kCfgWin64 || (cfg == 0 && win64_enough(B, H))) && !addend_ptr &&
        conv_win64_ok((int)N, (int)C, (int)KH, (int)KW, (int)stride, (int)pad, (int)H, (int)W)) {
        Tensor dx = empty_nhwc(B, C, H, W, dy);
        const int grid = conv_win64_grid((int)B, (int)H);
        Tensor psums, slab;
        const void* bx = nullptr;
        const uint8_t* bm = nullptr;
        const float *bmean = nullptr, *binv = nullptr;
        if (bnr) {
            check_nhwc_bf16(*bn_x, "bn_x");
            TORCH_CHECK(bn_x->sizes().vec() == x_shape, "bn_x must have the shape of dx");
            TORCH_CHECK(bn_mask.has_value() && bn_mask->is_cuda() && bn_mask->scalar_type() == at::kByte &&
                            bn_mask->is_contiguous() && bn_mask->numel() == B * H * W * (C / 8),
                        "bn_mask: expected contiguous uint8 [B*H*W, C/8]");
            TORCH_CHECK(bn_mean.has_value() && bn_invstd.has_value(), "bn_mean / bn_invstd missing");
            check_vec_f32(*bn_mean, C, "bn_mean"); check_vec_f32(*bn_invstd, C, "bn_invstd");
            bx = bn_x->data_ptr(); bm = bn_mask->data_ptr<uint8_t>();
            bmean = bn_mean->data_ptr<float>(); binv = bn_invstd->data_ptr<float>();
            slab = at::empty({grid, 2 * C}, dy.options().dtype(at::kFloat));
        }
        launch_conv_win64(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), bnr ? slab.data_ptr<float>() : nullptr, (int)B, (int)H,
                          (int)W, 1, bx, bm, bmean, binv, s);
        if (bnr) psums = shrink_rows(slab, grid, 2 * C, kReduceChunks).view({-1, 2, C});
        return {dx, psums, Tensor()};
    }
    if (cfg == kCfgWin64) cfg = 0;  // (with an addend: implicit-GEMM kernel)
    const bool taps = conv_gemm_taps_ok(B, N, C, H, W, KH, KW, stride, pad, cfg);
    // With everything fused into its epilogue (gated addend + BN-backward sums: three activation-sized streams next to
    // one K-tile of MFMAs) the layer-1 launch (Cout = 64) is a streaming kernel, and the 256x128 implicit-GEMM build
    // with two workgroups per CU keeps more loads in flight than one persistent workgroup: 1435 vs 1600 us at B = 1280
    // (bench/micro/fat_dgrad.py; from Cout = 128 on the two are within 2 %, at Cout = 512 the GEMM core wins by 6 %).
    static const bool stream_rule = [] {
        const char* e = std::getenv("PDT_FUSED_DGRAD_IGEMM");
        return !(e && std::string(e) == "0");
    }();
    const bool fused_stream = stream_rule && (addend_ptr || bnr) && KH == 1 && KW == 1 && N <= 64 && cfg == 0;
    if (fused_stream) cfg = 10;
    if (stride == 1 && !(addend_ptr && addend_stride == 2) && !fused_stream &&
        (taps || (conv_gemm_core_ok(N, C, KH, KW, stride, pad, cfg) && (cfg == kCfgGemmCore || enough_gemm_tiles(B * H * W, C))))) {
        // dx[M,Cin] = dy[M,Cout] . wt[Cin,Cout]^T on the GEMM core (spatial filters: mirrored taps)
        Tensor dx = empty_nhwc(B, C, H, W, dy);
        GemmArgs ga;
        std::memset(&ga, 0, sizeof(ga));
        ga.A = dy.data_ptr(); ga.B = wt.data_ptr(); ga.C = dx.data_ptr();
        ga.M = B * H * W; ga.N = (int)C; ga.K = (int)N;
        ga.lda = (int)N; ga.ldb = (int)(N * KH * KW); ga.ldc = (int)C;
        if (taps) set_gemm_taps(ga, H, W, KH, KW, pad, true);
        Tensor psums, slab;
        const int m_tiles = gemm_m_tiles(ga.M);
        if (addend_ptr || bnr) {
            ga.conv_mode = 2;
            ga.addend = addend_ptr;
            ga.addend_mask = addend_mask_ptr;
        }
        if (bnr) {
            check_nhwc_bf16(*bn_x, "bn_x");
            TORCH_CHECK(bn_x->sizes().vec() == x_shape, "bn_x must have the shape of dx");
            TORCH_CHECK(bn_mask.has_value() && bn_mask->is_cuda() && bn_mask->scalar_type() == at::kByte &&
                            bn_mask->is_contiguous() && bn_mask->numel() == B * H * W * (C / 8),
                        "bn_mask: expected contiguous uint8 [B*H*W, C/8]");
            TORCH_CHECK(bn_mean.has_value() && bn_invstd.has_value(), "bn_mean / bn_invstd missing");
            check_vec_f32(*bn_mean, C, "bn_mean"); check_vec_f32(*bn_invstd, C, "bn_invstd");
            ga.bnr_x = bn_x->data_ptr();
            ga.bnr_mask = bn_mask->data_ptr<uint8_t>();
            ga.bnr_mean = bn_mean->data_ptr<float>();
            ga.bnr_invstd = bn_invstd->data_ptr<float>();
            slab = at::empty({m_tiles, 2 * C}, dy.options().dtype(at::kFloat));
            ga.stats_slab = slab.data_ptr<float>();
        }
        Tensor psums2, slab2;
        const bool dual = bnr && bnr2 && !taps;
        if (dual) {
            check_nhwc_bf16(*bn2_x, "bn2_x");
            TORCH_CHECK(bn2_x->sizes().vec() == x_shape, "bn2_x must have the shape of dx");
            TORCH_CHECK(bn2_mean.has_value() && bn2_invstd.has_value(), "bn2_mean / bn2_invstd missing");
            check_vec_f32(*bn2_mean, C, "bn2_mean"); check_vec_f32(*bn2_invstd, C, "bn2_invstd");
            ga.bnr_x2 = bn2_x->data_ptr();
            ga.bnr_mean2 = bn2_mean->data_ptr<float>();
            ga.bnr_invstd2 = bn2_invstd->data_ptr<float>();
            slab2 = at::empty({m_tiles, 2 * C}, dy.options().dtype(at::kFloat));
            ga.stats_slab2 = slab2.data_ptr<float>();
        }
        launch_gemm_nt(ga, s);
        if (bnr) psums = shrink_rows(slab, m_tiles, 2 * C, kReduceChunks).view({-1, 2, C});
        if (dual) psums2 = shrink_rows(slab2, m_tiles, 2 * C, kReduceChunks).view({-1, 2, C});
        return {dx, psums, psums2};
    }
    if (cfg == kCfgGemmCore) cfg = 0;  // (compact stride-2 addend: implicit-GEMM kernel only)
    const long dy_img = Ho * Wo * N * 2, dx_img = H * W * C * 2;
    const long per = chunk_images(B, std::max(dy_img, dx_img));
    if (stride == 1) {
        Tensor dx = empty_nhwc(B, C, H, W, dy);
        ConvGeom g = base_geom();
        g.nOH = H; g.nOW = W; g.os = 1;
        g.ntaps = KH * KW;
        for (int r = 0; r < KH; ++r)
            for (int q = 0; q < KW; ++q) {
                const int t = r * KW + q;
                g.set_tap(t, (int)pad - r, (int)pad - q, t);
            }
        g.addend = addend_ptr;
        g.addend_mask = addend_mask_ptr;
        g.addend_stride = (int)addend_stride;
        const long add_img = addend_stride == 2 ? ((H + 1) / 2) * ((W + 1) / 2) * C * 2 : dx_img;
        const uint8_t* bmask = nullptr;
        if (bnr) {
            check_nhwc_bf16(*bn_x, "bn_x");
            TORCH_CHECK(bn_x->sizes().vec() == x_shape, "bn_x must have the shape of dx");
            TORCH_CHECK(bn_mask.has_value() && bn_mask->is_cuda() && bn_mask->scalar_type() == at::kByte &&
                            bn_mask->is_contiguous() && bn_mask->numel() == B * H * W * (C / 8),
                        "bn_mask: expected contiguous uint8 [B*H*W, C/8]");
            TORCH_CHECK(bn_mean.has_value() && bn_invstd.has_value(), "bn_mean / bn_invstd missing");
            check_vec_f32(*bn_mean, C, "bn_mean"); check_vec_f32(*bn_invstd, C, "bn_invstd");
            g.bnr_x = bn_x->data_ptr();
            bmask = bn_mask->data_ptr<uint8_t>();
            g.bnr_mask = bmask;
            g.bnr_mean = bn_mean->data_ptr<float>();
            g.bnr_invstd = bn_invstd->data_ptr<float>();
        }
        std::vector<int> tiles;
        int m_tiles = 0;
        for (long b0 = 0; b0 < B; b0 += per) {
            g.B = (int)std::min(per, B - b0);
            tiles.push_back(conv_igemm_m_tiles(g, nullptr, (int)cfg));
            m_tiles += tiles.back();
        }
        Tensor psums, slab;
        float* slab_ptr = nullptr;
        if (bnr) {
            slab = at::empty({m_tiles, 2 * C}, dy.options().dtype(at::kFloat));
            slab_ptr = slab.data_ptr<float>();
        }
        long row = 0, ci = 0;
        for (long b0 = 0; b0 < B; b0 += per, ++ci) {
            ConvGeom gc = g;
            gc.B = (int)std::min(per, B - b0);
            gc.addend = at_image(addend_ptr, b0, add_img);
            gc.addend_mask = reinterpret_cast<const uint8_t*>(at_image(addend_mask_ptr, b0, dx_img / 16));
            if (bnr) {
                gc.bnr_x = at_image(bn_x->data_ptr(), b0, dx_img);
                gc.bnr_mask = reinterpret_cast<const uint8_t*>(at_image(bmask, b0, dx_img / 16));
            }
            launch_conv_igemm(at_image(dy.data_ptr(), b0, dy_img), wt.data_ptr(), at_image(dx.data_ptr(), b0, dx_img), gc,
                              slab_ptr ? slab_ptr + row * 2 * C : nullptr, s, (int)cfg);
            row += tiles[ci];
        }
        if (bnr) psums = shrink_rows(slab, m_tiles, 2 * C, kReduceChunks).view({-1, 2, C});
        return {dx, psums, Tensor()};
    }
    TORCH_CHECK(!bnr, "the fused BN-backward reduction is implemented for stride-1 data gradients");
    // stride 2: one launch per output parity class; a class without taps stays zero
    bool need_zero = false;
    std::vector<ConvGeom> launches;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            ConvGeom g = base_geom();
            g.os = 2; g.ooh = ph; g.oow = pw;
            g.nOH = (H - ph + 1) / 2; g.nOW = (W - pw + 1) / 2;
            int nt = 0;
            for (int r = 0; r < KH; ++r) {
                if (((ph + pad - r) % 2) != 0) continue;
                for (int q = 0; q < KW; ++q) {
                    if (((pw + pad - q) % 2) != 0) continue;
                    g.set_tap(nt, (int)((ph + pad - r) / 2), (int)((pw + pad - q) / 2), (int)(r * KW + q));
                    ++nt;
                }
            }
            g.ntaps = nt;
            if (g.nOH <= 0 || g.nOW <= 0) continue;
            if (nt == 0) { need_zero = true; continue; }
            launches.push_back(g);
        }
    Tensor dx = empty_nhwc(B, C, H, W, dy);  // (at::zeros would drop the channels_last request)
    if (need_zero) dx.zero_();
    for (long b0 = 0; b0 < B; b0 += per)
        for (const auto& g : launches) {
            ConvGeom gc = g;
            gc.B = (int)std::min(per, B - b0);
            launch_conv_igemm(at_image(dy.data_ptr(), b0, dy_img), wt.data_ptr(), at_image(dx.data_ptr(), b0, dx_img), gc,
                              nullptr, s, (int)cfg);
        }
    return {dx, Tensor(), Tensor()};
}

bool wgrad_win_enabled() {
    static const bool on = [] {
        // measured on MI355X (profiles/r1_wgrad_win_vs_tap.md): with three taps per wave the window kernel's
        // per-chunk index math outweighs its MFMA work and the per-tap kernel wins on every ResNet 3x3
        // shape, so it is opt-in for convs (the packed stem always uses it: nothing else fits K = 147)
        const char* e = std::getenv("PDT_WGRAD_WIN");
        return e && std::string(e) == "1";
    }();
    return on;
}

Tensor conv2d_wgrad(const Tensor& dy, const Tensor& x, int64_t KH, int64_t KW, int64_t stride, int64_t pad,
                    int64_t splits) {
    check_nhwc_bf16(dy, "dy");
    check_nhwc_bf16(x, "x");
    const long B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    const long N = dy.size(1), Ho = dy.size(2), Wo = dy.size(3);
    TORCH_CHECK(dy.size(0) == B, "batch mismatch");
    TORCH_CHECK(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1, "wgrad geometry mismatch");
    TORCH_CHECK(C % 64 == 0 && N % 64 == 0, "conv2d_wgrad: channels must be multiples of 64");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const long wsize = N * KH * KW * C;
    const long x_img = H * W * C * 2, dy_img = Ho * Wo * N * 2;
    const long per = chunk_images(B, std::max(x_img, dy_img));
    const long nchunks = (B + per - 1) / per;
    TORCH_CHECK(per * Ho * Wo < (1L << 31), "too many output pixels per launch");
    // every batch chunk is a group of split-K rows of the same slab; one reduction over all of them
    auto finish = [&](const Tensor& slab, long rows) {
        Tensor flat = rows == 1 ? slab.view({wsize}) : reduce_slab(slab, (int)rows, wsize);
        return flat.view({N, KH, KW, C}).permute({0, 3, 1, 2});  // physical [N][KH][KW][C] == channels_last [N,C,KH,KW]
    };
    if (splits <= 0 && wgrad_win64_ok((int)std::min(per, B), (int)H, (int)W, (int)C, (int)N, (int)KH, (int)KW,
                                      (int)stride, (int)pad)) {  // wgrad_win64.hip
        long rows = 0;
        for (long b0 = 0; b0 < B; b0 += per) rows += wgrad_win64_grid((int)std::min(per, B - b0), (int)H);
        Tensor slab = at::empty({rows, wsize}, x.options().dtype(at::kFloat));
        long row = 0;
        for (long b0 = 0; b0 < B; b0 += per) {
            const int nb = (int)std::min(per, B - b0);
            launch_wgrad_win64(at_image(dy.data_ptr(), b0, dy_img), at_image(x.data_ptr(), b0, x_img),
                               slab.data_ptr<float>() + row * wsize, nb, (int)H, (int)W, cur_stream(x));
            row += wgrad_win64_grid(nb, (int)H);
        }
        return finish(slab, rows);
    }
    if (KH == 3 && KW == 3 && wgrad_win_enabled()) {  // all nine taps per workgroup from one staged window
        std::vector<WinGeom> plans;
        long rows = 0;
        bool ok = true;
        for (long b0 = 0; b0 < B && ok; b0 += per) {
            WinGeom wg{(int)std::min(per, B - b0), (int)Ho, (int)Wo, (int)N, (int)H, (int)W, (int)C, (int)C, 3, 3,
                       (int)stride, (int)stride, (int)pad, (int)pad, 0, 0, 0};
            ok = wgrad_win_plan(wg);
            plans.push_back(wg);
            rows += wg.splits;
        }
        if (ok) {
            Tensor slab = at::empty({rows, wsize}, x.options().dtype(at::kFloat));
            long row = 0, ci = 0;
            for (long b0 = 0; b0 < B; b0 += per, ++ci) {
                launch_wgrad_win(at_image(dy.data_ptr(), b0, dy_img), at_image(x.data_ptr(), b0, x_img),
                                 slab.data_ptr<float>() + row * wsize, plans[ci], cur_stream(x));
                row += plans[ci].splits;
            }
            return finish(slab, rows);
        }
    }
    std::vector<WgradGeom> geoms;
    long rows = 0;
    for (long b0 = 0; b0 < B; b0 += per) {
        const long nb = std::min(per, B - b0);
        WgradGeom g{(int)nb, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, (int)N, (int)KH, (int)KW, (int)stride, (int)pad, 1};
        g.splits = conv_wgrad_splits(g);
        if (splits > 0) {  // explicit split-K factor (tests force the multi-split slab path on small shapes)
            const long max_splits = std::max<long>(1, (nb * Ho * Wo + 31) / 32);
            g.splits = (int)std::min<long>(std::min<long>(splits, 512), max_splits);
        } else if (nchunks > 1 && !gemm_tn_ok(g)) {
            // the chunks run back to back: the small-tile kernel keeps its total workgroup count.  (The 256 x 256 kernel
            // is one workgroup per CU per LAUNCH: dividing its splits left half the CUs idle in every chunk.)
            g.splits = std::max<int>(1, (int)((g.splits + nchunks - 1) / nchunks));
        }
        geoms.push_back(g);
        rows += g.splits;
    }
    Tensor slab = at::empty({rows, wsize}, x.options().dtype(at::kFloat));
    long row = 0, ci = 0;
    for (long b0 = 0; b0 < B; b0 += per, ++ci) {
        launch_conv_wgrad(at_image(dy.data_ptr(), b0, dy_img), at_image(x.data_ptr(), b0, x_img),
                          slab.data_ptr<float>() + row * wsize, geoms[ci], cur_stream(x));
        row += geoms[ci].splits;
    }
    return finish(slab, rows);
}

// ---- 3-channel stem: 7x7 / stride 2 / pad 3 on the packed image (conv_stem.hip) ---------------------
// -> (y, stats, xp): xp is the packed input the weight gradient reads (saved instead of x)
std::tuple<Tensor, Tensor, Tensor> conv_stem_fwd(const Tensor& x, const Tensor& w, bool want_stats) {
    check_nhwc_bf16(x, "x");
    check_conv_weight(w, "w");
    const long B = x.size(0), H = x.size(2), W = x.size(3), N = w.size(0);
    TORCH_CHECK(x.size(1) == 3 && w.size(1) == 3 && w.size(2) == 7 && w.size(3) == 7, "conv_stem_fwd: 7x7 conv on 3 channels");
    TORCH_CHECK(N % 64 == 0, "conv_stem_fwd: Cout must be a multiple of 64");
    TORCH_CHECK(H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2, "conv_stem_fwd: even input size expected");
    const long Hp = H + 6, Wp = W + 8, Ho = H / 2, Wo = W / 2;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    hipStream_t s = cur_stream(x);
    Tensor xp = at::empty({B, Hp, Wp, 4}, x.options().memory_format(at::MemoryFormat::Contiguous));
    Tensor wp = at::empty({N, 7, 32}, x.options().memory_format(at::MemoryFormat::Contiguous));
    launch_stem_pack_w(w.data_ptr(), wp.data_ptr(), (int)N, s);
    Tensor y = empty_nhwc(B, N, Ho, Wo, x);
    ConvGeom g;
    std::memset(&g, 0, sizeof(g));
    g.B = B; g.Hin = Hp; g.Win = Wp; g.C = 32; g.pix_elems = 4; g.Hout = Ho; g.Wout = Wo; g.N = N;
    g.nOH = Ho; g.nOW = Wo; g.os = 1; g.is = 2; g.ntaps = 7; g.wtaps = 7;
    for (int t = 0; t < 7; ++t) g.set_tap(t, t, 0, t);
    const long x_img = H * W * 3 * 2, xp_img = Hp * Wp * 8, y_img = Ho * Wo * N * 2;
    const long per = chunk_images(B, std::max(xp_img, y_img));
    std::vector<int> tiles;
    int m_tiles = 0;
    // (decided on the whole batch: every chunk takes the same kernel)
    const bool win = conv_stem_win_ok((int)std::min(per, B), (int)Ho, (int)Wo, (int)N);
    for (long b0 = 0; b0 < B; b0 += per) {
        g.B = (int)std::min(per, B - b0);
        tiles.push_back(win ? conv_stem_win_grid(g.B, (int)Ho) : conv_igemm_m_tiles(g, nullptr));
        m_tiles += tiles.back();
    }
    Tensor stats, slab;
    float* slab_ptr = nullptr;
    if (want_stats) {
        slab = at::empty({m_tiles, 2 * N}, x.options().dtype(at::kFloat));
        slab_ptr = slab.data_ptr<float>();
    }
    long row = 0, ci = 0;
    for (long b0 = 0; b0 < B; b0 += per, ++ci) {
        g.B = (int)std::min(per, B - b0);
        launch_stem_pack_x(at_image(x.data_ptr(), b0, x_img), at_image(xp.data_ptr(), b0, xp_img), g.B, (int)H, (int)W, s);
        if (win)
            launch_conv_stem_win(at_image(xp.data_ptr(), b0, xp_img), wp.data_ptr(), at_image(y.data_ptr(), b0, y_img),
                                 slab_ptr ? slab_ptr + row * 2 * N : nullptr, g.B, (int)Ho, (int)Wo, s);
        else
            launch_conv_igemm(at_image(xp.data_ptr(), b0, xp_img), wp.data_ptr(), at_image(y.data_ptr(), b0, y_img), g,
                              slab_ptr ? slab_ptr + row * 2 * N : nullptr, s);
        row += tiles[ci];
    }
    if (want_stats) stats = shrink_rows(slab, m_tiles, 2 * N, kMaxFinalizeRows).view({-1, 2, N});
    return {y, stats, xp};
}

Tensor stem_pack_x(const Tensor& x) {
    check_nhwc_bf16(x, "x");
    const long B = x.size(0), H = x.size(2), W = x.size(3);
    TORCH_CHECK(x.size(1) == 3 && H % 2 == 0 && W % 2 == 0, "stem_pack_x: [B,3,H,W] with even H, W");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor xp = at::empty({B, H + 6, W + 8, 4}, x.options().memory_format(at::MemoryFormat::Contiguous));
    const long x_img = H * W * 3 * 2, xp_img = (H + 6) * (W + 8) * 8;
    const long per = chunk_images(B, xp_img);
    for (long b0 = 0; b0 < B; b0 += per)
        launch_stem_pack_x(at_image(x.data_ptr(), b0, x_img), at_image(xp.data_ptr(), b0, xp_img),
                           (int)std::min(per, B - b0), (int)H, (int)W, cur_stream(x));
    return xp;
}

// dy [B,64,Ho,Wo] NHWC, xp from conv_stem_fwd -> dW logical [64,3,7,7] (channels_last, fp32)
Tensor conv_stem_wgrad(const Tensor& dy, const Tensor& xp) {
    check_nhwc_bf16(dy, "dy");
    TORCH_CHECK(xp.is_cuda() && xp.scalar_type() == at::kBFloat16 && xp.dim() == 4 && xp.is_contiguous() && xp.size(3) == 4,
                "xp: expected the packed stem input [B, H+6, W+8, 4]");
    const long B = dy.size(0), N = dy.size(1), Ho = dy.size(2), Wo = dy.size(3);
    const long Hp = xp.size(1), Wp = xp.size(2);
    TORCH_CHECK(xp.size(0) == B && Hp == 2 * Ho + 6 && Wp == 2 * Wo + 8 && N % 64 == 0, "conv_stem_wgrad: shape mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dy.device());
    const long wsize = N * 147;
    const long dy_img = Ho * Wo * N * 2, xp_img = Hp * Wp * 8;
    const long per = chunk_images(B, std::max(dy_img, xp_img));
    if (conv_stem_wgrad_win_ok((int)std::min(per, B), (int)Ho, (int)Wo, (int)N)) {  // conv_stem_win.hip
        long rows = 0;
        for (long b0 = 0; b0 < B; b0 += per) rows += conv_stem_wgrad_win_grid((int)std::min(per, B - b0), (int)Ho);
        Tensor slab = at::empty({rows, wsize}, dy.options().dtype(at::kFloat));
        long row = 0;
        for (long b0 = 0; b0 < B; b0 += per) {
            const int nb = (int)std::min(per, B - b0);
            launch_conv_stem_wgrad_win(at_image(dy.data_ptr(), b0, dy_img), at_image(xp.data_ptr(), b0, xp_img),
                                       slab.data_ptr<float>() + row * wsize, nb, (int)Ho, (int)Wo, cur_stream(dy));
            row += conv_stem_wgrad_win_grid(nb, (int)Ho);
        }
        Tensor flat = rows == 1 ? slab.view({wsize}) : reduce_slab(slab, (int)rows, wsize);
        return flat.view({N, 7, 7, 3}).permute({0, 3, 1, 2});
    }
    std::vector<WinGeom> plans;
    long rows = 0;
    for (long b0 = 0; b0 < B; b0 += per) {
        WinGeom g{(int)std::min(per, B - b0), (int)Ho, (int)Wo, (int)N, (int)Hp, (int)(Wp / 2), 8, 32, 7, 1, 2, 1, 0, 0, 1, 0, 0};
        TORCH_CHECK(wgrad_win_plan(g), "conv_stem_wgrad: output rows of ", Wo, " pixels do not fit the window kernel");
        plans.push_back(g);
        rows += g.splits;
    }
    Tensor slab = at::empty({rows, wsize}, dy.options().dtype(at::kFloat));
    long row = 0, ci = 0;
    for (long b0 = 0; b0 < B; b0 += per, ++ci) {
        launch_wgrad_win(at_image(dy.data_ptr(), b0, dy_img), at_image(xp.data_ptr(), b0, xp_img),
                         slab.data_ptr<float>() + row * wsize, plans[ci], cur_stream(dy));
        row += plans[ci].splits;
    }
    Tensor flat = rows == 1 ? slab.view({wsize}) : reduce_slab(slab, (int)rows, wsize);
    return flat.view({N, 7, 7, 3}).permute({0, 3, 1, 2});
}

Tensor transpose_weight(const Tensor& w) {
    check_conv_weight(w, "w");
    const long N = w.size(0), C = w.size(1), T = w.size(2) * w.size(3);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(w.device());
    Tensor wt = at::empty({C, T, N}, w.options().memory_format(at::MemoryFormat::Contiguous));
    launch_transpose_ntc(w.data_ptr(), wt.data_ptr(), (int)N, (int)T, (int)C, cur_stream(w));
    return wt;
}

// all weights of a model in ONE launch (<= 64 per launch): returns the [C][T][N] transposes
std::vector<Tensor> transpose_weights(std::vector<Tensor> ws, c10::optional<std::vector<Tensor>> outs) {
    std::vector<Tensor> res;
    if (ws.empty()) return res;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(ws[0].device());
    hipStream_t stream = cur_stream(ws[0]);
    TransposeArgs a;
    a.n = 0;
    a.block_start[0] = 0;
    for (size_t i = 0; i < ws.size(); ++i) {
        const Tensor& w = ws[i];
        check_conv_weight(w, "w");
        const long N = w.size(0), C = w.size(1), T = w.size(2) * w.size(3);
        Tensor wt;
        if (outs.has_value()) {
            wt = (*outs)[i];
            TORCH_CHECK(wt.is_cuda() && wt.scalar_type() == at::kBFloat16 && wt.is_contiguous() && wt.dim() == 3 &&
                            wt.size(0) == C && wt.size(1) == T && wt.size(2) == N,
                        "transpose_weights: bad output buffer");
        } else {
            wt = at::empty({C, T, N}, w.options().memory_format(at::MemoryFormat::Contiguous));
        }
        res.push_back(wt);
        a.src[a.n] = w.data_ptr();
        a.dst[a.n] = wt.data_ptr();
        a.N[a.n] = (int)N; a.T[a.n] = (int)T; a.C[a.n] = (int)C;
        a.block_start[a.n + 1] = a.block_start[a.n] + (int)(((C + 31) / 32) * ((N + 31) / 32) * T);
        if (++a.n == kTransposeMax) {
            launch_multi_transpose(a, stream);
            a.n = 0;
            a.block_start[0] = 0;
        }
    }
    launch_multi_transpose(a, stream);
    return res;
}

// --------------------------------------------------------------------------------------- bn
Tensor bn_stats(const Tensor& x) {
    check_nhwc_bf16(x, "x");
    const long C = x.size(1), M = x.numel() / C;
    TORCH_CHECK(bn_channels_ok(C), "bn: unsupported channel count ", C);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int parts = bn_parts(M, (int)C);
    Tensor slab = at::empty({parts, 2 * C}, x.options().dtype(at::kFloat));
    launch_bn_stats(x.data_ptr(), M, (int)C, slab.data_ptr<float>(), parts, cur_stream(x));
    return slab.view({parts, 2, C});
}

void check_partials(const Tensor& stats, const char* name) {
    TORCH_CHECK(stats.is_cuda() && stats.scalar_type() == at::kFloat && stats.dim() == 3 && stats.size(1) == 2 &&
                    stats.is_contiguous() && stats.size(0) >= 1 && stats.size(0) <= kMaxFinalizeRows,
                name, ": expected contiguous fp32 [P<=512, 2, C]");
}
const float* count_ptr(const c10::optional<Tensor>& count_dev) {
    if (count_dev.has_value() && count_dev->defined()) {
        TORCH_CHECK(count_dev->is_cuda() && count_dev->scalar_type() == at::kFloat && count_dev->numel() == 1,
                    "count_dev: expected a 1-element fp32 GPU tensor");
        return count_dev->data_ptr<float>();
    }
    return nullptr;
}

// stats: partial rows [P,2,C]
// ipc (SyncBatchNorm, world > 1): the kernel exchanges the statistics with the peers over the IPC mailboxes itself;
// the third result is then the global element count (1-element fp32 tensor) the backward kernels divide by
std::tuple<Tensor, Tensor, Tensor> bn_finalize(const Tensor& stats, double count, const c10::optional<Tensor>& count_dev,
                                               double eps, double momentum, const c10::optional<Tensor>& running_mean,
                                               const c10::optional<Tensor>& running_var,
                                               const c10::optional<Tensor>& num_batches_tracked,
                                               const std::shared_ptr<IpcAllReduce>& ipc) {
    check_partials(stats, "stats");
    const long P = stats.size(0), C = stats.size(2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(stats.device());
    Tensor mean = at::empty({C}, stats.options()), invstd = at::empty({C}, stats.options());
    Tensor count_total;
    const IpcMailboxes* mb = nullptr;
    if (ipc && ipc->world_size() > 1) {
        TORCH_CHECK(count >= 1.0 && count < 4294967296.0 && !(count_dev.has_value() && count_dev->defined()),
                    "bn_finalize: the exchanged count is this rank's host-side element count");
        mb = &ipc->mailboxes_for(3 * C + 1);
        count_total = at::empty({1}, stats.options());
    }
    float *rm = nullptr, *rv = nullptr;
    if (running_mean.has_value() && running_mean->defined()) {
        check_vec_f32(*running_mean, C, "running_mean");
        check_vec_f32(*running_var, C, "running_var");
        rm = running_mean->data_ptr<float>();
        rv = running_var->data_ptr<float>();
    }
    long* nbt = nullptr;
    if (num_batches_tracked.has_value() && num_batches_tracked->defined()) {
        TORCH_CHECK(num_batches_tracked->is_cuda() && num_batches_tracked->scalar_type() == at::kLong &&
                        num_batches_tracked->numel() == 1,
                    "num_batches_tracked: expected a 1-element int64 GPU tensor");
        nbt = reinterpret_cast<long*>(num_batches_tracked->data_ptr<int64_t>());
    }
    launch_bn_finalize(stats.data_ptr<float>(), (int)P, (int)C, count, count_ptr(count_dev), (float)eps,
                       (float)momentum, rm, rv, nbt, mean.data_ptr<float>(), invstd.data_ptr<float>(), nullptr,
                       cur_stream(stats), mb, mb ? count_total.data_ptr<float>() : nullptr);
    return {mean, invstd, count_total};
}

// partial rows [P,2,C] -> [2,C] (SyncBN needs the local sums before the all-reduce)
Tensor bn_sum_partials(const Tensor& stats) {
    check_partials(stats, "stats");
    const long P = stats.size(0), C = stats.size(2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(stats.device());
    Tensor out = at::empty({2, C}, stats.options());
    launch_bn_finalize(stats.data_ptr<float>(), (int)P, (int)C, 1.f, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr,
                       nullptr, nullptr, out.data_ptr<float>(), cur_stream(stats));
    return out;
}

// -> (z, mask): mask is the 1-bit-per-element ReLU activity map [M, C/8] uint8 (undefined without relu);
// the backward kernels read it instead of the 16x larger z
std::tuple<Tensor, Tensor> bn_apply_act(const Tensor& x, const Tensor& mean, const Tensor& invstd, const Tensor& gamma,
                                        const Tensor& beta, const c10::optional<Tensor>& residual, bool relu,
                                        const c10::optional<std::vector<Tensor>>& res_affine) {
    check_nhwc_bf16(x, "x");
    const long C = x.size(1), M = x.numel() / C;
    TORCH_CHECK(bn_channels_ok(C), "bn: unsupported channel count ", C);
    check_vec_f32(mean, C, "mean"); check_vec_f32(invstd, C, "invstd");
    check_vec_f32(gamma, C, "gamma"); check_vec_f32(beta, C, "beta");
    const void* res = nullptr;
    if (residual.has_value() && residual->defined()) {
        check_nhwc_bf16(*residual, "residual");
        TORCH_CHECK(residual->sizes() == x.sizes(), "residual shape mismatch");
        res = residual->data_ptr();
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor z = at::empty_like(x), mask;
    if (relu) mask = at::empty({M, C / 8}, x.options().dtype(at::kByte));
    BnAffine ra{nullptr, nullptr, nullptr, nullptr};
    if (res_affine.has_value()) {  // the residual is the INPUT of a ReLU-less BN whose affine is applied here
        TORCH_CHECK(res != nullptr && res_affine->size() == 4, "res_affine: (mean, invstd, gamma, beta) of the residual");
        for (const auto& t : *res_affine) check_vec_f32(t, C, "res_affine");
        ra = BnAffine{(*res_affine)[0].data_ptr<float>(), (*res_affine)[1].data_ptr<float>(),
                      (*res_affine)[2].data_ptr<float>(), (*res_affine)[3].data_ptr<float>()};
    }
    launch_bn_apply(x.data_ptr(), res, z.data_ptr(), relu ? mask.data_ptr<uint8_t>() : nullptr, M, (int)C,
                    mean.data_ptr<float>(), invstd.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                    relu ? 1 : 0, ra, cur_stream(x));
    return {z, mask};
}

const void* relu_mask_ptr(const c10::optional<Tensor>& mask, long M, long C) {
    TORCH_CHECK(mask.has_value() && mask->defined(), "relu=True needs the activity mask of bn_apply_act");
    TORCH_CHECK(mask->is_cuda() && mask->scalar_type() == at::kByte && mask->is_contiguous() &&
                    mask->numel() == M * (C / 8),
                "mask: expected contiguous uint8 [M, C/8]");
    return mask->data_ptr();
}

Tensor bn_bwd_reduce(const Tensor& dz, const c10::optional<Tensor>& z, const Tensor& x, const Tensor& mean,
                     const Tensor& invstd, bool relu) {
    check_nhwc_bf16(dz, "dz"); check_nhwc_bf16(x, "x");
    TORCH_CHECK(dz.sizes() == x.sizes(), "dz/x shape mismatch");
    const long C = x.size(1), M = x.numel() / C;
    TORCH_CHECK(bn_channels_ok(C), "bn: unsupported channel count ", C);
    check_vec_f32(mean, C, "mean"); check_vec_f32(invstd, C, "invstd");
    const void* zp = nullptr;
    if (relu) {
        zp = relu_mask_ptr(z, M, C);
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int parts = bn_parts(M, (int)C);
    Tensor slab = at::empty({parts, 2 * C}, x.options().dtype(at::kFloat));
    launch_bn_bwd_reduce(dz.data_ptr(), zp, x.data_ptr(), M, (int)C, mean.data_ptr<float>(), invstd.data_ptr<float>(),
                         relu ? 1 : 0, slab.data_ptr<float>(), parts, cur_stream(x));
    return shrink_rows(slab, parts, 2 * C, kReduceChunks).view({-1, 2, C});
}

// psums: partial rows [P<=32,2,C] of bn_bwd_reduce (or [1,2,C] globally reduced sums); returns the
// summed [2,C] as third output (local affine gradients)
std::tuple<Tensor, Tensor, Tensor> bn_bwd_elemt(const Tensor& dz, const c10::optional<Tensor>& z, const Tensor& x,
                                                const Tensor& mean, const Tensor& invstd, const Tensor& gamma,
                                                const Tensor& psums, double count,
                                                const c10::optional<Tensor>& count_dev, bool relu, bool want_dres,
                                                const std::shared_ptr<IpcAllReduce>& ipc) {
    check_nhwc_bf16(dz, "dz"); check_nhwc_bf16(x, "x");
    TORCH_CHECK(dz.sizes() == x.sizes(), "dz/x shape mismatch");
    const long C = x.size(1), M = x.numel() / C;
    TORCH_CHECK(bn_channels_ok(C), "bn: unsupported channel count ", C);
    check_vec_f32(mean, C, "mean"); check_vec_f32(invstd, C, "invstd"); check_vec_f32(gamma, C, "gamma");
    check_partials(psums, "psums");
    TORCH_CHECK(psums.size(0) <= kReduceChunks && psums.size(2) == C, "psums: expected [P<=32, 2, C]");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(psums.data_ptr()) % 16 == 0 && C % 4 == 0, "psums alignment");
    const void* zp = nullptr;
    if (relu) {
        zp = relu_mask_ptr(z, M, C);
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor dx = at::empty_like(x), dres;
    Tensor sums = at::empty({2, C}, psums.options());
    Tensor coef = at::empty({3, C}, psums.options());
    void* drp = nullptr;
    if (want_dres) {
        dres = at::empty_like(x);
        drp = dres.data_ptr();
    }
    launch_bn_bwd_elemt(dz.data_ptr(), zp, x.data_ptr(), dx.data_ptr(), drp, M, (int)C, mean.data_ptr<float>(),
                        invstd.data_ptr<float>(), gamma.data_ptr<float>(), psums.data_ptr<float>(), (int)psums.size(0),
                        sums.data_ptr<float>(), (float)count, count_ptr(count_dev), relu ? 1 : 0,
                        coef.data_ptr<float>(), cur_stream(x),
                        ipc && ipc->world_size() > 1 ? &ipc->mailboxes_for(2 * C) : nullptr);
    return {dx, dres, sums};
}

// bn_bwd_elemt of two BatchNorms that share the gradient dz and its ReLU mask (z = relu(bn_a(xa) + bn_b(xb))):
// -> (dxa, dxb, sums_a, sums_b)
std::tuple<Tensor, Tensor, Tensor, Tensor> bn_bwd_elemt2(
    const Tensor& dz, const c10::optional<Tensor>& z, const Tensor& xa, const Tensor& mean_a, const Tensor& invstd_a,
    const Tensor& gamma_a, const Tensor& psums_a, const Tensor& xb, const Tensor& mean_b, const Tensor& invstd_b,
    const Tensor& gamma_b, const Tensor& psums_b, double count, const c10::optional<Tensor>& count_dev,
    const std::shared_ptr<IpcAllReduce>& ipc) {
    check_nhwc_bf16(dz, "dz"); check_nhwc_bf16(xa, "xa"); check_nhwc_bf16(xb, "xb");
    TORCH_CHECK(dz.sizes() == xa.sizes() && dz.sizes() == xb.sizes(), "dz/xa/xb shape mismatch");
    const long C = xa.size(1), M = xa.numel() / C;
    TORCH_CHECK(bn_channels_ok(C), "bn: unsupported channel count ", C);
    for (const Tensor* v : {&mean_a, &invstd_a, &gamma_a, &mean_b, &invstd_b, &gamma_b}) check_vec_f32(*v, C, "bn vector");
    for (const Tensor* p : {&psums_a, &psums_b}) {
        check_partials(*p, "psums");
        TORCH_CHECK(p->size(2) == C, "psums: expected [P, 2, C]");
    }
    const void* zp = (z.has_value() && z->defined()) ? relu_mask_ptr(z, M, C) : nullptr;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(xa.device());
    Tensor dxa = at::empty_like(xa), dxb = at::empty_like(xb);
    Tensor sums = at::empty({2, 2, C}, psums_a.options());
    Tensor coef = at::empty({2, 3, C}, psums_a.options());
    BnBwdSide a{xa.data_ptr(), dxa.data_ptr(), mean_a.data_ptr<float>(), invstd_a.data_ptr<float>(),
                gamma_a.data_ptr<float>(), psums_a.data_ptr<float>(), (int)psums_a.size(0), sums.data_ptr<float>(),
                coef.data_ptr<float>()};
    BnBwdSide b{xb.data_ptr(), dxb.data_ptr(), mean_b.data_ptr<float>(), invstd_b.data_ptr<float>(),
                gamma_b.data_ptr<float>(), psums_b.data_ptr<float>(), (int)psums_b.size(0),
                sums.data_ptr<float>() + 2 * C, coef.data_ptr<float>() + 3 * C};
    launch_bn_bwd_elemt2(dz.data_ptr(), zp, M, (int)C, (float)count, count_ptr(count_dev), a, b, cur_stream(xa),
                         ipc && ipc->world_size() > 1 ? &ipc->mailboxes_for(2 * C) : nullptr);
    return {dxa, dxb, sums[0], sums[1]};
}

// backward of bn_relu_maxpool_fwd: dy [B,C,Ho,Wo], argmax taps, u (conv output) and its 1-bit ReLU mask -> (du, sums [2,C])
std::tuple<Tensor, Tensor> bn_relu_maxpool_bwd(const Tensor& dy, const Tensor& argmax, const Tensor& u, const Tensor& mask,
                                               const Tensor& mean, const Tensor& invstd, const Tensor& gamma,
                                               double count, const c10::optional<Tensor>& count_dev,
                                               const std::shared_ptr<IpcAllReduce>& ipc) {
    check_nhwc_bf16(dy, "dy"); check_nhwc_bf16(u, "u");
    const long B = u.size(0), C = u.size(1), H = u.size(2), W = u.size(3), Ho = dy.size(2), Wo = dy.size(3);
    TORCH_CHECK(dy.size(0) == B && dy.size(1) == C && H % 2 == 0 && W % 2 == 0 && Ho == H / 2 && Wo == W / 2,
                "bn_relu_maxpool_bwd: expects even H, W and the 3x3/2/1 pooled shape");
    TORCH_CHECK(bn_channels_ok(C), "bn: unsupported channel count ", C);
    TORCH_CHECK(argmax.is_cuda() && argmax.scalar_type() == at::kByte && argmax.is_contiguous() && argmax.numel() == dy.numel(),
                "argmax: expected uint8 taps of the pooled shape");
    check_vec_f32(mean, C, "mean"); check_vec_f32(invstd, C, "invstd"); check_vec_f32(gamma, C, "gamma");
    const long M = B * H * W;
    const void* mp = relu_mask_ptr(mask, M, C);
    TORCH_CHECK(M * C < (1L << 32), "bn_relu_maxpool_bwd: more than 2^32 elements (use the unfused path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
    Tensor du = at::empty_like(u);
    const int parts = bn_parts(M / 4, (int)C);
    Tensor slab = at::empty({parts, 2 * C}, mean.options());
    Tensor sums = at::empty({2, C}, mean.options()), coef = at::empty({3, C}, mean.options());
    launch_bn_relu_maxpool_bwd(dy.data_ptr(), argmax.data_ptr<uint8_t>(), u.data_ptr(), static_cast<const uint8_t*>(mp),
                               du.data_ptr(), (int)B, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, mean.data_ptr<float>(),
                               invstd.data_ptr<float>(), gamma.data_ptr<float>(), (float)count, count_ptr(count_dev),
                               slab.data_ptr<float>(), parts, sums.data_ptr<float>(), coef.data_ptr<float>(), cur_stream(u),
                               ipc && ipc->world_size() > 1 ? &ipc->mailboxes_for(2 * C) : nullptr);
    return {du, sums};
}

// -------------------------------------------------------------------------------- layernorm
std::tuple<Tensor, Tensor, Tensor> layernorm_fwd(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kBFloat16 && x.is_contiguous() && x.dim() >= 2,
                "layernorm_fwd: expected a contiguous bf16 GPU tensor [..., D]");
    const long D = x.size(-1), M = x.numel() / D;
    TORCH_CHECK(layernorm_dim_ok((int)D), "layernorm: unsupported width ", D);
    TORCH_CHECK(reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0, "x must be 16-B aligned");
    check_vec_f32(gamma, D, "gamma"); check_vec_f32(beta, D, "beta");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor y = at::empty_like(x);
    Tensor mean = at::empty({M}, x.options().dtype(at::kFloat)), rstd = at::empty({M}, x.options().dtype(at::kFloat));
    if (M > 0)
        launch_layernorm_fwd(x.data_ptr(), gamma.data_ptr<float>(), beta.data_ptr<float>(), y.data_ptr(),
                             mean.data_ptr<float>(), rstd.data_ptr<float>(), M, (int)D, (float)eps, cur_stream(x));
    return {y, mean, rstd};
}

// -> (dx, dgamma, dbeta)
std::tuple<Tensor, Tensor, Tensor> layernorm_bwd(const Tensor& dy, const Tensor& x, const Tensor& mean,
                                                 const Tensor& rstd, const Tensor& gamma,
                                                 const c10::optional<Tensor>& addend) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kBFloat16 && x.is_contiguous() && x.dim() >= 2, "x: contiguous bf16");
    TORCH_CHECK(dy.is_cuda() && dy.scalar_type() == at::kBFloat16 && dy.is_contiguous() && dy.sizes() == x.sizes(),
                "dy: contiguous bf16 with the shape of x");
    const long D = x.size(-1), M = x.numel() / D;
    TORCH_CHECK(layernorm_dim_ok((int)D), "layernorm: unsupported width ", D);
    TORCH_CHECK(reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(dy.data_ptr()) % 16 == 0,
                "x / dy must be 16-B aligned");
    check_vec_f32(gamma, D, "gamma"); check_vec_f32(mean, M, "mean"); check_vec_f32(rstd, M, "rstd");
    const void* addp = nullptr;
    if (addend.has_value() && addend->defined()) {
        TORCH_CHECK(addend->is_cuda() && addend->scalar_type() == at::kBFloat16 && addend->is_contiguous() &&
                        addend->sizes() == x.sizes() && reinterpret_cast<uintptr_t>(addend->data_ptr()) % 16 == 0,
                    "addend: contiguous bf16 with the shape of x");
        addp = addend->data_ptr();
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor dx = at::empty_like(x);
    const int blocks = layernorm_bwd_blocks(M);
    Tensor slab = at::empty({blocks, 2 * D}, x.options().dtype(at::kFloat));
    launch_layernorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(),
                         gamma.data_ptr<float>(), dx.data_ptr(), slab.data_ptr<float>(), blocks, M, (int)D, addp,
                         cur_stream(x));
    Tensor sums = blocks == 1 ? slab.view({2 * D}) : reduce_slab(slab, blocks, 2 * D);
    return {dx, sums.narrow(0, D, D), sums.narrow(0, 0, D)};
}

// -------------------------------------------------------------------------------- attention
void check_attn(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kBFloat16 && t.is_contiguous() && t.dim() == 3, name,
                ": expected a contiguous bf16 GPU tensor [B, S, width]");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, ": data pointer must be 16-B aligned");
}

// qkv [B, S, 3*H*hd] (packed q | k | v, heads inside) -> (out [B, S, H*hd], lse [B, H, S]); hd in {32, 64, 128}
static long attn_head_dim(const Tensor& qkv, int64_t heads) {
    const long D = qkv.size(2) / 3;
    TORCH_CHECK(qkv.size(2) == 3 * D && heads > 0 && D % heads == 0, "attention: width must be 3 * heads * head_dim");
    const long hd = D / heads;
    TORCH_CHECK(attention_shape_ok((int)qkv.size(1), (int)hd), "attention: head_dim ", hd, " / sequence length ", qkv.size(1),
                " not supported (head_dim 32, 64 or 128)");
    return hd;
}

std::tuple<Tensor, Tensor> attention_fwd(const Tensor& qkv, int64_t heads) {
    check_attn(qkv, "qkv");
    const long B = qkv.size(0), S = qkv.size(1), D = qkv.size(2) / 3;
    const long hd = attn_head_dim(qkv, heads);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
    Tensor out = at::empty({B, S, D}, qkv.options());
    Tensor lse = at::empty({B, heads, S}, qkv.options().dtype(at::kFloat));
    // the grid's second dimension carries (batch, head): cut the batch where that would pass 65535
    const long per = std::max<long>(1, 65535 / heads);
    for (long b0 = 0; b0 < B; b0 += per) {
        const long nb = std::min(per, B - b0);
        launch_attention_fwd(static_cast<const char*>(qkv.data_ptr()) + b0 * S * 3 * D * 2,
                             static_cast<char*>(out.data_ptr()) + b0 * S * D * 2, lse.data_ptr<float>() + b0 * heads * S,
                             (int)nb, (int)S, (int)heads, (int)hd, cur_stream(qkv));
    }
    return {out, lse};
}

Tensor attention_bwd(const Tensor& dout, const Tensor& qkv, const Tensor& out, const Tensor& lse, int64_t heads) {
    check_attn(qkv, "qkv"); check_attn(out, "out"); check_attn(dout, "dout");
    const long B = qkv.size(0), S = qkv.size(1), D = qkv.size(2) / 3;
    const long hd = attn_head_dim(qkv, heads);
    TORCH_CHECK(out.size(0) == B && out.size(1) == S && out.size(2) == D && dout.sizes() == out.sizes(), "out / dout shape");
    TORCH_CHECK(lse.is_cuda() && lse.scalar_type() == at::kFloat && lse.is_contiguous() && lse.numel() == B * heads * S,
                "lse: expected contiguous fp32 [B, H, S]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
    Tensor dqkv = at::empty_like(qkv);
    Tensor dsum;
    if (attention_needs_dsum((int)S, (int)hd)) dsum = at::empty({B, heads, S}, lse.options());
    const long per = std::max<long>(1, 65535 / heads);
    for (long b0 = 0; b0 < B; b0 += per) {
        const long nb = std::min(per, B - b0);
        launch_attention_bwd(static_cast<const char*>(qkv.data_ptr()) + b0 * S * 3 * D * 2,
                             static_cast<const char*>(out.data_ptr()) + b0 * S * D * 2,
                             static_cast<const char*>(dout.data_ptr()) + b0 * S * D * 2, lse.data_ptr<float>() + b0 * heads * S,
                             static_cast<char*>(dqkv.data_ptr()) + b0 * S * 3 * D * 2,
                             dsum.defined() ? dsum.data_ptr<float>() + b0 * heads * S : nullptr, (int)nb, (int)S, (int)heads,
                             (int)hd, cur_stream(qkv));
    }
    return dqkv;
}

// ------------------------------------------------------------------------------------ pools
std::tuple<Tensor, Tensor> maxpool3x3s2_fwd(const Tensor& x) {
    check_nhwc_bf16(x, "x");
    const long B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    TORCH_CHECK(C % 8 == 0, "maxpool: C must be a multiple of 8");
    const long Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor y = empty_nhwc(B, C, Ho, Wo, x);
    Tensor arg = at::empty({B, Ho, Wo, C}, x.options().dtype(at::kByte));
    const long x_img = H * W * C * 2, y_img = Ho * Wo * C * 2;
    const long per = chunk_images(B, x_img);
    for (long b0 = 0; b0 < B; b0 += per)
        launch_maxpool3x3s2_fwd(at_image(x.data_ptr(), b0, x_img), at_image(y.data_ptr(), b0, y_img),
                                reinterpret_cast<uint8_t*>(at_image(arg.data_ptr(), b0, y_img / 2)),
                                std::min(per, B - b0), H, W, C, Ho, Wo, cur_stream(x));
    return {y, arg};
}
Tensor maxpool3x3s2_bwd(const Tensor& dy, const Tensor& arg, std::vector<int64_t> x_shape) {
    check_nhwc_bf16(dy, "dy");
    const long B = x_shape[0], C = x_shape[1], H = x_shape[2], W = x_shape[3];
    const long Ho = dy.size(2), Wo = dy.size(3);
    TORCH_CHECK(dy.size(0) == B && dy.size(1) == C && Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "maxpool bwd shape mismatch");
    TORCH_CHECK(arg.is_cuda() && arg.scalar_type() == at::kByte && arg.is_contiguous() && arg.numel() == dy.numel(),
                "argmax tensor mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dy.device());
    Tensor dx = empty_nhwc(B, C, H, W, dy);
    const long x_img = H * W * C * 2, y_img = Ho * Wo * C * 2;
    const long per = chunk_images(B, x_img);
    for (long b0 = 0; b0 < B; b0 += per)
        launch_maxpool3x3s2_bwd(at_image(dy.data_ptr(), b0, y_img),
                                reinterpret_cast<const uint8_t*>(at_image(arg.data_ptr(), b0, y_img / 2)),
                                at_image(dx.data_ptr(), b0, x_img), std::min(per, B - b0), H, W, C, Ho, Wo, cur_stream(dy));
    return dx;
}
// ---- stem tail fused: maxpool3x3/2/1(relu(bn(u))) --------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> bn_relu_maxpool_fwd(const Tensor& u, const Tensor& mean, const Tensor& invstd,
                                               const Tensor& gamma, const Tensor& beta) {
    check_nhwc_bf16(u, "u");
    const long B = u.size(0), C = u.size(1), H = u.size(2), W = u.size(3);
    TORCH_CHECK(bn_channels_ok(C), "bn_relu_maxpool: unsupported channel count ", C);
    check_vec_f32(mean, C, "mean"); check_vec_f32(invstd, C, "invstd");
    check_vec_f32(gamma, C, "gamma"); check_vec_f32(beta, C, "beta");
    const long Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    c10::hip::HIPGuardMasqueradingAsCUDA guard(u.device());
    Tensor y = empty_nhwc(B, C, Ho, Wo, u);
    Tensor arg = at::empty({B, Ho, Wo, C}, u.options().dtype(at::kByte));
    Tensor mask = at::empty({B * H * W, C / 8}, u.options().dtype(at::kByte));
    const long u_img = H * W * C * 2, y_img = Ho * Wo * C * 2;
    const long per = chunk_images(B, u_img);
    for (long b0 = 0; b0 < B; b0 += per)
        launch_bn_relu_maxpool_fwd(at_image(u.data_ptr(), b0, u_img), at_image(y.data_ptr(), b0, y_img),
                                   reinterpret_cast<uint8_t*>(at_image(arg.data_ptr(), b0, y_img / 2)),
                                   reinterpret_cast<uint8_t*>(at_image(mask.data_ptr(), b0, u_img / 16)),
                                   std::min(per, B - b0), H, W, C, Ho, Wo, mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                   gamma.data_ptr<float>(), beta.data_ptr<float>(), cur_stream(u));
    return {y, arg, mask};
}

Tensor global_avgpool_fwd(const Tensor& x) {
    check_nhwc_bf16(x, "x");
    const long B = x.size(0), C = x.size(1), HW = x.size(2) * x.size(3);
    TORCH_CHECK(C % 8 == 0, "avgpool: C must be a multiple of 8");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor y = at::empty({B, C}, x.options().memory_format(at::MemoryFormat::Contiguous));
    launch_global_avgpool_fwd(x.data_ptr(), y.data_ptr(), B, HW, C, cur_stream(x));
    return y;
}
Tensor global_avgpool_bwd(const Tensor& dy, std::vector<int64_t> x_shape) {
    TORCH_CHECK(dy.is_cuda() && dy.scalar_type() == at::kBFloat16 && dy.dim() == 2 && dy.is_contiguous(), "dy: expected contiguous bf16 [B,C]");
    const long B = x_shape[0], C = x_shape[1], H = x_shape[2], W = x_shape[3];
    TORCH_CHECK(dy.size(0) == B && dy.size(1) == C && C % 8 == 0, "avgpool bwd shape mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dy.device());
    Tensor dx = empty_nhwc(B, C, H, W, dy);
    launch_global_avgpool_bwd(dy.data_ptr(), dx.data_ptr(), B, H * W, C, cur_stream(dy));
    return dx;
}

// ------------------------------------------------------------------------------------- loss
std::tuple<Tensor, Tensor> softmax_ce(const Tensor& logits, const Tensor& labels) {
    TORCH_CHECK(logits.is_cuda() && logits.dim() == 2 && logits.is_contiguous(), "logits: expected contiguous GPU [B,N]");
    TORCH_CHECK(logits.scalar_type() == at::kBFloat16 || logits.scalar_type() == at::kFloat, "logits: bf16 or fp32");
    TORCH_CHECK(labels.is_cuda() && labels.scalar_type() == at::kLong && labels.is_contiguous() &&
                    labels.numel() == logits.size(0),
                "labels: expected contiguous int64 GPU [B]");
    const long B = logits.size(0), N = logits.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(logits.device());
    Tensor dlogits = at::empty_like(logits);
    Tensor row_loss = at::empty({B}, logits.options().dtype(at::kFloat));
    Tensor loss = at::empty({}, logits.options().dtype(at::kFloat));
    launch_softmax_ce(logits.data_ptr(), logits.scalar_type() == at::kBFloat16 ? kBF16 : kF32,
                      labels.data_ptr<int64_t>(), (int)B, (int)N, row_loss.data_ptr<float>(), dlogits.data_ptr(),
                      loss.data_ptr<float>(), cur_stream(logits));
    return {loss, dlogits};
}

// ------------------------------------------------------------------------------------- cast
Tensor to_nhwc_bf16(const Tensor& x, double scale) {
    TORCH_CHECK(x.is_cuda() && x.dim() == 4 && x.is_contiguous(), "x: expected contiguous 4-D GPU tensor");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    if (x.scalar_type() == at::kFloat) {  // NCHW fp32
        const long B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
        TORCH_CHECK((size_t)C * (W + 1) * 4 <= 64 * 1024, "to_nhwc_bf16: C*W too large for the LDS tile");
        Tensor y = at::empty({B, C, H, W}, x.options().dtype(at::kBFloat16).memory_format(at::MemoryFormat::ChannelsLast));
        launch_to_nhwc_bf16(x.data_ptr(), kF32, y.data_ptr(), B, C, H, W, (float)scale, cur_stream(x));
        return y;
    }
    TORCH_CHECK(x.scalar_type() == at::kByte, "to_nhwc_bf16: fp32 NCHW or uint8 NHWC input");
    const long B = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);  // NHWC uint8
    Tensor y = at::empty({B, C, H, W}, x.options().dtype(at::kBFloat16).memory_format(at::MemoryFormat::ChannelsLast));
    launch_to_nhwc_bf16(x.data_ptr(), kU8, y.data_ptr(), B, C, H, W, (float)scale, cur_stream(x));
    return y;
}

// ----------------------------------------------------------------------------- multi-tensor
struct Table {
    std::vector<int64_t> key;
    Tensor dev;
    int ntensors = 0, nchunks = 0;
};
std::unordered_map<uint64_t, Table> g_tables;
std::mutex g_tables_mu;  // the autograd thread (reducer hooks) and the main thread (optimizer) both use it

uint64_t hash_key(const std::vector<int64_t>& k) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t v : k) h = (h ^ (uint64_t)v) * 1099511628211ull;
    return h;
}

// cols: up to 5 pointer columns (nullptr allowed); returns a cached device table
const Table& get_table(const std::vector<std::vector<const Tensor*>>& cols, const std::vector<int64_t>& numels,
                       const at::Device& dev) {
    const size_t n = numels.size();
    std::vector<int64_t> key;
    key.reserve(n * (cols.size() + 1));
    for (size_t i = 0; i < n; ++i) {
        for (const auto& col : cols) key.push_back(col[i] ? (int64_t)(uintptr_t)col[i]->data_ptr() : 0);
        key.push_back(numels[i]);
    }
    const uint64_t h = hash_key(key);
    std::lock_guard<std::mutex> lock(g_tables_mu);
    auto it = g_tables.find(h);
    if (it != g_tables.end() && it->second.key == key) return it->second;
    if (g_tables.size() > 256) g_tables.clear();
    std::vector<int64_t> host;
    host.resize(n * kMTCols, 0);
    std::vector<int64_t> chunks;
    for (size_t i = 0; i < n; ++i) {
        for (size_t c = 0; c < cols.size(); ++c)
            host[i * kMTCols + c] = cols[c][i] ? (int64_t)(uintptr_t)cols[c][i]->data_ptr() : 0;
        host[i * kMTCols + 5] = numels[i];
        host[i * kMTCols + 6] = (int64_t)chunks.size();
        const int64_t nc = (numels[i] + kMTChunk - 1) / kMTChunk;
        for (int64_t c = 0; c < nc; ++c) chunks.push_back(((int64_t)i << 32) | c);
    }
    Table t;
    t.ntensors = (int)n;
    t.nchunks = (int)chunks.size();
    host.insert(host.end(), chunks.begin(), chunks.end());
    // uploaded through kernel arguments, not a host-to-device copy: a table can be missed inside a hipGraph capture
    // (gradients that are not bucket views are re-allocated from the graph's pool), where a copy from pageable
    // memory is not permitted
    t.dev = at::empty({(long)host.size()}, at::TensorOptions().dtype(at::kLong).device(dev));
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    for (size_t off = 0; off < host.size(); off += kTableWords) {
        TableWords tw{};
        const int cnt = (int)std::min<size_t>(kTableWords, host.size() - off);
        std::memcpy(tw.w, host.data() + off, cnt * sizeof(int64_t));
        launch_write_words(tw, cnt, t.dev.data_ptr<int64_t>() + off, stream);
    }
    t.key = std::move(key);
    auto& slot = g_tables[h];
    slot = std::move(t);
    return slot;
}

void check_mt(const Tensor& t, at::ScalarType st, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == st, name, ": wrong device/dtype");
    TORCH_CHECK(t.is_non_overlapping_and_dense(), name, ": tensor must be dense");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, ": data pointer must be 16-B aligned");
}

int dtype_code(at::ScalarType st) {
    TORCH_CHECK(st == at::kFloat || st == at::kBFloat16, "multi-tensor ops take fp32 or bf16");
    return st == at::kFloat ? kF32 : kBF16;
}

// dsts[i] <- srcs[i] * scale; every pair dense with identical strides.  Pointer table in kernel args.
void multi_flatten(std::vector<Tensor> srcs, std::vector<Tensor> dsts, double scale) {
    TORCH_CHECK(srcs.size() == dsts.size(), "multi_flatten: list size mismatch");
    if (srcs.empty()) return;
    const auto sst = srcs[0].scalar_type(), dst_t = dsts[0].scalar_type();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(srcs[0].device());
    hipStream_t stream = cur_stream(srcs[0]);
    FlatArgs a;
    a.n = 0;
    a.chunk_start[0] = 0;
    auto flush = [&]() {
        if (a.n > 0) launch_flatten_args(a, (float)scale, dtype_code(sst), dtype_code(dst_t), stream);
        a.n = 0;
        a.chunk_start[0] = 0;
    };
    for (size_t i = 0; i < srcs.size(); ++i) {
        check_mt(srcs[i], sst, "src");
        check_mt(dsts[i], dst_t, "dst");
        TORCH_CHECK(srcs[i].numel() == dsts[i].numel() && srcs[i].strides() == dsts[i].strides(),
                    "multi_flatten: src/dst must have the same dense layout");
        TORCH_CHECK(srcs[i].numel() < (1L << 31), "multi_flatten: tensor too large");
        if (srcs[i].numel() == 0) continue;
        a.src[a.n] = srcs[i].data_ptr();
        a.dst[a.n] = dsts[i].data_ptr();
        a.numel[a.n] = (int)srcs[i].numel();
        a.chunk_start[a.n + 1] = a.chunk_start[a.n] + (int)((srcs[i].numel() + kMTChunk - 1) / kMTChunk);
        if (++a.n == kFlatMax) flush();
    }
    flush();
}

void fill_cols(const std::vector<Tensor>& ps, const std::vector<Tensor>& gs, const std::vector<Tensor>* ms,
               const std::vector<Tensor>* vs, const std::vector<c10::optional<Tensor>>& lps,
               std::vector<std::vector<const Tensor*>>& cols, std::vector<int64_t>& numels) {
    const size_t n = ps.size();
    TORCH_CHECK(gs.size() == n && lps.size() == n && (!ms || ms->size() == n) && (!vs || vs->size() == n),
                "optimizer lists differ in length");
    cols.assign(5, std::vector<const Tensor*>(n, nullptr));
    const auto gst = gs[0].scalar_type();
    for (size_t i = 0; i < n; ++i) {
        check_mt(ps[i], at::kFloat, "param");
        check_mt(gs[i], gst, "grad");
        TORCH_CHECK(gs[i].numel() == ps[i].numel() && gs[i].strides() == ps[i].strides(), "grad layout != param layout");
        cols[0][i] = &ps[i];
        cols[1][i] = &gs[i];
        if (ms) {
            check_mt((*ms)[i], at::kFloat, "state");
            TORCH_CHECK((*ms)[i].strides() == ps[i].strides(), "state layout != param layout");
            cols[2][i] = &(*ms)[i];
        }
        if (vs) {
            check_mt((*vs)[i], at::kFloat, "state");
            TORCH_CHECK((*vs)[i].strides() == ps[i].strides(), "state layout != param layout");
            cols[3][i] = &(*vs)[i];
        }
        if (lps[i].has_value() && lps[i]->defined()) {
            check_mt(*lps[i], at::kBFloat16, "low-precision copy");
            TORCH_CHECK(lps[i]->strides() == ps[i].strides(), "low-precision copy layout != param layout");
            cols[4][i] = &*lps[i];
        }
        numels.push_back(ps[i].numel());
    }
}

void multi_adam(std::vector<Tensor> ps, std::vector<Tensor> gs, std::vector<Tensor> ms, std::vector<Tensor> vs,
                std::vector<c10::optional<Tensor>> lps, double lr, double beta1, double beta2, double eps,
                double weight_decay, int64_t step, bool adamw, double grad_scale,
                const c10::optional<Tensor>& step_dev) {
    if (ps.empty()) return;
    const float* step_ptr = nullptr;
    if (step_dev.has_value() && step_dev->defined()) {
        TORCH_CHECK(step_dev->is_cuda() && step_dev->scalar_type() == at::kFloat && step_dev->numel() == 1,
                    "step_dev: expected a 1-element fp32 GPU tensor");
        step_ptr = step_dev->data_ptr<float>();
    }
    std::vector<std::vector<const Tensor*>> cols;
    std::vector<int64_t> numels;
    fill_cols(ps, gs, &ms, &vs, lps, cols, numels);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(ps[0].device());
    const Table& t = get_table(cols, numels, ps[0].device());
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    launch_multi_adam(t.dev.data_ptr<int64_t>(), t.ntensors, t.nchunks, (float)lr, (float)beta1, (float)beta2,
                      (float)eps, (float)weight_decay, (float)bc1, (float)std::sqrt(bc2), step_ptr, adamw ? 1 : 0,
                      (float)grad_scale, dtype_code(gs[0].scalar_type()), cur_stream(ps[0]));
}

void multi_sgd(std::vector<Tensor> ps, std::vector<Tensor> gs, std::vector<Tensor> bufs,
               std::vector<c10::optional<Tensor>> lps, double lr, double momentum, double dampening,
               double weight_decay, bool nesterov, bool first, double grad_scale) {
    if (ps.empty()) return;
    std::vector<std::vector<const Tensor*>> cols;
    std::vector<int64_t> numels;
    fill_cols(ps, gs, &bufs, nullptr, lps, cols, numels);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(ps[0].device());
    const Table& t = get_table(cols, numels, ps[0].device());
    launch_multi_sgd(t.dev.data_ptr<int64_t>(), t.ntensors, t.nchunks, (float)lr, (float)momentum, (float)dampening,
                     (float)weight_decay, nesterov ? 1 : 0, first ? 1 : 0, (float)grad_scale,
                     dtype_code(gs[0].scalar_type()), cur_stream(ps[0]));
}

}  // namespace

void multi_flatten_tensors(std::vector<at::Tensor>& srcs, std::vector<at::Tensor>& dsts, double scale) {
    multi_flatten(srcs, dsts, scale);
}

// fp32 column sums of a row-major bf16 matrix [M, N] (Linear bias gradients): one streaming pass + slab reduce
Tensor colsum(const Tensor& x) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kBFloat16 && x.dim() == 2 && x.is_contiguous(),
                "colsum: expected a contiguous 2-D bf16 GPU tensor");
    const long M = x.size(0), N = x.size(1);
    TORCH_CHECK(N % 8 == 0 && N < (1L << 24) && reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 == 0,
                "colsum: columns must be a multiple of 8 and the data 16-B aligned");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    if (M == 0) return at::zeros({N}, x.options().dtype(at::kFloat));
    constexpr int kMaxChunks = 256;
    Tensor slab = at::empty({kMaxChunks, N}, x.options().dtype(at::kFloat));
    const int chunks = launch_colsum_bf16(x.data_ptr(), M, (int)N, slab.data_ptr<float>(), kMaxChunks, cur_stream(x));
    return reduce_slab(slab, chunks, N);
}

// ------------------------------------------------------------------------------------- gemm
void check_mat_bf16(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kBFloat16 && t.dim() == 2, name, ": expected a 2-D bf16 GPU tensor");
    TORCH_CHECK(t.stride(1) == 1 && t.stride(0) >= t.size(1) && t.stride(0) % 8 == 0 && t.size(1) % 8 == 0, name,
                ": rows must be contiguous, with length and pitch multiples of 8 elements");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, ": data pointer must be 16-B aligned");
    TORCH_CHECK(t.stride(0) < (1L << 22), name, ": row pitch too large");
}

// c = epilogue(a @ b^T): a [M,K], b [N,K] bf16 (K contiguous).  bias fp32 [N] (added before rounding);
// act 1 = GELU (erf); residual [M,N] bf16 added last; gelu_in [M,N]: result multiplied by GELU'(gelu_in);
// want_preact: also return the value before the activation; want_colsum: fp32 column sums of the result.
std::tuple<Tensor, Tensor, Tensor> gemm_nt(const Tensor& a, const Tensor& b, const c10::optional<Tensor>& bias,
                                           const c10::optional<Tensor>& residual, const c10::optional<Tensor>& gelu_in,
                                           int64_t act, bool want_preact, bool want_colsum) {
    check_mat_bf16(a, "a");
    check_mat_bf16(b, "b");
    const long M = a.size(0), K = a.size(1), N = b.size(0);
    TORCH_CHECK(b.size(1) == K, "gemm_nt: a is [", M, ",", K, "], b is [", N, ",", b.size(1), "]");
    TORCH_CHECK(N % 8 == 0 && N < (1L << 24) && K < (1L << 24), "gemm_nt: N must be a multiple of 8");
    TORCH_CHECK(act == 0 || act == 1, "gemm_nt: act is 0 (none) or 1 (GELU)");
    TORCH_CHECK(!want_preact || act == 1, "gemm_nt: want_preact needs an activation");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
    Tensor c = at::empty({M, N}, a.options());
    Tensor preact, colsum_out;
    GemmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.A = a.data_ptr(); g.B = b.data_ptr(); g.C = c.data_ptr();
    g.M = M; g.N = (int)N; g.K = (int)K;
    g.lda = (int)a.stride(0); g.ldb = (int)b.stride(0); g.ldc = (int)N;
    g.act = (int)act;
    if (bias.has_value() && bias->defined()) {
        check_vec_f32(*bias, N, "bias");
        g.bias = bias->data_ptr<float>();
    }
    if (residual.has_value() && residual->defined()) {
        check_mat_bf16(*residual, "residual");
        TORCH_CHECK(residual->size(0) == M && residual->size(1) == N, "residual must be [M,N]");
        g.residual = residual->data_ptr();
        g.ldr = (int)residual->stride(0);
    }
    if (gelu_in.has_value() && gelu_in->defined()) {
        check_mat_bf16(*gelu_in, "gelu_in");
        TORCH_CHECK(gelu_in->size(0) == M && gelu_in->size(1) == N, "gelu_in must be [M,N]");
        g.gelu_in = gelu_in->data_ptr();
        g.ldu = (int)gelu_in->stride(0);
    }
    if (want_preact) {
        preact = at::empty({M, N}, a.options());
        g.preact_out = preact.data_ptr();
    }
    Tensor slab;
    const int m_tiles = gemm_m_tiles(M);
    if (want_colsum) {
        slab = at::empty({std::max(m_tiles, 1), N}, a.options().dtype(at::kFloat));
        g.colsum_slab = slab.data_ptr<float>();
    }
    if (M > 0) launch_gemm_nt(g, cur_stream(a));
    if (want_colsum) {
        if (M == 0) colsum_out = at::zeros({N}, a.options().dtype(at::kFloat));
        else colsum_out = m_tiles == 1 ? slab.view({N}) : reduce_slab(slab, m_tiles, N);
    }
    return {c, preact, colsum_out};
}

// dW [N,K] fp32 = dy[M,N]^T @ x[M,K]: the weight-gradient kernel of a 1x1 convolution over M "pixels"
// (transposing LDS reads, deterministic split-K slab)
Tensor linear_wgrad(const Tensor& dy, const Tensor& x) {
    check_mat_bf16(dy, "dy");
    check_mat_bf16(x, "x");
    TORCH_CHECK(dy.is_contiguous() && x.is_contiguous(), "linear_wgrad: contiguous operands expected");
    const long M = x.size(0), K = x.size(1), N = dy.size(1);
    TORCH_CHECK(dy.size(0) == M, "linear_wgrad: row count mismatch");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    if (M == 0) return at::zeros({N, K}, x.options().dtype(at::kFloat));
    // k_conv_wgrad masks ragged channel tiles with the offset kOOB = 2^31 on a descriptor over the whole operand:
    // that offset is out of range (reads as zero) only while the operand a launch sees is smaller than 2 GiB
    // (common.h).  Longer token axes (ViT at a batch sized for 288 GB) go in equal row chunks, each a group of
    // split-K rows of one slab, like the batch chunks of conv2d_wgrad.
    const long window = g_window_bytes;
    const long row_bytes = std::max(K, N) * 2;
    TORCH_CHECK(row_bytes <= window, "linear_wgrad: one row exceeds the activation window");
    const long max_rows = std::max(1L, window / row_bytes);
    const long nchunks = (M + max_rows - 1) / max_rows;
    long per = (M + nchunks - 1) / nchunks;
    if (nchunks > 1) per = std::min(max_rows, (per + 255) / 256 * 256);
    std::vector<WgradGeom> geoms;
    long rows = 0;
    for (long m0 = 0; m0 < M; m0 += per) {
        WgradGeom g{(int)std::min(per, M - m0), 1, 1, (int)K, 1, 1, (int)N, 1, 1, 1, 0, 1};
        g.splits = conv_wgrad_splits(g);
        // (gemm_tn.hip is one workgroup per CU per launch: every chunk keeps its full split count.  ViT-B/16 at
        // B = 2560, two chunks: 135-144 workgroups per launch and 722 TFLOP/s before, see profiles/r3_attention_pipelined.md)
        if (nchunks > 1 && !gemm_tn_ok(g)) g.splits = std::max<int>(1, (int)((g.splits + nchunks - 1) / nchunks));
        geoms.push_back(g);
        rows += g.splits;
    }
    const long wsize = N * K;
    Tensor slab = at::empty({rows, wsize}, x.options().dtype(at::kFloat));
    long row = 0, ci = 0;
    for (long m0 = 0; m0 < M; m0 += per, ++ci) {
        launch_conv_wgrad(static_cast<const char*>(dy.data_ptr()) + m0 * N * 2,
                          static_cast<const char*>(x.data_ptr()) + m0 * K * 2, slab.data_ptr<float>() + row * wsize,
                          geoms[ci], cur_stream(x));
        row += geoms[ci].splits;
    }
    Tensor flat = rows == 1 ? slab.view({wsize}) : reduce_slab(slab, (int)rows, wsize);
    return flat.view({N, K});
}

// ------------------------------------------------------------------------------------- vit glue
// x [B,3,H,W] channels_last bf16 -> patches [B*G*G, P*P*3] in (py, px, c) order
Tensor patchify(const Tensor& x, int64_t P) {
    check_nhwc_bf16(x, "x");
    const long B = x.size(0), H = x.size(2), W = x.size(3);
    TORCH_CHECK(x.size(1) == 3 && H == W && P > 0 && P % 8 == 0 && H % P == 0, "patchify: [B,3,S,S] image, S % P == 0, P % 8 == 0");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const long G = H / P;
    Tensor out = at::empty({B * G * G, P * P * 3}, x.options().memory_format(at::MemoryFormat::Contiguous));
    launch_patchify(x.data_ptr(), out.data_ptr(), B, (int)G, (int)P, cur_stream(x));
    return out;
}

Tensor vit_embed_fwd(const Tensor& patch, const Tensor& cls, const Tensor& pos) {
    TORCH_CHECK(patch.is_cuda() && patch.scalar_type() == at::kBFloat16 && patch.dim() == 3 && patch.is_contiguous(),
                "patch: contiguous bf16 [B,T,D]");
    const long B = patch.size(0), T = patch.size(1), D = patch.size(2);
    TORCH_CHECK(D % 8 == 0 && reinterpret_cast<uintptr_t>(patch.data_ptr()) % 16 == 0, "patch: D % 8 == 0, 16-B aligned");
    check_vec_f32(cls, D, "class_token");
    check_vec_f32(pos, (T + 1) * D, "pos_embedding");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(patch.device());
    Tensor out = at::empty({B, T + 1, D}, patch.options());
    launch_vit_embed_fwd(patch.data_ptr(), cls.data_ptr<float>(), pos.data_ptr<float>(), out.data_ptr(), B, (int)T, (int)D,
                         cur_stream(patch));
    return out;
}

// dx [B,T+1,D] -> (dpatch [B,T,D] bf16, dpos [T+1,D] fp32)
std::tuple<Tensor, Tensor> vit_embed_bwd(const Tensor& dx) {
    TORCH_CHECK(dx.is_cuda() && dx.scalar_type() == at::kBFloat16 && dx.dim() == 3 && dx.is_contiguous() && dx.size(1) >= 1,
                "dx: contiguous bf16 [B,T+1,D]");
    const long B = dx.size(0), T = dx.size(1) - 1, D = dx.size(2);
    TORCH_CHECK(D % 8 == 0 && reinterpret_cast<uintptr_t>(dx.data_ptr()) % 16 == 0 && B < (1L << 31), "dx: D % 8 == 0, 16-B aligned");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dx.device());
    Tensor dpatch = at::empty({B, T, D}, dx.options());
    const int slices = (int)std::max<long>(1, std::min<long>(16, B / 8));
    Tensor slab = at::empty({slices, (T + 1) * D}, dx.options().dtype(at::kFloat));
    if (B == 0) return {dpatch, at::zeros({T + 1, D}, slab.options())};
    launch_vit_embed_bwd(dx.data_ptr(), dpatch.data_ptr(), slab.data_ptr<float>(), (int)B, (int)T, (int)D, slices,
                         cur_stream(dx));
    Tensor dpos = slices == 1 ? slab.view({T + 1, D}) : reduce_slab(slab, slices, (T + 1) * D).view({T + 1, D});
    return {dpatch, dpos};
}

void bind_ops(pybind11::module& m) {
    m.def("patchify", &patchify);
    m.def("vit_embed_fwd", &vit_embed_fwd);
    m.def("vit_embed_bwd", &vit_embed_bwd);
    m.def("gemm_nt", &gemm_nt, pybind11::arg("a"), pybind11::arg("b"), pybind11::arg("bias") = pybind11::none(),
          pybind11::arg("residual") = pybind11::none(), pybind11::arg("gelu_in") = pybind11::none(),
          pybind11::arg("act") = 0, pybind11::arg("want_preact") = false, pybind11::arg("want_colsum") = false);
    m.def("set_attention_pipelined", &set_attention_pipelined, "whole-head attention of 5+ tiles on the persistent pipelined kernels (default on)");
    m.def("set_gemm_tile_width", &set_gemm_tile_width, "force the GEMM core's tile width (128 / 256; 0: by shape)");
    m.def("linear_wgrad", &linear_wgrad);
    m.def("conv2d_fwd", &conv2d_fwd, pybind11::arg("x"), pybind11::arg("w"), pybind11::arg("stride"), pybind11::arg("pad"),
          pybind11::arg("want_stats"), pybind11::arg("cfg") = 0);
    m.def("conv2d_dgrad", &conv2d_dgrad, pybind11::arg("dy"), pybind11::arg("wt"), pybind11::arg("x_shape"), pybind11::arg("kh"),
          pybind11::arg("kw"), pybind11::arg("stride"), pybind11::arg("pad"), pybind11::arg("addend") = pybind11::none(),
          pybind11::arg("bn_x") = pybind11::none(), pybind11::arg("bn_mask") = pybind11::none(),
          pybind11::arg("bn_mean") = pybind11::none(), pybind11::arg("bn_invstd") = pybind11::none(),
          pybind11::arg("addend_mask") = pybind11::none(), pybind11::arg("addend_stride") = 1, pybind11::arg("cfg") = 0,
          pybind11::arg("bn2_x") = pybind11::none(), pybind11::arg("bn2_mean") = pybind11::none(),
          pybind11::arg("bn2_invstd") = pybind11::none());
    m.def("conv2d_wgrad", &conv2d_wgrad, pybind11::arg("dy"), pybind11::arg("x"), pybind11::arg("kh"), pybind11::arg("kw"),
          pybind11::arg("stride"), pybind11::arg("pad"), pybind11::arg("splits") = 0);
    m.def("conv_stem_fwd", &conv_stem_fwd);
    m.def("set_stem_win_mode", &set_stem_win_mode);
    m.def("set_wgrad_win64_mode", &set_wgrad_win64_mode);
    m.def("stem_pack_x", &stem_pack_x);
    m.def("conv_stem_wgrad", &conv_stem_wgrad);
    m.def("transpose_weight", &transpose_weight);
    m.def("transpose_weights", &transpose_weights, pybind11::arg("weights"), pybind11::arg("outs") = pybind11::none());
    m.def("colsum", &colsum);
    m.def("bn_stats", &bn_stats);
    m.def("bn_finalize", &bn_finalize);
    m.def("bn_sum_partials", &bn_sum_partials);
    m.def("bn_apply_act", &bn_apply_act, pybind11::arg("x"), pybind11::arg("mean"), pybind11::arg("invstd"),
          pybind11::arg("gamma"), pybind11::arg("beta"), pybind11::arg("residual"), pybind11::arg("relu"),
          pybind11::arg("res_affine") = pybind11::none());
    m.def("bn_bwd_reduce", &bn_bwd_reduce);
    m.def("bn_bwd_elemt", &bn_bwd_elemt);
    m.def("bn_bwd_elemt2", &bn_bwd_elemt2);
    m.def("bn_relu_maxpool_bwd", &bn_relu_maxpool_bwd);
    m.def("attention_fwd", &attention_fwd);
    m.def("attention_bwd", &attention_bwd);
    m.def("layernorm_fwd", &layernorm_fwd);
    m.def("layernorm_bwd", &layernorm_bwd, pybind11::arg("dy"), pybind11::arg("x"), pybind11::arg("mean"),
          pybind11::arg("rstd"), pybind11::arg("gamma"), pybind11::arg("addend") = pybind11::none());
    m.def("set_wgrad_tn_min_rows", [](int64_t rows) { set_gemm_tn_min_rows((long)rows); },
          "reduction length (output pixels / tokens) from which wide layers use the 256x256 weight-gradient kernel");
    m.def("set_activation_window", &set_activation_window,
          "bytes of one operand a convolution / pooling launch may span (tests shrink it to force batch chunking)");
    m.def("maxpool3x3s2_fwd", &maxpool3x3s2_fwd);
    m.def("maxpool3x3s2_bwd", &maxpool3x3s2_bwd);
    m.def("bn_relu_maxpool_fwd", &bn_relu_maxpool_fwd);
    m.def("global_avgpool_fwd", &global_avgpool_fwd);
    m.def("global_avgpool_bwd", &global_avgpool_bwd);
    m.def("softmax_ce", &softmax_ce);
    m.def("to_nhwc_bf16", &to_nhwc_bf16);
    m.def("multi_flatten", &multi_flatten);
    m.def("multi_adam", &multi_adam, pybind11::arg("params"), pybind11::arg("grads"), pybind11::arg("exp_avg"),
          pybind11::arg("exp_avg_sq"), pybind11::arg("lowp"), pybind11::arg("lr"), pybind11::arg("beta1"),
          pybind11::arg("beta2"), pybind11::arg("eps"), pybind11::arg("weight_decay"), pybind11::arg("step"),
          pybind11::arg("adamw"), pybind11::arg("grad_scale"), pybind11::arg("step_dev") = pybind11::none());
    m.def("multi_sgd", &multi_sgd);
}

}  // namespace pdt
