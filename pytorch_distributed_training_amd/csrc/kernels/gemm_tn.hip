// Weight gradients of wide layers on the matrix cores with the GEMM core's schedule:
//
//   dW[n][tap][c] = sum_m dY[m][n] * X[pix(m, tap)][c]          n, c in 256 x 256 tiles, m = output pixel / token
//
// Both operands keep the reduction index m as their ROW (NHWC / token-major), so the MFMA fragments are fetched with
// the CDNA4 transposing LDS read (ds_read_b64_tr_b16), as in conv_wgrad.hip.  What differs from that kernel (128 x
// 128 tiles, 4 waves, register-staged, 3 workgroups per CU: 620-780 TFLOP/s, and every dY / X byte crosses the
// L2 -> LDS path twice as often) is everything around the MFMAs, taken from gemm.hip:
//  * 256 x 256 output tile, one workgroup of 8 waves (2 x 4, 128 x 64 each, 128 accumulator registers) per CU;
//  * operands arrive by LDS-DMA into a ring of 4 stages of 32 reduction rows ([32][256] bf16 per operand, 16 KiB),
//    swizzle applied to the SOURCE address, retired by a counted s_waitcnt vmcnt(6): the X rows of K-tile t+2 are
//    issued in the first phase of K-tile t, the dY rows of t+3 in the second, so 2-3 K-tiles are always in flight
//    (past the end of a split the issues go on, out of range: zeros into stages nobody reads, uniform counting);
//  * waves 0-3 and 4-7 run one phase apart: one group issues its 16 MFMAs while its SIMD partners issue fragment
//    reads and DMA;
//  * split-K over m across the CUs: (tiles x splits) ~ one workgroup per CU; every split writes its fp32 tile into
//    the slab [split][N][taps][C] (same layout as conv_wgrad.hip: one deterministic reduce_slab afterwards).
// Spatial filters and strided layers: the tap is part of the tile index, the X row of an output pixel is tracked as
// (ho, wo, byte offset) per staged row and advanced by the constant decomposition of 32 pixels, as in conv_wgrad.hip;
// padding and the ragged end of a split are DMA lanes sent out of range (zeros).
// Measured at B = 1024 (bench/micro/wgrad_shapes.py, profiles/r2_wgrad_shapes_b1024.log): 256-ch 3x3 347 -> 287 us
// (825 TFLOP/s), 512->2048 1x1 154 -> 115 us, ViT 3072x768 177 -> 117 us (1020 TFLOP/s).  With the DMA switched off
// the loop runs at 1300 TFLOP/s, with the fragment reads off nothing changes, a deeper ring is slower and 6x less
// fabric traffic (XCD grouping) does not help either: what bounds it is the LDS itself, which serves the
// transposing reads (8 B per lane) and the DMA writes of 8 waves at once.
// Used when Cout and Cin are multiples of 256 (ResNet-50: 26 of its 53 weight gradients; every ViT-B/16 linear layer).
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

constexpr int TKM = 32;                    // reduction rows per K-tile
constexpr int TPF = 2;                     // prefetch distance of the X operand in K-tiles (dY: one more)
constexpr int TSTAGES = TPF + 2;
constexpr int OPER_BYTES = TKM * 512;      // [32][256] bf16
constexpr int TSTAGE_BYTES = 2 * OPER_BYTES;
constexpr int TN_LDS = TSTAGES * TSTAGE_BYTES;  // 128 KiB (a 5-stage ring, TPF = 3, measured 10-15 % slower: the
                                                // operand path is throughput-bound, more bytes in flight only queue)
static_assert(TN_LDS <= 163840, "ring fits the LDS");
__device__ __forceinline__ void wait_ring() {  // everything but the youngest 2 * TPF - 1 issues (2 DMA each) has landed
    if constexpr (TPF == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}
static_assert(TPF == 2 || TPF == 3, "wait_ring knows these two depths");

// 16-B chunk swizzle of a 512-B row: the 16 lanes of one transposed read touch 4 consecutive rows x 32 B, which
// must fall into 4 different 32-B granules of the 256-B bank period; rows 8 apart (the other lane groups of the
// instruction) go to the other half
__device__ __forceinline__ int row_key(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

struct TnArgs {
    const void* dy;
    const void* x;
    float* slab;
    long M, m_per_split;
    int N, C, T, KW;
    int Hin, Win, Hout, Wout, stride, pad;
    int n_tiles, c_tiles, splits;
    FastDiv div_wo, div_howo;
    int kh, kw;  // 32 pixels = kb images + kh rows + kw columns of the output grid
    uint32_t xstep, xwrap_w, xwrap_h, x_bytes;
    int xcd_remap;
    int dy_stream, x_stream;  // pointwise layer, operand read by ONE tile column / row: non-temporal DMA
};

// fragment of MFMA 16x16x32 from a row-major [k][col] tile: lane l gets (col = colbase + l % 16, k = 8 * (l / 16) + 0..7)
__device__ __forceinline__ bf16x8 tr_frag16(const char* tile, int lane_off, int const_chunk_xor) {
    const int o = lane_off ^ (const_chunk_xor << 4);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + o));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + o + 4 * 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool LINEAR>
__global__ __launch_bounds__(512, 2) void k_gemm_tn(const TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2): give every XCD a
    // CONTIGUOUS range of logical ids, so that the tiles and taps of one split -- which stream the same dY rows and
    // (shifted) X rows -- run on one XCD at the same time and fetch those rows from HBM once, not once per XCD
    int bid = blockIdx.x;
    if (a.xcd_remap) {
        const int total = (int)gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q8 = total >> 3, r8 = total & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    // order: tiles, then the taps of one filter ROW, then the splits, then the filter rows.  With the contiguous XCD
    // ranges the KW taps of a row (and the tiles) of a split share an XCD: they stream the same dY rows and X rows
    // one pixel apart, so those come over the fabric once per filter row instead of once per tap (FETCH_SIZE of the
    // 256-channel 3x3 layer: 6.2x less with all 9 taps on one XCD, but then 9 workgroups hammer the same L2 lines
    // at the same moment and the kernel is no faster; 3 is the middle)
    const int c_tile = bid % a.c_tiles;
    bid /= a.c_tiles;
    const int n_tile = bid % a.n_tiles;
    bid /= a.n_tiles;
    const int tcol = bid % a.KW;
    bid /= a.KW;
    const int split = bid % a.splits;
    const int trow = bid / a.splits;
    const int tap = trow * a.KW + tcol;
    const int r = tap / a.KW, s = tap - r * a.KW;
    const long m_begin = (long)split * a.m_per_split;
    long m_end = m_begin + a.m_per_split;
    if (m_end > a.M) m_end = a.M;
    const int span = (int)(m_end - m_begin);
    const int nkt = (span + TKM - 1) / TKM;
    const int n0 = n_tile * 256, c0 = c_tile * 256;
    if (span <= 0) {  // a split past the end (forced split counts): its slab tile is zeros
        float* out = a.slab + (long)split * a.N * a.T * a.C;
        for (int e = tid; e < 256 * 64; e += 512) {
            const int n = n0 + (e >> 6), c = c0 + (e & 63) * 4;
            *reinterpret_cast<f32x4_t*>(out + ((long)n * a.T + tap) * a.C + c) = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // ---- DMA: per K-tile and operand each thread fetches 2 pieces (j) of 1 KiB = 2 rows x 512 B; lane -> row
    // piece * 2 + lane / 32, chunk position lane % 32, which holds logical chunk (lane % 32) ^ row_key(row)
    const i32x4 rdy = make_rsrc_words(static_cast<const char*>(a.dy) + (m_begin * a.N + n0) * 2,
                                      (uint32_t)((long)(span - 1) * a.N * 2 + 512));
    i32x4 rx;
    uint32_t vdy[2], vx[2];
    int xho[2], xwo[2], rowj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 2 + (lane >> 5);
        const int lch = (lane & 31) ^ row_key(row);
        rowj[j] = row;
        vdy[j] = (uint32_t)row * (uint32_t)(a.N * 2) + (uint32_t)(lch * 16);
        if (LINEAR) {
            vx[j] = (uint32_t)row * (uint32_t)(a.C * 2) + (uint32_t)(lch * 16);
        } else {
            const uint32_t mm = (uint32_t)(m_begin + row);
            const uint32_t b = fdiv(mm, a.div_howo);
            const uint32_t rem = mm - b * a.div_howo.d;
            const uint32_t ho = fdiv(rem, a.div_wo);
            const uint32_t wo = rem - ho * a.div_wo.d;
            xho[j] = (int)ho;
            xwo[j] = (int)wo;
            const int hi = (int)ho * a.stride - a.pad + r, wi = (int)wo * a.stride - a.pad + s;
            vx[j] = (uint32_t)((((int)b * a.Hin + hi) * a.Win + wi) * a.C + c0 + lch * 8) * 2u;  // modular when hi / wi < 0
        }
    }
    if (LINEAR) rx = make_rsrc_words(static_cast<const char*>(a.x) + (m_begin * a.C + c0) * 2, (uint32_t)((long)(span - 1) * a.C * 2 + 512));
    else rx = make_rsrc_words(a.x, a.x_bytes);
    const int dr = r - a.pad, ds = s - a.pad;
    const uint32_t piece_base = lds_address(smem) + (uint32_t)wave * 2048u;

    auto issue_dy = [&](int tt) {
        const uint32_t dst = piece_base + (uint32_t)((tt % TSTAGES) * TSTAGE_BYTES);
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(tt * TKM * a.N * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (a.dy_stream) lds_dma16s_stream(rdy, vdy[j], soff, dst + (uint32_t)(j * 1024));
            else lds_dma16s(rdy, vdy[j], soff, dst + (uint32_t)(j * 1024));
        }
    };
    auto issue_x = [&](int tt) {
        const uint32_t dst = piece_base + (uint32_t)((tt % TSTAGES) * TSTAGE_BYTES + OPER_BYTES);
        if (LINEAR) {
            const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(tt * TKM * a.C * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (a.x_stream) lds_dma16s_stream(rx, vx[j], soff, dst + (uint32_t)(j * 1024));
                else lds_dma16s(rx, vx[j], soff, dst + (uint32_t)(j * 1024));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hi = __mul24(xho[j], a.stride) + dr, wi = __mul24(xwo[j], a.stride) + ds;
                const bool ok = (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win && tt * TKM + rowj[j] < span;
                lds_dma16s(rx, ok ? vx[j] : kOOB, 0u, dst + (uint32_t)(j * 1024));
                // advance this row by 32 pixels
                int wo = xwo[j] + a.kw;
                const bool cw = wo >= a.Wout;
                wo -= cw ? a.Wout : 0;
                int ho = xho[j] + a.kh + (cw ? 1 : 0);
                const bool ch = ho >= a.Hout;
                ho -= ch ? a.Hout : 0;
                xwo[j] = wo;
                xho[j] = ho;
                vx[j] += a.xstep + (cw ? a.xwrap_w : 0u) + (ch ? a.xwrap_h : 0u);
            }
        }
    };

    // ---- fragment read offsets: lane group g = lane / 16 reads rows 8g + q (and + 4), q = (lane % 16) / 4, columns
    // base + 4 * (lane % 4); the column chunk is (lane part) | (compile-time part), disjoint bits
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int frow = 8 * g + q;
    const int key = row_key(frow);
    const int a_lane = frow * 512 + ((((wr << 4) | (p >> 1)) ^ key) << 4) + (p & 1) * 8;       // + const chunk mh*8 + i*2
    const int b_lane = OPER_BYTES + frow * 512 + ((((wc << 3) | (p >> 1)) ^ key) << 4) + (p & 1) * 8;  // + const chunk j*2

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: what the steady state would have issued before K-tile 0, in its order
#pragma unroll
    for (int i = 0; i < TPF; ++i) {
        issue_dy(i);
        issue_x(i);
    }
    issue_dy(TPF);
    wait_ring();
    __builtin_amdgcn_s_barrier();

    if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one phase behind their SIMD partners
    bf16x8 bfrag[4];
    for (int t = 0; t < nkt; ++t) {
        const char* st = smem + (t % TSTAGES) * TSTAGE_BYTES;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            // ---------------- load segment (the partner wave is in its MFMA segment)
            bf16x8 afrag[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) afrag[i] = tr_frag16(st, a_lane, mh * 8 + i * 2);
            if (mh == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bfrag[j] = tr_frag16(st, b_lane, j * 2);
                issue_x(t + TPF);  // (past the end of the split: out-of-range lanes, zeros into a stage nobody reads)
            } else {
                issue_dy(t + TPF + 1);
                wait_ring();  // everything of K-tile t+1 has landed
            }
            __builtin_amdgcn_s_barrier();
            // ---------------- MFMA segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[j], afrag[i], acc[mh * 4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMA of the last K-tiles must not outlive the block

    // ---- partial tile -> slab[split][n][tap][c]: lane holds n = i*16 + lane%16, c = j*16 + (lane/16)*4 + 0..3
    float* out = a.slab + (long)split * a.N * a.T * a.C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = n0 + wr * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<f32x4_t*>(out + ((long)n * a.T + tap) * a.C + c) = acc[i][j];
        }
    }
}

bool linear_layer(const WgradGeom& g) { return g.KH == 1 && g.KW == 1 && g.stride == 1 && g.pad == 0; }

bool tn_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("PDT_WGRAD_TN");
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}

}  // namespace

static long g_tn_min_m = [] {
    const char* e = std::getenv("PDT_WGRAD_TN_MIN_M");
    return e ? std::atol(e) : 8192L;
}();
void set_gemm_tn_min_rows(long rows) { g_tn_min_m = rows; }

bool gemm_tn_ok(const WgradGeom& g) {
    const long M = (long)g.B * g.Hout * g.Wout;
    // (short reductions -- the CIFAR-shaped step has 512 to 8192 output pixels per layer -- stay on the small-tile
    // kernel: one 256 x 256 fp32 slab tile per workgroup is then most of the traffic; PDT_WGRAD_TN_MIN_M overrides)
    return tn_enabled() && g.N % 256 == 0 && g.C % 256 == 0 && M >= g_tn_min_m && M >= 4 * TKM && M < (1L << 31) &&
           (long)g.B * g.Hin * g.Win * g.C * 2 < (1L << 31);
}

int gemm_tn_splits(const WgradGeom& g) {
    const long tiles = (long)(g.N / 256) * (g.C / 256) * g.KH * g.KW;
    static const long target = [] {
        const char* e = std::getenv("PDT_WGRAD_TN_BLOCKS");
        return e ? std::atol(e) : 256L;
    }();
    const long M = (long)g.B * g.Hout * g.Wout;
    long splits = target / tiles;
    const long max_splits = M / (4 * TKM);  // at least 4 K-tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits > 512) splits = 512;
    if (splits < 1) splits = 1;
    return (int)splits;
}

void launch_gemm_tn(const void* dy, const void* x, float* slab, const WgradGeom& g, hipStream_t s) {
    TnArgs a;
    a.dy = dy; a.x = x; a.slab = slab;
    static const int remap = [] {
        const char* e = std::getenv("PDT_WGRAD_TN_REMAP");
        return e ? std::atoi(e) : 1;
    }();
    // (Same-box A/B on the final tree: grouping also the pointwise layers with many tiles is +0.8 % on ViT-B/16 and
    // neutral on ResNet-50; an earlier "3 % slower" reading came from a launcher that used the tile counts before
    // setting them.)
    a.xcd_remap = remap ? 1 : 0;
    a.M = (long)g.B * g.Hout * g.Wout;
    a.N = g.N; a.C = g.C; a.T = g.KH * g.KW; a.KW = g.KW;
    a.Hin = g.Hin; a.Win = g.Win; a.Hout = g.Hout; a.Wout = g.Wout; a.stride = g.stride; a.pad = g.pad;
    a.n_tiles = g.N / 256; a.c_tiles = g.C / 256; a.splits = g.splits;
    static const bool stream_on = [] {
        const char* e = std::getenv("PDT_WGRAD_STREAM");
        return !(e && e[0] == '0');
    }();
    a.dy_stream = stream_on && linear_layer(g) && a.c_tiles == 1 ? 1 : 0;
    a.x_stream = stream_on && linear_layer(g) && a.n_tiles == 1 ? 1 : 0;
    long per = (a.M + g.splits - 1) / g.splits;
    per = (per + TKM - 1) / TKM * TKM;
    a.m_per_split = per;
    const long widest = g.N > g.C ? g.N : g.C;
    if (!gemm_tn_ok(g) || g.splits < 1 || (per + 4L * TKM) * widest * 2 >= (1L << 31)) {
        fprintf(stderr, "launch_gemm_tn: unsupported launch (%d splits of %ld rows, N = %d, C = %d)\n", g.splits, per, g.N, g.C);
        abort();
    }
    a.div_wo = make_fastdiv((uint32_t)g.Wout);
    a.div_howo = make_fastdiv((uint32_t)(g.Hout * g.Wout));
    a.x_bytes = (uint32_t)((long)g.B * g.Hin * g.Win * g.C * 2);
    const int hw = g.Hout * g.Wout;
    const int kb = TKM / hw, rem = TKM % hw;
    a.kh = rem / g.Wout;
    a.kw = rem % g.Wout;
    const uint32_t c2 = (uint32_t)g.C * 2u;
    a.xstep = ((uint32_t)kb * g.Hin * g.Win + (uint32_t)a.kh * g.stride * g.Win + (uint32_t)a.kw * g.stride) * c2;
    a.xwrap_w = ((uint32_t)g.stride * g.Win - (uint32_t)g.Wout * g.stride) * c2;
    a.xwrap_h = ((uint32_t)g.Hin * g.Win - (uint32_t)g.Hout * g.stride * g.Win) * c2;
    const bool linear = linear_layer(g);
    const int grid = a.n_tiles * a.c_tiles * a.T * g.splits;
    static bool attr_set = false;
    if (!attr_set) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS));
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS));
        attr_set = true;
    }
    if (linear) hipLaunchKernelGGL((k_gemm_tn<true>), dim3(grid), dim3(512), TN_LDS, s, a);
    else hipLaunchKernelGGL((k_gemm_tn<false>), dim3(grid), dim3(512), TN_LDS, s, a);
}

}  // namespace pdt
