// Convolution weight gradient on the gfx950 matrix cores.
//
//   dW[n][r][s][c] = sum_m dY[m][n] * X[pix(m, r, s)][c],   m = (b, ho, wo)
//
// GEMM view: rows = n (Cout), cols = c (Cin) per tap (r,s), and the reduction runs over the
// B*Ho*Wo output pixels: enormous K, small output.  Both operands are K-STRIDED in memory
// (NHWC keeps channels contiguous, not pixels), so tiles are staged row-major [pixel][channel]
// in LDS and the MFMA fragments are fetched with the CDNA4 transposing LDS read
// (ds_read_b64_tr_b16): no transposed copies of dY / X are ever materialised.
//
//  * block = 256 threads = 2x2 waves, output tile TN x TC (64/128 each) of one tap, MFMA
//    32x32x16 bf16 with fp32 accumulators.
//  * split-K over pixels: grid = tiles x splits; each split writes its fp32 partial tile to a
//    slab [split][N][taps][C] with plain stores, and a second launch sums the slab rows in a
//    fixed order: deterministic, no float atomics (which cap at ~1.3 TB/s chip-wide and lose
//    run-to-run reproducibility).
//  * LDS rows are 128/256 B with an XOR swizzle of the 16-B chunk chosen so that the four
//    rows x 64 B that one half-wave touches in a transposed read fall on 64 distinct banks.
//  * zero padding and the ragged end of a split come from the buffer range check.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int KM = 32;  // pixels (reduction rows) per K-step

template <int RB>
__device__ __forceinline__ int swz_off(int row, int ch) {
    if constexpr (RB == 256) {
        return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    } else {
        return row * 128 + ((ch ^ (((row >> 1) & 1) << 2)) << 4);
    }
}

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

// fragment for MFMA 32x32x16 from a row-major [k][col] LDS tile: lane l gets
// elements (k = 8*(l>>5) + 0..7, col = colbase + (l&31)).
template <int RB>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int krow0, int colbase, int lane) {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3;
    const int col = colbase + ((lane >> 4) & 1) * 16 + p * 4;
    const int row = krow0 + 8 * (lane >> 5) + q;
    const int o0 = swz_off<RB>(row, col >> 3) + (col & 7) * 2;
    const int o1 = swz_off<RB>(row + 4, col >> 3) + (col & 7) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + o0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + o1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct WgradArgs {
    int B, Hin, Win, C, Hout, Wout, N, KH, KW, stride, pad;
    int n_tiles, c_tiles, splits, xcd_remap;
    int dy_stream, x_stream;  // the operand is read by one tile column / row only: non-temporal loads
    long M, m_per_split;
    FastDiv div_wo, div_howo;
    uint32_t x_bytes, dy_bytes;
    // KM pixels = kb images + kh rows + kw columns (of the OUTPUT grid), and what that is in X bytes for a fixed tap:
    // xstep without carries, xwrap_w / xwrap_h the corrections when the column / row index wraps (mod 2^32)
    int kh, kw;
    uint32_t xstep, xwrap_w, xwrap_h;
};

// (An LDS-DMA ring variant of this kernel -- 3 or 4 stages, source-side swizzle, counted vmcnt as in conv_igemm.hip --
// passed all numerics and measured 14.55 / 14.77 ms against 14.28 ms for this register-staged double buffer on
// the sum of ResNet-50's weight gradients at B=1024, so it was not kept.)
// (Round 2: 256 x 128 and 128 x 256 tiles -- 128 x 64 per wave, 1.5 transposed LDS reads per MFMA instead of 2, 2
// workgroups per CU -- passed the tests and measured 8-12 % SLOWER on every ResNet-50 layer with >= 256 channels and
// on ViT-B/16's linear layers at B = 1024 (bench/micro/wgrad_shapes.py: 256->256 3x3 347 -> 381 us, 512->2048 1x1
// 154 -> 176 us, linear 768x2304 145 -> 173 us): with register-staged operands the kernel lives on its 12 waves
// per CU hiding the global-load latency, not on LDS bandwidth.  Not kept.)
template <int TNW, int TCW>
__global__ __launch_bounds__(256, 3) void k_conv_wgrad(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                       float* __restrict__ slab, const WgradArgs a) {
    constexpr int RBN = TNW * 2, RBC = TCW * 2;        // LDS row bytes
    constexpr int CHN = TNW / 8, CHC = TCW / 8;        // 16-B chunks per row
    constexpr int RPN = 256 / CHN, RPC = 256 / CHC;    // rows per staging pass
    constexpr int PN = KM / RPN, PC = KM / RPC;        // passes per K-step
    constexpr int TM = TNW / 64, TC = TCW / 64;        // 32x32 MFMA tiles per wave (2x2 waves)
    constexpr int N_BYTES = KM * RBN, C_BYTES = KM * RBC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ns = smem;                 // [2][KM][RBN]  dY tile
    char* Cs = smem + 2 * N_BYTES;   // [2][KM][RBC]  X tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;

    // block -> (split, tap, n tile, c tile); tiles of one split are adjacent so the dY / X rows
    // they share are hot in L2
    // XCD-aware bijective remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with
    // its own L2), so give every XCD a CONTIGUOUS range of logical ids: the tiles and taps of one split then run
    // on one XCD at about the same time and share their dY / X rows in that L2 instead of re-fetching them
    int bid = blockIdx.x;
    if (a.xcd_remap) {
        const int total = (int)gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = total >> 3, r = total & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile_c = bid % a.c_tiles;
    bid /= a.c_tiles;
    const int tile_n = bid % a.n_tiles;
    bid /= a.n_tiles;
    const int T = a.KH * a.KW;
    const int tap = bid % T;
    const int split = bid / T;
    const int r = tap / a.KW, s = tap % a.KW;

    const long m_begin = (long)split * a.m_per_split;
    long m_end = m_begin + a.m_per_split;
    if (m_end > a.M) m_end = a.M;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, a.x_bytes);

    const int nrow = tid / CHN, nch = tid % CHN;
    const int crow = tid / CHC, cch = tid % CHC;
    const int n0 = tile_n * TNW, c0 = tile_c * TCW;
    const bool n_ok = (n0 + nch * 8) < a.N;
    const bool c_ok = (c0 + cch * 8) < a.C;

    // The K loop walks the pixels front to back, KM per step, so NO per-step address arithmetic is redone from
    // scratch (the first version spent 3x the MFMA time in quarter-rate integer multiplies and 64-bit adds):
    //  * dY rows are linear in the pixel index: the lane offsets are fixed and the buffer descriptor (scalar
    //    registers) moves KM rows per step; its record count ends at the split's last row, so the ragged end
    //    reads zeros.
    //  * X rows need (ho, wo) for the padding test: each staged row carries (ho, wo, byte offset) and advances them
    //    by the constant decomposition of KM pixels = a.kb images + a.kh rows + a.kw columns, with two carries.
    uint32_t vn[PN];
#pragma unroll
    for (int p = 0; p < PN; ++p) vn[p] = n_ok ? (uint32_t)(((p * RPN + nrow) * a.N + n0 + nch * 8) * 2) : kOOB;
    int xho[PC], xwo[PC];
    uint32_t xoff[PC];
#pragma unroll
    for (int p = 0; p < PC; ++p) {
        const uint32_t mm = (uint32_t)(m_begin + p * RPC + crow);
        const uint32_t b = fdiv(mm, a.div_howo);
        const uint32_t rem = mm - b * a.div_howo.d;
        const uint32_t ho = fdiv(rem, a.div_wo);
        const uint32_t wo = rem - ho * a.div_wo.d;
        xho[p] = (int)ho;
        xwo[p] = (int)wo;
        const int hi = (int)ho * a.stride - a.pad + r, wi = (int)wo * a.stride - a.pad + s;
        xoff[p] = (uint32_t)((((int)b * a.Hin + hi) * a.Win + wi) * a.C + c0 + cch * 8) * 2u;  // modular when hi/wi < 0
    }
    const int dr = r - a.pad, ds = s - a.pad;

    bf16x8 rn[PN], rc[PC];
    auto gload = [&](long mbase) {
        const long left = m_end - mbase;
        const __amdgpu_buffer_rsrc_t rdy = make_rsrc(reinterpret_cast<const char*>(dy) + mbase * a.N * 2,
                                                     left > 0 ? (uint32_t)(left * a.N * 2) : 0u);
#pragma unroll
        for (int p = 0; p < PN; ++p) rn[p] = a.dy_stream ? buffer_load_bf16x8_stream(rdy, vn[p]) : buffer_load_bf16x8(rdy, vn[p]);
#pragma unroll
        for (int p = 0; p < PC; ++p) {
            const int hi = __mul24(xho[p], a.stride) + dr, wi = __mul24(xwo[p], a.stride) + ds;
            const bool ok = c_ok && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win;
            rc[p] = a.x_stream ? buffer_load_bf16x8_stream(rx, ok ? xoff[p] : kOOB) : buffer_load_bf16x8(rx, ok ? xoff[p] : kOOB);
            // advance this row by KM pixels
            int wo = xwo[p] + a.kw;
            const bool cw = wo >= a.Wout;
            wo -= cw ? a.Wout : 0;
            int ho = xho[p] + a.kh + (cw ? 1 : 0);
            const bool ch = ho >= a.Hout;
            ho -= ch ? a.Hout : 0;
            xwo[p] = wo;
            xho[p] = ho;
            xoff[p] += a.xstep + (cw ? a.xwrap_w : 0u) + (ch ? a.xwrap_h : 0u);
        }
    };
    auto lds_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PN; ++p)
            *reinterpret_cast<bf16x8*>(Ns + buf * N_BYTES + swz_off<RBN>(p * RPN + nrow, nch)) = rn[p];
#pragma unroll
        for (int p = 0; p < PC; ++p)
            *reinterpret_cast<bf16x8*>(Cs + buf * C_BYTES + swz_off<RBC>(p * RPC + crow, cch)) = rc[p];
    };

    f32x16 acc[TM][TC];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const long span = m_end - m_begin;
    const int nks = span > 0 ? (int)((span + KM - 1) / KM) : 0;
    if (nks > 0) {
        gload(m_begin);
        lds_store(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nks;
        if (more) gload(m_begin + (long)(ks + 1) * KM);
        const char* Nb = Ns + cur * N_BYTES;
        const char* Cb = Cs + cur * C_BYTES;
#pragma unroll
        for (int kk = 0; kk < KM / 16; ++kk) {
            bf16x8 af[TM], bfr[TC];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = tr_frag<RBN>(Nb, kk * 16, wn * (TNW / 2) + i * 32, lane);
#pragma unroll
            for (int j = 0; j < TC; ++j) bfr[j] = tr_frag<RBC>(Cb, kk * 16, wc * (TCW / 2) + j * 32, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) lds_store(cur ^ 1);
        __syncthreads();
    }

    // partial tile -> slab[split][n][tap][c]  (lane = column c: 128-B contiguous segments)
    float* out = slab + (long)split * a.N * T * a.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const int c = c0 + wc * (TCW / 2) + j * 32 + (lane & 31);
            const int nb = n0 + wn * (TNW / 2) + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int n = nb + (k & 3) + 8 * (k >> 2);
                if (n < a.N && c < a.C) out[((long)n * T + tap) * a.C + c] = acc[i][j][k];
            }
        }
}

template <int TNW, int TCW>
void launch_cfg(const void* dy, const void* x, float* slab, const WgradGeom& g, hipStream_t s) {
    WgradArgs a;
    a.B = g.B; a.Hin = g.Hin; a.Win = g.Win; a.C = g.C; a.Hout = g.Hout; a.Wout = g.Wout; a.N = g.N;
    a.KH = g.KH; a.KW = g.KW; a.stride = g.stride; a.pad = g.pad;
    a.n_tiles = (g.N + TNW - 1) / TNW;
    a.c_tiles = (g.C + TCW - 1) / TCW;
    a.splits = g.splits;
    // A pointwise layer between a narrow and a wide tensor reads the WIDE one through a single tile column / row:
    // exactly once, whole 128-B lines per row -> non-temporal (PDT_WGRAD_STREAM=0: off)
    static const bool stream_on = [] {
        const char* e = std::getenv("PDT_WGRAD_STREAM");
        return !(e && e[0] == '0');
    }();
    const bool one_tap = g.KH * g.KW == 1;
    a.dy_stream = stream_on && one_tap && a.c_tiles == 1 ? 1 : 0;
    a.x_stream = stream_on && one_tap && a.n_tiles == 1 && g.stride == 1 ? 1 : 0;
    // measured on MI355X (bench/resnet50_roofline.py, B=256): sharing one XCD's L2 between the taps / tiles of a
    // split wins where the layer is fetch-bound (64-128 channel 3x3: 260 -> 187 us, every 1x1: -5..-15 %) and
    // loses where >= 36 MFMA-bound workgroups then hammer the same L2 lines at once (256-channel 3x3: 100 -> 123 us)
    a.xcd_remap = !(g.KH * g.KW > 1 && g.C >= 256 && g.N >= 256);
    a.M = (long)g.B * g.Hout * g.Wout;
    long per = (a.M + g.splits - 1) / g.splits;
    per = (per + KM - 1) / KM * KM;
    a.m_per_split = per;
    a.div_wo = make_fastdiv((uint32_t)g.Wout);
    a.div_howo = make_fastdiv((uint32_t)(g.Hout * g.Wout));
    a.x_bytes = (uint32_t)((long)g.B * g.Hin * g.Win * g.C * 2);
    a.dy_bytes = (uint32_t)(a.M * g.N * 2);
    {
        const int hw = g.Hout * g.Wout;
        const int kb = KM / hw, rem = KM % hw;
        a.kh = rem / g.Wout;
        a.kw = rem % g.Wout;
        const uint32_t c2 = (uint32_t)g.C * 2u;
        a.xstep = ((uint32_t)kb * g.Hin * g.Win + (uint32_t)a.kh * g.stride * g.Win + (uint32_t)a.kw * g.stride) * c2;
        a.xwrap_w = ((uint32_t)g.stride * g.Win - (uint32_t)g.Wout * g.stride) * c2;
        a.xwrap_h = ((uint32_t)g.Hin * g.Win - (uint32_t)g.Hout * g.stride * g.Win) * c2;
    }
    const size_t lds = (size_t)2 * KM * (TNW + TCW) * 2;
    const int grid = a.n_tiles * a.c_tiles * g.KH * g.KW * g.splits;
    hipLaunchKernelGGL((k_conv_wgrad<TNW, TCW>), dim3(grid), dim3(256), lds, s, (const bf16*)dy, (const bf16*)x, slab,
                       a);
}

inline int tile_w(int dim) { return dim % 128 == 0 ? 128 : 64; }
}  // namespace

// Workgroups the split-K factor aims for.  Pointwise layers want ~4 per CU (1024).  Spatial filters want shorter
// splits: the taps of a split are adjacent workgroups that stream the same dY / X rows, and they only hit each
// other's lines in L2 while they stay close together; B=1024 sweep (bench/resnet50_roofline.py): 2048 gives
// -4..-12 % on the 128..512-channel 3x3s, and the 64-channel 3x3 (nine 64x64 tiles, fetch-bound) wants the
// 512-split cap (610 -> 495 us).  PDT_WGRAD_BLOCKS overrides all of them.
static const long kTargetBlocksEnv = [] {
    const char* e = std::getenv("PDT_WGRAD_BLOCKS");
    return e ? std::atol(e) : 0L;
}();
static long target_blocks(const WgradGeom& g) {
    if (kTargetBlocksEnv > 0) return kTargetBlocksEnv;
    if (g.KH * g.KW == 1) return 1024;
    return (g.C <= 64 && g.N <= 64) ? 8192 : 2048;
}

int conv_wgrad_splits(const WgradGeom& g) {
    if (gemm_tn_ok(g)) return gemm_tn_splits(g);  // wide layers: 256 x 256 tiles, one workgroup per CU (gemm_tn.hip)
    const int tn = tile_w(g.N), tc = tile_w(g.C);
    const long tiles = (long)((g.N + tn - 1) / tn) * ((g.C + tc - 1) / tc) * g.KH * g.KW;
    const long M = (long)g.B * g.Hout * g.Wout;
    long splits = (target_blocks(g) + tiles - 1) / tiles;
    if (kTargetBlocksEnv <= 0 && g.KH * g.KW > 1) {
        // the larger spatial-filter targets were measured with >= 4096 pixels per split; smaller batches keep at
        // least the pointwise target
        const long base = (1024 + tiles - 1) / tiles;
        const long by_len = M / 4096;
        if (splits > by_len) splits = by_len > base ? by_len : base;
    }
    const long max_splits = (M + 4 * KM - 1) / (4 * KM);  // at least 4 K-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 512) splits = 512;
    return (int)splits;
}

void launch_conv_wgrad(const void* dy, const void* x, float* slab, const WgradGeom& g, hipStream_t s) {
    if (gemm_tn_ok(g)) {
        launch_gemm_tn(dy, x, slab, g, s);
        return;
    }
    const int tn = tile_w(g.N), tc = tile_w(g.C);
    if (tn == 128 && tc == 128) launch_cfg<128, 128>(dy, x, slab, g, s);
    else if (tn == 128) launch_cfg<128, 64>(dy, x, slab, g, s);
    else if (tc == 128) launch_cfg<64, 128>(dy, x, slab, g, s);
    else launch_cfg<64, 64>(dy, x, slab, g, s);
}

}  // namespace pdt
