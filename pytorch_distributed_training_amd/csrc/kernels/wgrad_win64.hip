// Weight gradient of the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolutions (ResNet layer1: 56x56 images, the
// largest pixel count of the net) as a TN product out of an LDS window.
//
//   dW[n][tap][c] = sum over pixels dy[px][n] * x[px + tap][c]           (64 x 9 x 64 outputs, M = B*H*W pixels)
//
// The per-tap split-K kernel (conv_wgrad.hip) stages dy and a shifted x tile for every tap: 9 passes over both
// operands through LDS at 64-wide tiles, ~490 TFLOP/s.  Here a workgroup (4 waves, one per SIMD, persistent, one per
// CU) owns WR = 4 image rows: the dy rows and the (WR + 2) x (W + 2) input window with its zero halo are staged ONCE by
// LDS-DMA into one of two buffers (the next tile loads while this one is computed), and all nine taps read their
// shifted operand out of that window.  Wave w owns input channels 16w .. 16w + 15 of ALL 64 output channels and taps:
// 4 x 9 accumulator tiles of MFMA 16x16x32 (144 registers), so per 32 pixels it reads 4 dy fragments + 9 x fragments
// (transposing LDS reads, ds_read_tr16_b64: both operands have the pixel as their slow axis) for 36 MFMAs.  Nothing is
// summed across waves; every workgroup writes one partial [64][9][64] row of the slab.
//
// LDS rows are 128-B pixels; one transposed read touches 4 consecutive pixels x 32 B, which collide pairwise (256-B
// bank period).  The 32-B granule index is therefore XORed with bit 1 of the pixel index on the way in (DMA source
// swizzle): any 4 consecutive pixels land in 4 different granules.
//
// Measured at B = 1280, 56x56 (MI355X): 615 -> 308 us per launch (960 TFLOP/s incl. the slab reduction).  Steps on the
// way: first version 395 us; pinned read / MFMA interleave with swapped fragment sets instead of register copies 365;
// step loop unrolled with per-lane constant addresses 350; per-lane constant DMA source offsets 308 (the address
// arithmetic of the 18 staging pieces per wave and tile -- divisions, halo tests -- had been a quarter of a tile's
// time).  Sharing the x reads between the taps of a filter row changed nothing: a 32-pixel step runs its 36 MFMAs in
// ~970 cycles, which is the ~1.45 PFLOP/s the matrix pipe sustains in every kernel here, not the LDS.  Without any
// staging the kernel takes 280 us; the one read of both tensors at 5.5 TB/s is 210 us.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int WR = 4;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct WgWin64Args {
    const void* dy;  // [B][H][W][64]
    const void* x;   // [B][H][W][64]
    float* slab;     // [grid][64][9][64]
    int B, H, W, W2;
    int row_tiles, tiles;
    int dpw, wpw;    // 1-KiB DMA pieces per wave: dy rows, window
    uint32_t bytes;  // B*H*W*128 (both tensors)
    FastDiv div_w, div_w2;
};

// 16-B chunk swizzle of a 128-B pixel row (see the header): bit 1 of the pixel index flips the 32-B granule.  (Also
// flipping the 64-B half with bit 3, for the next lane group of a read, measured 6 % slower.)
__device__ __forceinline__ int swz(int px) { return px & 2; }

__device__ __forceinline__ bf16x8 tr_pair(const char* p) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 512));  // pixels + 4
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// STEPS: pixel steps per tile (W / 8) known at compile time -> the step loop is unrolled and every fragment address
// of a tile is a per-lane constant (one v_add per read); 0 = any W, addresses computed per step
template <int STEPS>
__global__ __launch_bounds__(256, 1) void k_wgrad_win64(const WgWin64Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i32x4 rs_dy = make_rsrc_words(a.dy, a.bytes), rs_x = make_rsrc_words(a.x, a.bytes);
    const uint32_t lds0 = lds_address(smem);
    const uint32_t dy_part = (uint32_t)a.W * (WR * 128u);
    const uint32_t buf_bytes = dy_part + (uint32_t)a.wpw * 4096u;
    const int steps = STEPS ? STEPS : a.W >> 3;  // WR * W / 32

    // ---- staging.  A piece is 8 pixels x 128 B; lane l fills (pixel l / 8, 16-B position l % 8) with the logical
    // chunk (l % 8) ^ (pixel index & 2)
    const int prow = lane >> 3, ppos = lane & 7;
    // The source offset of a piece relative to the tile's first pixel does not depend on the tile (only whether its
    // image row exists does): with the piece counts known at compile time they are per-lane constants, and a tile's
    // staging costs 4 vector instructions per piece instead of ~25 (divisions, halo tests) -- measured: the address
    // arithmetic of 18 pieces per wave and tile was a quarter of the tile's time.
    constexpr int DPW = STEPS;                                   // W / 8 pieces of dy rows per wave
    constexpr int WPW = STEPS ? ((WR + 2) * (STEPS * 8 + 2) * 128 + 4095) / 4096 : 1;
    uint32_t drel[STEPS ? DPW : 1], xrel[WPW];
    int drow[STEPS ? DPW : 1], xwr[WPW];
    if constexpr (STEPS > 0) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int p = wave + 4 * i, px = p * 8 + prow;
            drel[i] = (uint32_t)px * 128u + (uint32_t)((ppos ^ swz(px)) << 4);
            drow[i] = (int)fdiv((uint32_t)(p * 8), a.div_w);
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int p = wave + 4 * i, wp = p * 8 + prow;
            const int wr = (int)fdiv((uint32_t)wp, a.div_w2), wc = wp - wr * a.W2;
            const bool ok = wr < WR + 2 && (unsigned)(wc - 1) < (unsigned)a.W;
            xrel[i] = (uint32_t)(((wr - 1) * a.W + wc - 1) * 128) + (uint32_t)((ppos ^ swz(wp)) << 4);
            xwr[i] = ok ? wr : (1 << 24);  // (fails every row test)
        }
    }
    auto issue = [&](int tile, int buf) {
        const int b = tile / a.row_tiles, rt = tile - b * a.row_tiles;
        const int oh0 = rt * WR;
        const uint32_t l0 = lds0 + (uint32_t)buf * buf_bytes;
        const uint32_t img = (uint32_t)(b * a.H) * (uint32_t)a.W;  // pixel index of the image
        if constexpr (STEPS > 0) {
            const uint32_t tbase = (img + (uint32_t)(oh0 * a.W)) * 128u;
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                const uint32_t piece = (uint32_t)(wave + 4 * i) * 1024u;
                lds_dma16_stream(rs_dy, oh0 + drow[i] < a.H ? tbase + drel[i] : kOOB, __builtin_amdgcn_readfirstlane(l0 + piece));
            }
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                const uint32_t piece = (uint32_t)(wave + 4 * i) * 1024u;
                const bool live = (unsigned)(oh0 - 1 + xwr[i]) < (unsigned)a.H;
                lds_dma16(rs_x, live ? tbase + xrel[i] : kOOB, __builtin_amdgcn_readfirstlane(l0 + dy_part + piece));
            }
            return;
        }
        for (int i = 0; i < a.dpw; ++i) {
            const int p = wave + 4 * i;
            const int px = p * 8 + prow;
            const bool live = oh0 + (int)fdiv((uint32_t)(p * 8), a.div_w) < a.H;  // rows past the image: zeros
            const uint32_t off = (img + (uint32_t)(oh0 * a.W + px)) * 128u + (uint32_t)((ppos ^ swz(px)) << 4);
            lds_dma16_stream(rs_dy, live ? off : kOOB, __builtin_amdgcn_readfirstlane(l0 + (uint32_t)p * 1024u));
        }
        for (int i = 0; i < a.wpw; ++i) {
            const int p = wave + 4 * i;
            const int wp = p * 8 + prow;                       // window pixel: row wp / W2 (image row oh0 - 1 + ..)
            const int wr = (int)fdiv((uint32_t)wp, a.div_w2), wc = wp - wr * a.W2;
            const int ih = oh0 - 1 + wr, iw = wc - 1;
            const bool live = wr < WR + 2 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const uint32_t off = (img + (uint32_t)(ih * a.W + iw)) * 128u + (uint32_t)((ppos ^ swz(wp)) << 4);
            lds_dma16(rs_x, live ? off : kOOB, __builtin_amdgcn_readfirstlane(l0 + dy_part + (uint32_t)p * 1024u));
        }
    };

    // ---- fragment reads: 16-lane group gi covers pixels 8 gi + q (and + 4), q = (lane % 16) / 4, of a 32-pixel step,
    // columns 4 * (lane % 4) .. + 3 of a 16-column group -> MFMA operand (row / column lane % 16, k = 8 gi + 0..7)
    const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    uint32_t a_off[4];  // dy: 16 output channels ng*16.. = granule ng of the pixel row
#pragma unroll
    for (int ng = 0; ng < 4; ++ng)
        a_off[ng] = (uint32_t)(8 * gi + qq) * 128u + (uint32_t)((((ng << 1) | (pp >> 1)) ^ swz(8 * gi + qq)) << 4) + (uint32_t)(pp & 1) * 8u;
    const uint32_t xchunk = (uint32_t)((wave << 1) | (pp >> 1));  // this wave's 16 input channels
    const uint32_t xlow = (uint32_t)(pp & 1) * 8u;

    f32x4 acc[4][9];
#pragma unroll
    for (int ng = 0; ng < 4; ++ng)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[ng][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // x fragments: the three taps of a filter row read pixels p .. p+7, p+1 .. p+8, p+2 .. p+9 of the same window row.
    // One set of reads covers p .. p+11 (three transposed reads of 4 pixels); tap dw = 0 is dwords 0-3, dw = 2 dwords
    // 1-4, dw = 1 four v_alignbit across neighbours: 9 reads per step instead of 18.
    struct Frags {
        bf16x8 d[4];
        u32x4 xa[3];   // per filter row: pixels 0-7 (two bf16 per dword)
        uint32_t xe[3];  // pixels 8, 9
    };
    // pixel-step bookkeeping of a fragment set: the 8 pixels of a lane group lie in one image row (W % 8 == 0)
    auto window_pixel = [&](int st) {
        const int px0 = st * 32 + 8 * gi;
        const int r = (int)fdiv((uint32_t)px0, a.div_w), col = px0 - r * a.W;
        return r * a.W2 + col + qq;  // under tap (0, 0)
    };
    auto x_offset = [&](int wp0, int dh) {
        const int wp = wp0 + dh * a.W2;
        return dy_part + (uint32_t)wp * 128u + ((xchunk ^ (uint32_t)(wp & 2)) << 4) + xlow;
    };
    uint32_t xoff[STEPS ? STEPS : 1][3];
    if constexpr (STEPS > 0) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) xoff[st][dh] = x_offset(window_pixel(st), dh);
    }
    auto load_x = [&](Frags& f, const char* buf, int st, int dh) {
        const char* p = buf + (STEPS ? xoff[STEPS ? st : 0][dh] : x_offset(window_pixel(st), dh));
        f.xa[dh] = __builtin_bit_cast(u32x4, tr_pair(p));
        const bf16x4 e = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 1024));  // pixels + 8 .. + 11
        f.xe[dh] = __builtin_bit_cast(u32x2, e)[0];
    };
    auto x_frag = [&](const Frags& f, int t) {
        const u32x4 v = f.xa[t / 3];
        const uint32_t e = f.xe[t / 3];
        u32x4 o;
        if (t % 3 == 0) o = v;
        else if (t % 3 == 2) o = u32x4{v[1], v[2], v[3], e};
        else o = u32x4{__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                       __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(e, v[3], 16)};
        return __builtin_bit_cast(bf16x8, o);
    };
    // One 32-pixel step: the 36 MFMAs of `cur`, with the 17 fragment reads of the NEXT step (`nxt`, pixel step stn)
    // issued in between.  The order is pinned (sched_barrier): with one wave per SIMD nothing else hides the LDS
    // latency, and the counter behind s_waitcnt lgkmcnt only goes to 15.
    auto step = [&](const Frags& cur, Frags& nxt, const char* buf, int stn) {
        const char* dp = buf + (uint32_t)stn * 4096u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < 4) nxt.d[t] = tr_pair(dp + a_off[t]);
            if (t >= 4 && t < 7) load_x(nxt, buf, stn, t - 4);
            const bf16x8 xf = x_frag(cur, t);
#pragma unroll
            for (int ng = 0; ng < 4; ++ng)
                acc[ng][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.d[ng], xf, acc[ng][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int tile = blockIdx.x;
    if (tile < a.tiles) issue(tile, 0);
    for (int it = 0; tile < a.tiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // buffer `buf` complete; every wave is done reading the other one
        const int next = tile + gridDim.x;
        if (next < a.tiles) issue(next, buf ^ 1);
        const char* base = smem + (size_t)buf * buf_bytes;
        Frags fa, fb;
#pragma unroll
        for (int ng = 0; ng < 4; ++ng) fa.d[ng] = tr_pair(base + a_off[ng]);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) load_x(fa, base, 0, dh);
        // two steps per trip, the fragment sets swap roles (no register copies); past the last step the prefetch
        // re-reads the last one (in range, unused)
        if constexpr (STEPS > 0) {
#pragma unroll
            for (int st = 0; st < STEPS; st += 2) {
                step(fa, fb, base, st + 1 < STEPS ? st + 1 : STEPS - 1);
                if (st + 1 < STEPS) step(fb, fa, base, st + 2 < STEPS ? st + 2 : STEPS - 1);
            }
        } else {
            for (int st = 0; st < steps; st += 2) {
                step(fa, fb, base, min(st + 1, steps - 1));
                if (st + 1 < steps) step(fb, fa, base, min(st + 2, steps - 1));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // D: lane holds input channel lane % 16 of this wave's group, output channels 16 ng + 4 (lane / 16) + 0..3
    float* out = a.slab + (size_t)blockIdx.x * (64 * 9 * 64) + wave * 16 + (lane & 15);
#pragma unroll
    for (int ng = 0; ng < 4; ++ng)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[((16 * ng + 4 * (lane >> 4) + i) * 9 + t) * 64] = acc[ng][t][i];
}

int wg64_mode() {  // 0 off, 1 by work size, 2 whenever the geometry fits (tests)
    static const int mode = [] {
        const char* e = std::getenv("PDT_WGRAD_WIN64");
        return e ? std::atoi(e) : 1;
    }();
    return mode;
}
int g_wg64_force = -1;

int wg64_wpw(int W) { return (int)(((long)(WR + 2) * (W + 2) * 128 + 4095) / 4096); }
long wg64_lds(int W) { return 2L * ((long)W * WR * 128 + (long)wg64_wpw(W) * 4096); }
bool wg64_geometry(int H, int W) { return H >= 1 && W >= 8 && W % 8 == 0 && wg64_lds(W) <= 160 * 1024; }

}  // namespace

void set_wgrad_win64_mode(int mode) { g_wg64_force = mode; }

bool wgrad_win64_ok(int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad) {
    const int mode = g_wg64_force >= 0 ? g_wg64_force : wg64_mode();
    if (mode == 0 || C != 64 || N != 64 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || !wg64_geometry(H, W))
        return false;
    const long tiles = (long)B * ((H + WR - 1) / WR);
    if (tiles >= (1L << 31) || (long)B * H * W * 128 >= (1L << 31)) return false;
    return mode >= 2 || tiles >= 1024;  // persistent, one workgroup per CU: needs a few tiles per CU to pay
}

int wgrad_win64_grid(int B, int H) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n;
    }();
    const long tiles = (long)B * ((H + WR - 1) / WR);
    return (int)(tiles < cus ? tiles : cus);
}

// slab: [wgrad_win64_grid(B, H)][64][3][3][64] fp32, one partial weight gradient per workgroup
void launch_wgrad_win64(const void* dy, const void* x, float* slab, int B, int H, int W, hipStream_t s) {
    if (!wg64_geometry(H, W) || (long)B * H * W * 128 >= (1L << 31)) {
        fprintf(stderr, "launch_wgrad_win64: unsupported geometry %d x %d x %d\n", B, H, W);
        abort();
    }
    WgWin64Args a;
    a.dy = dy; a.x = x; a.slab = slab;
    a.B = B; a.H = H; a.W = W; a.W2 = W + 2;
    a.row_tiles = (H + WR - 1) / WR;
    a.tiles = B * a.row_tiles;
    a.dpw = W * WR / 8 / 4;
    a.wpw = wg64_wpw(W);
    a.bytes = (uint32_t)((long)B * H * W * 128);
    a.div_w = make_fastdiv((uint32_t)W);
    a.div_w2 = make_fastdiv((uint32_t)(W + 2));
    const int lds = (int)wg64_lds(W);
    static int attr_lds = 0;
    if (lds > attr_lds) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_win64<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_win64<7>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    static const bool unrolled = [] {
        const char* e = std::getenv("PDT_WGRAD_WIN64_UNROLL");
        return !(e && std::atoi(e) == 0);
    }();
    if (W == 56 && unrolled) hipLaunchKernelGGL(k_wgrad_win64<7>, dim3(wgrad_win64_grid(B, H)), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_wgrad_win64<0>, dim3(wgrad_win64_grid(B, H)), dim3(256), lds, s, a);
}

}  // namespace pdt
