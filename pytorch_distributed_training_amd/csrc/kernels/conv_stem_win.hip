// 3-channel stem forward (7x7 / stride 2 / pad 3 -> 64 channels) as a window kernel on the packed image of
// conv_stem.hip.
//
// The generic implicit-GEMM kernel runs this layer at ~400 TFLOP/s: K = 7 x 32 is three short K tiles, the A tile is
// re-staged for every filter row although consecutive rows of output pixels share 5 of their 7 input rows, and the
// 64-column tile leaves half of a 256 x 128 build idle.  Here
//   * the packed weights (64 x 224 bf16 = 28 KiB) live in REGISTERS for the whole kernel: 7 filter rows x 4 channel
//     groups x one 16x16x32 A fragment = 112 VGPRs per wave (the MFMA runs transposed: D[channel][pixel], so the weights
//     are the A operand and 16 output pixels the B operand);
//   * a workgroup (8 waves, one per CU, persistent) owns RT = 8 output rows of one image: the 2*RT + 5 packed input
//     rows under them are ONE contiguous piece of memory, staged by LDS-DMA as linear 1-KiB pieces into one of two
//     buffers while the previous tile is being computed (counted vmcnt: the DMA is older than the tile's stores);
//   * an MFMA B fragment is a plain 16-B LDS read: lane (pixel p, k-quarter q) reads the two packed pixels
//     2*(ow0 + p) + 2*q of row 2*oh + kh; the 16 pixels of a fragment are 256 contiguous bytes -> conflict-free;
//   * the channel permutation of the weight fragments (row 4g + i of group nf = channel 16g + 4nf + i) leaves every
//     lane with 16 CONSECUTIVE channels of one pixel: the tile goes to memory as two 16-B stores per lane straight
//     from the accumulators (a fragment is 2 KiB contiguous), no LDS staging, and the per-channel sum / sum of
//     squares of the rounded outputs (BatchNorm statistics) accumulate per lane across the whole kernel.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int RT = 8;            // output rows per tile
constexpr int SW_THREADS = 512;  // 8 waves: RT * (Wo / 16) fragments, Wo / 16 per wave

struct StemWinArgs {
    const void* xp;   // [B][Hp][Wp][4] bf16
    const void* wp;   // [64][7][32] bf16
    void* y;          // [B][Ho][Wo][64] bf16
    float* slab;      // [grid][2][64] or null
    int B, Ho, Wo, Hp, Wp;
    int row_tiles, tiles;
    int ppw;          // 1-KiB DMA pieces per wave and window
    uint32_t xp_bytes;
};

__global__ __launch_bounds__(SW_THREADS, 1) void k_stem_win(const StemWinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = lane & 15, g = lane >> 4;

    // A fragments: lane (row r = px, k-quarter g) holds W[channel(nf, r)][kh*32 + 8g .. +7]
    bf16x8 wf[7][4];
    {
        const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const int ch = 16 * (px >> 2) + 4 * nf + (px & 3);
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) wf[kh][nf] = *reinterpret_cast<const bf16x8*>(wp + (ch * 7 + kh) * 32 + g * 8);
        }
    }
    // retire the weight loads where the compiler's wait bookkeeping sees it (vmcnt(0), builtin form): it cannot see the
    // hand-issued waits in the tile loop and would otherwise put its own vmcnt(0) in front of the first MFMA of
    // EVERY tile, i.e. wait for the window DMA of the next tile that was issued a moment before
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const i32x4 rs = make_rsrc_words(a.xp, a.xp_bytes);
    const uint32_t lds0 = lds_address(smem);
    const uint32_t buf_bytes = (uint32_t)a.ppw * 8192u;
    const int fpr = a.Wo >> 4;  // fragments per output row == fragments per wave
    char* xchg = smem + 2 * (size_t)buf_bytes + wave * 2304;  // this wave's output exchange tile [16][144 B]
    const uint32_t row_bytes = (uint32_t)a.Wp * 8u;

    auto issue = [&](int tile, int buf) {
        const int b = tile / a.row_tiles, rt = tile - b * a.row_tiles;
        const uint32_t base = (uint32_t)(b * a.Hp + 2 * rt * RT) * row_bytes + (uint32_t)lane * 16u;
        for (int p = 0; p < a.ppw; ++p) {
            const uint32_t piece = (uint32_t)(p * 8 + wave) * 1024u;
            lds_dma16(rs, base + piece, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)buf * buf_bytes + piece));
        }
    };

    float s[16], q[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = q[k] = 0.f;

    int tile = blockIdx.x;
    if (tile < a.tiles) issue(tile, 0);
    bool counted = false;  // the previous tile issued exactly 2 * 7 stores after the DMA of this one
    for (int it = 0; tile < a.tiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        if (counted) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // window `buf` complete; every wave is done reading the other one
        const int next = tile + gridDim.x;
        if (next < a.tiles) issue(next, buf ^ 1);
        const int b = tile / a.row_tiles, rt = tile - b * a.row_tiles;
        const char* win = smem + (size_t)buf * buf_bytes;
        bool full = fpr == 7;
        for (int j = 0; j < fpr; ++j) {
            const int f = wave + 8 * j;
            const int ohl = f / fpr, fc = f - ohl * fpr;
            const int oh = rt * RT + ohl;
            const char* src = win + (uint32_t)(2 * ohl) * row_bytes + (uint32_t)(2 * (fc * 16 + px) + 2 * g) * 8u;
            bf16x8 xf[7];
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) xf[kh] = *reinterpret_cast<const bf16x8*>(src + (uint32_t)kh * row_bytes);
            f32x4 acc[4];
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                    acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][nf], xf[kh], acc[nf], 0, 0, 0);
            if (oh < a.Ho) {  // (wave-uniform) rows past the image: the last tile of an image when Ho % RT != 0
                bf16x8 o[2];
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bf16 r = (bf16)acc[nf][i];
                        const float v = (float)r;
                        const int k = 4 * nf + i;
                        o[k >> 3][k & 7] = r;
                        s[k] += v;
                        q[k] += v * v;
                    }
                // A lane holds 32 B of a 128-B pixel, so a store instruction straight from here covers every line in
                // pieces: fine for plain stores (the L2 merges them) but 1.5x slower with the non-temporal hint, and
                // plain stores of 2 GB leave dirty lines that the next kernel's reads wait behind.  The fragment
                // (16 pixels x 128 B) therefore takes a turn through this wave's 2.25 KiB of LDS (pixel pitch 144 B:
                // conflict-free both ways) and leaves as two linear 1-KiB non-temporal stores.
                char* xs = xchg + px * 144 + g * 32;
                *reinterpret_cast<bf16x8*>(xs) = o[0];
                *reinterpret_cast<bf16x8*>(xs + 16) = o[1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(xchg + (lane >> 3) * 144 + (lane & 7) * 16);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(xchg + (8 + (lane >> 3)) * 144 + (lane & 7) * 16);
                char* dst = reinterpret_cast<char*>(a.y) + ((size_t)(b * a.Ho + oh) * a.Wo + fc * 16) * 128 + lane * 16;
                st_stream(dst, w0);
                st_stream(dst + 1024, w1);
                __builtin_amdgcn_wave_barrier();  // (the next fragment's writes come after these reads)
            } else {
                full = false;
            }
        }
        counted = full && next < a.tiles;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.slab == nullptr) return;
    // statistics: over the 16 pixel lanes, then over the 8 waves (fixed order -> deterministic)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            s[k] += __shfl_xor(s[k], o, 64);
            q[k] += __shfl_xor(q[k], o, 64);
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [8][2][64]
    if (px == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            red[(wave * 2 + 0) * 64 + 16 * g + k] = s[k];
            red[(wave * 2 + 1) * 64 + 16 * g + k] = q[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w * 128 + threadIdx.x];
        a.slab[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the stem on the same packed image: dW[n][kh][slot*4 + c] = sum over pixels dy[px][n] * X_kh[px][.]
// is a TN product (both operands have the REDUCTION index, the pixel, as their slow axis) of 64 x 224 outputs over
// B*Ho*Wo pixels -- bound by the one read of dy (2 GB at B = 1280).  A workgroup (4 waves, one per SIMD, persistent,
// one per CU) owns WRT = 4 output rows of an image: the dy rows (swizzled 128-B pixel rows) and the 2*WRT + 5 packed
// input rows under them are staged by LDS-DMA into one of two buffers while the previous tile is computed.  A wave
// holds the WHOLE 64 x 224 result in accumulators (2 x 7 tiles of MFMA 32x32x16, 224 registers) and takes every 4th
// block of 16 pixels: 2 dy fragments + 7 X fragments, all transposing LDS reads (ds_read_tr16_b64; the X fragment of
// filter row kh is simply the window at row 2*oh + kh, 16 B per pixel step), feed 14 MFMAs.  The four waves and then
// the workgroups are summed at the end (LDS, then the slab reduction of the other weight-gradient kernels).
constexpr int WRT = 4;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

struct StemWgradArgs {
    const void* dy;   // [B][Ho][Wo][64] bf16
    const void* xp;   // [B][Hp][Wp][4] bf16
    float* slab;      // [grid][64][147]
    int B, Ho, Wo, Hp, Wp;
    int row_tiles, tiles;
    int dpw, wpw;     // 1-KiB DMA pieces per wave: dy tile, window
    uint32_t dy_bytes, xp_bytes;
};

__device__ __forceinline__ bf16x8 tr_pair(const char* p, int hi_off) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + hi_off));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 1) void k_stem_wgrad_win(const StemWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i32x4 rs_dy = make_rsrc_words(a.dy, a.dy_bytes), rs_x = make_rsrc_words(a.xp, a.xp_bytes);
    const uint32_t lds0 = lds_address(smem);
    const uint32_t dy_tile = (uint32_t)a.Wo * (WRT * 128u);          // bytes of the dy part of a buffer
    const uint32_t buf_bytes = dy_tile + (uint32_t)a.wpw * 4096u;
    const uint32_t row_bytes = (uint32_t)a.Wp * 8u;
    const int fpr = a.Wo >> 4;

    // DMA lane plan of a dy piece (8 pixel rows x 128 B): LDS position (row, pos) holds the logical 16-B chunk
    // pos ^ key(row), key = (row & 2) | (bit 3 of the pixel index << 2): the 4 consecutive rows one transposed read
    // touches fall into 4 different 32-B granules of the 256-B bank period, the rows 8 further into the other four
    const uint32_t dy_lane = (uint32_t)(lane >> 3) * 128u +
                             (uint32_t)(((lane & 7) ^ (((lane >> 3) & 2) | ((wave & 1) << 2))) << 4);
    auto issue = [&](int tile, int buf) {
        const int b = tile / a.row_tiles, rt = tile - b * a.row_tiles;
        const int oh0 = rt * WRT;
        const uint32_t dbase = (uint32_t)((b * a.Ho + oh0) * a.Wo) * 128u + dy_lane;
        const uint32_t l0 = lds0 + (uint32_t)buf * buf_bytes;
        for (int i = 0; i < a.dpw; ++i) {
            const int p = wave + 4 * i;              // (p & 1 == wave & 1: the key above)
            const bool live = oh0 + (p * 8) / a.Wo < a.Ho;  // rows past the image contribute zeros
            lds_dma16_stream(rs_dy, live ? dbase + (uint32_t)p * 1024u : kOOB, __builtin_amdgcn_readfirstlane(l0 + (uint32_t)p * 1024u));
        }
        const uint32_t xbase = (uint32_t)(b * a.Hp + 2 * oh0) * row_bytes + (uint32_t)lane * 16u;
        for (int i = 0; i < a.wpw; ++i) {
            const uint32_t piece = (uint32_t)(wave + 4 * i) * 1024u;
            lds_dma16(rs_x, xbase + piece, __builtin_amdgcn_readfirstlane(l0 + dy_tile + piece));
        }
    };

    // fragment read plan: 16-lane group gi reads pixel rows kb + q (and + 4), kb = 8 * (gi / 2), q = (lane % 16) / 4,
    // columns 16 * (gi % 2) + 4 * (lane % 4) .. + 3 -> MFMA 32x32x16 operand (row or column lane % 32, k = 8 * (lane / 32) + 0..7)
    const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int kb = 8 * (gi >> 1);
    const int key = (qq & 2) | ((gi >> 1) << 2);
    const int low = ((gi & 1) << 1) | (pp >> 1);
    const uint32_t a_row = (uint32_t)(kb + qq) * 128u + (uint32_t)(pp & 1) * 8u;
    const uint32_t a_off0 = a_row + (uint32_t)((low ^ key) << 4), a_off1 = a_row + (uint32_t)(((low ^ key) ^ 4) << 4);
    const uint32_t b_off = (uint32_t)(kb + qq) * 16u + (uint32_t)(gi & 1) * 32u + (uint32_t)pp * 8u;

    f32x16 acc[2][7];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nh][kh][r] = 0.f;

    struct Frags {
        bf16x8 d[2], x[7];
    };
    auto load = [&](Frags& f, const char* buf, int j) {
        const int st = wave + 4 * j;
        const int ohl = st / fpr, fc = st - ohl * fpr;
        const char* dp = buf + (uint32_t)(ohl * a.Wo + fc * 16) * 128u;
        f.d[0] = tr_pair(dp + a_off0, 512);
        f.d[1] = tr_pair(dp + a_off1, 512);
        const char* xq = buf + dy_tile + (uint32_t)(2 * ohl) * row_bytes + (uint32_t)(fc * 16) * 16u + b_off;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) f.x[kh] = tr_pair(xq + (uint32_t)kh * row_bytes, 64);
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
        Frags cur, nxt;
        load(cur, base, 0);
        for (int j = 0; j < fpr; ++j) {
            if (j + 1 < fpr) load(nxt, base, j + 1);
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
                    acc[nh][kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.d[nh], cur.x[kh], acc[nh][kh], 0, 0, 0);
            cur = nxt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the four waves: one filter row at a time through LDS ([wave][64 n][32 cols] fp32 = 32 KiB), fixed order
    float* red = reinterpret_cast<float*>(smem);
    float* out = a.slab + (size_t)blockIdx.x * (64 * 147);
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        __syncthreads();
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                red[(wave * 64 + n) * 32 + (lane & 31)] = acc[nh][kh][r];
            }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = threadIdx.x + 256 * e;
            const int n = idx >> 5, col = idx & 31, slot = col >> 2, c = col & 3;
            const float t = red[idx] + red[2048 + idx] + red[4096 + idx] + red[6144 + idx];
            if (slot >= 1 && c < 3) out[n * 147 + kh * 21 + (slot - 1) * 3 + c] = t;
        }
    }
}

int stem_win_mode() {  // 0 off, 1 by work size, 2 whenever the geometry fits (tests)
    static const int mode = [] {
        const char* e = std::getenv("PDT_STEM_WIN");
        return e ? std::atoi(e) : 1;
    }();
    return mode;
}
int g_stem_win_force = -1;

int stem_ppw(int Wo) {
    const long window = (long)(2 * RT + 5) * (2 * Wo + 8) * 8;
    return (int)((window + 8191) / 8192);
}

}  // namespace

void set_stem_win_mode(int mode) { g_stem_win_force = mode; }

bool conv_stem_win_ok(int B, int Ho, int Wo, int N) {
    const int mode = g_stem_win_force >= 0 ? g_stem_win_force : stem_win_mode();
    if (mode == 0 || N != 64 || Wo < 16 || Wo % 16 != 0 || Ho < 1) return false;
    if (2L * stem_ppw(Wo) * 8192 + 8 * 2304 > 160 * 1024) return false;
    const long tiles = (long)B * ((Ho + RT - 1) / RT);
    if (tiles >= (1L << 31)) return false;
    // a persistent grid of one workgroup per CU: below two tiles per CU the implicit-GEMM launch fills the chip better
    return mode >= 2 || tiles >= 512;
}

int conv_stem_win_grid(int B, int Ho) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n;
    }();
    const long tiles = (long)B * ((Ho + RT - 1) / RT);
    return (int)(tiles < cus ? tiles : cus);
}

void launch_conv_stem_win(const void* xp, const void* wp, void* y, float* slab, int B, int Ho, int Wo, hipStream_t s) {
    const long xp_bytes = (long)B * (2 * Ho + 6) * (2 * Wo + 8) * 8;
    if (Wo < 16 || Wo % 16 != 0 || 2L * stem_ppw(Wo) * 8192 + 8 * 2304 > 160 * 1024 || xp_bytes >= (1L << 31) ||
        (long)B * Ho * Wo * 128 >= (1L << 32)) {
        fprintf(stderr, "launch_conv_stem_win: unsupported geometry %d x %d x %d\n", B, Ho, Wo);
        abort();
    }
    StemWinArgs a;
    a.xp = xp; a.wp = wp; a.y = y; a.slab = slab;
    a.B = B; a.Ho = Ho; a.Wo = Wo; a.Hp = 2 * Ho + 6; a.Wp = 2 * Wo + 8;
    a.row_tiles = (Ho + RT - 1) / RT;
    a.tiles = B * a.row_tiles;
    a.ppw = stem_ppw(Wo);
    a.xp_bytes = (uint32_t)xp_bytes;
    const int lds = 2 * a.ppw * 8192 + 8 * 2304;
    static int attr_lds = 0;
    if (lds > attr_lds) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_win),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k_stem_win, dim3(conv_stem_win_grid(B, Ho)), dim3(SW_THREADS), lds, s, a);
}

namespace {
int stem_wgrad_wpw(int Wo) {
    const long window = (long)(2 * WRT + 5) * (2 * Wo + 8) * 8;
    return (int)((window + 4095) / 4096);
}
long stem_wgrad_lds(int Wo) { return 2L * ((long)Wo * WRT * 128 + (long)stem_wgrad_wpw(Wo) * 4096); }
}  // namespace

bool conv_stem_wgrad_win_ok(int B, int Ho, int Wo, int N) {
    const int mode = g_stem_win_force >= 0 ? g_stem_win_force : stem_win_mode();
    if (mode == 0 || N != 64 || Wo < 16 || Wo % 16 != 0 || Ho < 1) return false;
    if (stem_wgrad_lds(Wo) > 160 * 1024) return false;
    const long tiles = (long)B * ((Ho + WRT - 1) / WRT);
    if (tiles >= (1L << 31)) return false;
    return mode >= 2 || tiles >= 1024;
}

int conv_stem_wgrad_win_grid(int B, int Ho) {
    const long tiles = (long)B * ((Ho + WRT - 1) / WRT);
    const int cus = conv_stem_win_grid(1 << 20, RT);  // (CU count)
    return (int)(tiles < cus ? tiles : cus);
}

// slab: [conv_stem_wgrad_win_grid(B, Ho)][64][147] fp32, one partial weight gradient per workgroup
void launch_conv_stem_wgrad_win(const void* dy, const void* xp, float* slab, int B, int Ho, int Wo, hipStream_t s) {
    const long xp_bytes = (long)B * (2 * Ho + 6) * (2 * Wo + 8) * 8, dy_bytes = (long)B * Ho * Wo * 128;
    if (Wo < 16 || Wo % 16 != 0 || stem_wgrad_lds(Wo) > 160 * 1024 || xp_bytes >= (1L << 31) || dy_bytes >= (1L << 31)) {
        fprintf(stderr, "launch_conv_stem_wgrad_win: unsupported geometry %d x %d x %d\n", B, Ho, Wo);
        abort();
    }
    StemWgradArgs a;
    a.dy = dy; a.xp = xp; a.slab = slab;
    a.B = B; a.Ho = Ho; a.Wo = Wo; a.Hp = 2 * Ho + 6; a.Wp = 2 * Wo + 8;
    a.row_tiles = (Ho + WRT - 1) / WRT;
    a.tiles = B * a.row_tiles;
    a.dpw = Wo * WRT / 8 / 4;
    a.wpw = stem_wgrad_wpw(Wo);
    a.dy_bytes = (uint32_t)dy_bytes;
    a.xp_bytes = (uint32_t)xp_bytes;
    const int lds = (int)std::max(stem_wgrad_lds(Wo), 32768L);
    static int attr_lds = 0;
    if (lds > attr_lds) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad_win),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k_stem_wgrad_win, dim3(conv_stem_wgrad_win_grid(B, Ho)), dim3(256), lds, s, a);
}

}  // namespace pdt
