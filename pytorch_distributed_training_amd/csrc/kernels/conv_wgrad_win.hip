// Window ("all taps at once") weight gradient for spatial filters on the gfx950 matrix cores.
//
//   dW[n][tap][c] = sum_pixels dY[pix][n] * X[src(pix, tap)][c]
//
// conv_wgrad.hip gives every filter tap its own workgroup, so a 3x3 layer streams dY and X nine
// times through L2 and a 64-channel layer (one 64x64 tile per tap) spends more LDS cycles than
// MFMA cycles.  Here a workgroup owns an (n, c) tile for ALL taps:
//
//  * a K-chunk is R whole output rows of one image.  dY rows go to LDS once ([pixel][64 n]); the
//    input window those rows touch ((R-1)*stride + KH source rows, halo included) goes to LDS once
//    ([window pixel][channels]); every tap then reads the SAME window at a different pixel offset.
//    Per chunk that is (1 + window/rows) tiles of global traffic instead of 2 * taps.
//  * both operands are K(pixel)-strided, so fragments come from the CDNA4 transposing LDS read
//    (ds_read_b64_tr_b16); the dY fragment is shared by all taps of a wave: 1 + taps fragment
//    reads per `taps` MFMAs.
//  * LDS rows are padded instead of swizzled (the tap shift must stay a constant byte offset):
//    the pixel pitch is chosen so that the four consecutive K rows a 16-lane group reads fall on
//    four distinct 64-B quarters of the 256-B bank row (dY rows 192 B; window pixels 64 B for stride 1,
//    96 B for stride 2).
//  * zero padding / ragged chunks come from the buffer range check (out-of-image -> zeros).
//  * split-K over chunks into an fp32 slab [split][...] summed by a second launch in a fixed order
//    (deterministic; float atomics are neither that nor fast on this chip).
//
// MODE 0: 3x3 convs, tile 64(n) x 32(c), 384 threads = 2(n) x 3(filter row) waves, each wave the three
//         taps of its filter row (48 accumulators: several workgroups per CU stay resident).
// MODE 1: the 3-channel stem on the packed input of conv_stem.hip: a "source pixel" is two image
//         pixels x 4 channels (16 B), a tap is one filter ROW (8 kw slots x 4 ch = 32 columns), tile
//         64(n) x 32 columns, 256 threads = 2(n) x 2(tap parity) waves.
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int kYPitch = 192;  // bytes per dY row in LDS: 64 n x 2 B + 64 B pad (see bank note above)
constexpr int max_slots(int mode) { return mode == 0 ? 6 : 8; }  // 16-B staging loads per thread and chunk
constexpr int kMaxKP = 112;              // K rows (pixels) per chunk, a multiple of 16
constexpr int kMaxKSteps = kMaxKP / 16;

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

struct WinArgs {
    int B, Ho, Wo, N;
    int Hs, Ws, xpix;
    int C, T;
    int sh, sw, ph, pw;
    int R, P, KP, WR, WCOLS;
    int cpp_log2, pixb;
    int nch_x, sd, sx;
    int chunks_per_img, total_chunks, chunks_per_split;
    int n_tiles, c_tiles;
    int tapoff[2];   // LDS bytes per window row, per window pixel
    FastDiv div_wo, div_wcols, div_cpi;
    uint32_t x_bytes, dy_bytes;
};

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 384 : 256, MODE == 0 ? 3 : 2) void k_wgrad_win(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                      float* __restrict__ slab, const WinArgs a) {
    constexpr int NT = MODE == 0 ? 384 : 256;
    constexpr int TPW = MODE == 0 ? 3 : 4;   // taps per wave
    constexpr int kMaxSlots = max_slots(MODE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ys = smem;                       // [KP][kYPitch]
    char* Xs = smem + a.KP * kYPitch;      // [WR*WCOLS][pixb]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1;
    const int wt = __builtin_amdgcn_readfirstlane(wave >> 1);  // MODE 0: filter row; MODE 1: tap parity

    int bid = blockIdx.x;
    const int tile_c = bid % a.c_tiles;
    bid /= a.c_tiles;
    const int tile_n = bid % a.n_tiles;
    const int split = bid / a.n_tiles;
    const int n0 = tile_n * 64, c0 = tile_c * 32;

    const int chunk_begin = split * a.chunks_per_split;
    int chunk_end = chunk_begin + a.chunks_per_split;
    if (chunk_end > a.total_chunks) chunk_end = a.total_chunks;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, a.x_bytes);
    const int cpp_mask = (1 << a.cpp_log2) - 1;

    // Everything about a staging slot that does not depend on the chunk is computed ONCE: the dY lane offset
    // (the chunk only moves the buffer descriptor, whose record count also cuts the ragged last rows of an
    // image), and for a window slot its row inside the window, its byte offset relative to the window origin
    // and whether its column lies in the image.  Per chunk a window slot then costs an add, two compares and a
    // select instead of two divisions and a chain of multiplies.
    uint32_t dyrel[kMaxSlots], xrel[kMaxSlots];
    int xwr[kMaxSlots];
    bool xcol_ok[kMaxSlots];
#pragma unroll
    for (int s = 0; s < kMaxSlots; ++s) {
        const int id = tid + s * NT;
        dyrel[s] = (uint32_t)(((id >> 3) * a.N + (id & 7) * 8) * 2);
        const int j = tid + (s - a.sd) * NT;
        const int cc = j & cpp_mask, pix = j >> a.cpp_log2;
        const int wr = (int)fdiv((uint32_t)(pix < 0 ? 0 : pix), a.div_wcols);
        const int ws = pix - wr * a.WCOLS - a.pw;
        xwr[s] = wr;
        xcol_ok[s] = j >= 0 && j < a.nch_x && (unsigned)ws < (unsigned)a.Ws;
        xrel[s] = (uint32_t)(((wr * a.Ws + ws) * a.xpix + c0 + cc * 8) * 2);  // modular when ws < 0
    }

    bf16x8 regs[kMaxSlots];
    auto stage_load = [&](int chunk) {
        const int b = (int)fdiv((uint32_t)chunk, a.div_cpi);
        const int ho0 = (chunk - b * a.chunks_per_img) * a.R;
        int rows = a.Ho - ho0;
        if (rows > a.R) rows = a.R;
        const int pvalid = rows * a.Wo;
        const long dy_base = (long)((b * a.Ho + ho0) * a.Wo) * a.N + n0;
        // offsets row * N + nc * 8 (+8) of rows < pvalid end before pvalid * N - n0 because the tile fits (N % 64 == 0)
        const __amdgpu_buffer_rsrc_t rdy =
            make_rsrc(reinterpret_cast<const char*>(dy) + dy_base * 2, (uint32_t)((pvalid * a.N - n0) * 2));
        const int hs0 = ho0 * a.sh - a.ph;
        const uint32_t xbase = (uint32_t)(((b * a.Hs + hs0) * a.Ws) * a.xpix * 2);  // modular when hs0 < 0
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s) {
            if (s < a.sd) {
                regs[s] = buffer_load_bf16x8(rdy, dyrel[s]);
            } else if (s < a.sd + a.sx) {
                const bool ok = xcol_ok[s] && (unsigned)(hs0 + xwr[s]) < (unsigned)a.Hs;
                regs[s] = buffer_load_bf16x8(rx, ok ? xbase + xrel[s] : kOOB);
            }
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int s = 0; s < kMaxSlots; ++s) {
            if (s < a.sd) {
                const int id = tid + s * NT;
                const int row = id >> 3, nc = id & 7;
                if (row < a.KP) *reinterpret_cast<bf16x8*>(Ys + row * kYPitch + nc * 16) = regs[s];
            } else if (s < a.sd + a.sx) {
                const int j = tid + (s - a.sd) * NT;
                if (j < a.nch_x) {
                    const int cc = j & cpp_mask, pix = j >> a.cpp_log2;
                    *reinterpret_cast<bf16x8*>(Xs + pix * a.pixb + cc * 16) = regs[s];
                }
            }
        }
    };

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;

    // transposing-read lane roles: 16 lanes cover 4 K rows (q) x 4 column quads (p); lanes 16-31 the next
    // 16 columns; lanes 32-63 K rows +8
    const int q = (lane & 15) >> 2, p = lane & 3, h16 = (lane >> 4) & 1, h32 = lane >> 5;
    const int ncolb = (wn * 32 + h16 * 16 + p * 4) * 2;
    const int ccolb = (h16 * 16 + p * 4) * 2;
    const int ksteps = a.KP >> 4;
    // window byte offset of the two K rows this lane reads in each 16-row step: the same for every chunk
    int xp0[kMaxKSteps], xp1[kMaxKSteps];
#pragma unroll
    for (int kk = 0; kk < kMaxKSteps; ++kk) {
        const int row0 = kk * 16 + 8 * h32 + q, row1 = row0 + 4;
        int pix0 = 0, pix1 = 0;
        if (row0 < a.P) {
            const int r = (int)fdiv((uint32_t)row0, a.div_wo);
            pix0 = r * a.sh * a.WCOLS + (row0 - r * a.Wo) * a.sw;
        }
        if (row1 < a.P) {
            const int r = (int)fdiv((uint32_t)row1, a.div_wo);
            pix1 = r * a.sh * a.WCOLS + (row1 - r * a.Wo) * a.sw;
        }
        xp0[kk] = pix0 * a.pixb + ccolb;
        xp1[kk] = pix1 * a.pixb + ccolb;
    }

    if (chunk_begin < chunk_end) stage_load(chunk_begin);
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        __syncthreads();  // the previous chunk's fragment reads are done
        stage_store();
        __syncthreads();
        if (chunk + 1 < chunk_end) stage_load(chunk + 1);  // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < kMaxKSteps; ++kk) {
            if (kk < ksteps) {
                const int row0 = kk * 16 + 8 * h32 + q;
                const bf16x8 af = tr_pair(Ys + row0 * kYPitch + ncolb, Ys + (row0 + 4) * kYPitch + ncolb);
                const char* xb0 = Xs + xp0[kk];
                const char* xb1 = Xs + xp1[kk];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int t = MODE == 0 ? wt * 3 + i : wt + i * 2;
                    if (t < a.T) {
                        // tapoff[0] = LDS bytes per window row, tapoff[1] = bytes per window pixel
                        const int toff = MODE == 0 ? wt * a.tapoff[0] + i * a.tapoff[1] : t * a.tapoff[0];
                        const bf16x8 bfr = tr_pair(xb0 + toff, xb1 + toff);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // partial tile -> slab (lane = column: 128-B contiguous segments per store instruction)
    const int col = lane & 31;
    const int nb = n0 + wn * 32 + 4 * h32;
    if constexpr (MODE == 0) {
        float* out = slab + (long)split * a.N * a.T * a.C;
        const int c = c0 + col;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wt * 3 + i;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int n = nb + (k & 3) + 8 * (k >> 2);
                out[((long)n * a.T + t) * a.C + c] = acc[i][k];
            }
        }
    } else {
        // column = (kw slot 0..7, channel 0..3) of filter row t; slot 0 and channel 3 are padding.
        // slab[split][n][kh][kw][3] with kw = slot - 1
        float* out = slab + (long)split * a.N * a.T * (a.T * 3);
        const int slot = col >> 2, ch = col & 3;
        if (slot >= 1 && ch < 3) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wt + i * 2;
                if (t < a.T) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int n = nb + (k & 3) + 8 * (k >> 2);
                        out[(((long)n * a.T + t) * a.T + (slot - 1)) * 3 + ch] = acc[i][k];
                    }
                }
            }
        }
    }
}

inline int ceil16(int v) { return (v + 15) & ~15; }

// fills R / splits; false when the shape does not fit the staging budget
bool plan(WinGeom& g, WinArgs& a) {
    const int T = g.KH * g.KW;
    if (g.mode == 0) {
        if (g.KH != 3 || g.KW != 3 || g.C % 32 || g.N % 64 || g.xpix != g.C || (g.sw != 1 && g.sw != 2)) return false;
    } else {
        if (g.KW != 1 || g.KH != 7 || g.N % 64 || g.xpix != 8 || g.sw != 1) return false;
    }
    const int wc_elems = g.mode == 0 ? 32 : 8;          // elements per window pixel in LDS
    const int cpp = wc_elems / 8;
    const int pixb = g.mode == 0 ? (g.sw == 1 ? 64 : 96) : 16;  // 4 consecutive K rows -> 4 distinct 64-B bank quarters
    const int kwspan = g.mode == 0 ? g.KW : 4;           // source pixels a tap row spans (stem: 32 cols / 8)
    const int wcols = (g.Wo - 1) * g.sw + kwspan;
    const int nt = g.mode == 0 ? 384 : 256;
    int best = 0;
    for (int R = 1; R <= g.Ho; ++R) {
        const int P = R * g.Wo, KP = ceil16(P);
        if (KP > kMaxKP) break;
        const int WR = (R - 1) * g.sh + g.KH;
        const int nch_x = WR * wcols * cpp;
        const int slots = (KP * 8 + nt - 1) / nt + (nch_x + nt - 1) / nt;
        const long lds = (long)KP * kYPitch + (long)WR * wcols * pixb;
        if (slots <= max_slots(g.mode) && lds <= 48 * 1024) best = R;
    }
    if (best == 0) return false;
    g.R = best;
    std::memset(&a, 0, sizeof(a));
    a.B = g.B; a.Ho = g.Ho; a.Wo = g.Wo; a.N = g.N; a.Hs = g.Hs; a.Ws = g.Ws; a.xpix = g.xpix;
    a.C = g.C; a.T = T; a.sh = g.sh; a.sw = g.sw; a.ph = g.ph; a.pw = g.pw;
    a.R = g.R; a.P = g.R * g.Wo; a.KP = ceil16(a.P);
    a.WR = (g.R - 1) * g.sh + g.KH; a.WCOLS = wcols;
    a.cpp_log2 = cpp == 4 ? 2 : 0; a.pixb = pixb;
    a.nch_x = a.WR * wcols * cpp;
    a.sd = (a.KP * 8 + nt - 1) / nt; a.sx = (a.nch_x + nt - 1) / nt;
    a.chunks_per_img = (g.Ho + g.R - 1) / g.R;
    a.total_chunks = g.B * a.chunks_per_img;
    a.n_tiles = g.N / 64;
    a.c_tiles = g.mode == 0 ? g.C / 32 : 1;
    const int tiles = a.n_tiles * a.c_tiles;
    int splits = (768 + tiles - 1) / tiles;   // ~3 workgroups per CU
    if (splits > a.total_chunks) splits = a.total_chunks;
    if (splits < 1) splits = 1;
    a.chunks_per_split = (a.total_chunks + splits - 1) / splits;
    g.splits = (a.total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.tapoff[0] = wcols * pixb;
    a.tapoff[1] = pixb;
    a.div_wo = make_fastdiv((uint32_t)g.Wo);
    a.div_wcols = make_fastdiv((uint32_t)wcols);
    a.div_cpi = make_fastdiv((uint32_t)a.chunks_per_img);
    a.x_bytes = (uint32_t)((long)g.B * g.Hs * g.Ws * g.xpix * 2);
    a.dy_bytes = (uint32_t)((long)g.B * g.Ho * g.Wo * g.N * 2);
    return true;
}

template <int MODE>
void launch_mode(const void* dy, const void* x, float* slab, const WinGeom& g, const WinArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.KP * kYPitch + (size_t)a.WR * a.WCOLS * a.pixb;
    static bool attr_set = false;
    if (!attr_set) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_win<MODE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
        attr_set = true;
    }
    const int grid = a.n_tiles * a.c_tiles * g.splits;
    hipLaunchKernelGGL((k_wgrad_win<MODE>), dim3(grid), dim3(MODE == 0 ? 384 : 256), lds, s, (const bf16*)dy, (const bf16*)x, slab, a);
}

}  // namespace

bool wgrad_win_plan(WinGeom& g) {
    WinArgs a;
    return plan(g, a);
}

void launch_wgrad_win(const void* dy, const void* x, float* slab, const WinGeom& g_in, hipStream_t s) {
    WinGeom g = g_in;
    WinArgs a;
    if (!plan(g, a) || g.splits != g_in.splits) {
        fprintf(stderr, "launch_wgrad_win: geometry was not planned with wgrad_win_plan\n");
        abort();
    }
    if (g.mode == 0) launch_mode<0>(dy, x, slab, g, a, s);
    else launch_mode<1>(dy, x, slab, g, a, s);
}

}  // namespace pdt
