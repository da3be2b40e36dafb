// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (ResNet layer1's conv2: forward and data
// gradient, 3.3 ms of the 91 ms step at B = 1280), SURVEY 2.4 N5.
//
// The implicit-GEMM kernel fetches every input pixel nine times through the L2 -> LDS path (one DMA per tap) and
// the 64 x 576 weights once per tile: 369 KB staged per 256 x 64 output tile, ~40 GB/s per CU, which is what that
// path delivers -- the layer ran at 470-540 TFLOP/s with the matrix cores 15 % busy.  Here:
//  * the WEIGHTS live in registers for the whole kernel: 36 K-steps (9 taps x 4 blocks of 16 channels) x 2 halves of
//    the output channels = 72 MFMA operands = 288 VGPRs per lane.  One wave per SIMD (256 threads, 1 workgroup per
//    CU) has 512 registers: 256 + AGPRs for the accumulators;
//  * the INPUT is staged once per tile as a window with its halo: 4 output rows x W pixels need 6 x (W + 2) pixels
//    (44.5 KB at W = 56, 1.5x the tile instead of 9x), double-buffered by LDS-DMA for the next tile while this one
//    computes; the nine taps are nine shifted fragment reads of the same window (XOR swizzle per window pixel);
//  * MFMA 32x32x16 with the weights as the A operand: D[channel][pixel], so a lane ends up with 4 consecutive
//    channels of one pixel per accumulator group -> packed 8-byte writes into the staging tile, then a coalesced
//    16-B-per-lane store pass that also takes the per-channel BatchNorm sums (forward: sum, sum of squares; data
//    gradient: the two BatchNorm-backward sums of the gated result) over ALL tiles of the persistent workgroup:
//    one slab row per workgroup.
// Tiles are (image, block of 4 rows): whole rows, so a tile's output pixels are contiguous in memory.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int WTH = 4;                    // output rows per tile
constexpr int WIN_PX_MAX = 348;           // (WTH + 2) x (W + 2) window pixels at W = 56
constexpr int WIN_BYTES = 45056;          // ... x 128 B, rounded up to whole 1 KiB DMA pieces (8 pixels)
constexpr int SPITCH = 144;               // staged output row pitch (bytes): conflict-free 8-B writes, 16-B aligned reads
constexpr int STAGE_OFF = 2 * WIN_BYTES;
constexpr int STAGE_BYTES = 256 * SPITCH;           // 36 864
constexpr int WIN_LDS = STAGE_OFF + 2 * STAGE_BYTES;  // 163 840 B: all of the CU's LDS
constexpr int NPASS = 7;                  // store passes of 32 pixels per tile (WTH * W <= 224)

struct Win64Args {
    const void* x;       // [B][H][W][64] bf16
    const void* w;       // [64 out][9][64 in] bf16 (data gradient: the transposed weights [Cin][9][Cout])
    void* y;             // [B][H][W][64]
    float* slab;         // [grid][2][64] or null
    const void* bnr_x;   // data gradient: input of the BatchNorm whose backward sums are taken, or null
    const uint8_t* bnr_mask;
    const float* bnr_mean;
    const float* bnr_invstd;
    int B, H, W, mirror;  // mirror: tap (r, s) reads window row / column (2 - r, 2 - s) (data gradient)
    int tiles_per_image, tiles;
    FastDiv div_w, div_tpi;
};

// a pixel is 128 B = half of the 64 banks: 16 consecutive pixels (one quarter of a fragment read) are 8 of each
// parity, and the 8 of one parity must take the 8 different 16-B slots of their half -> key (pixel / 2) % 8
__device__ __forceinline__ int win_key(int wp) { return (wp >> 1) & 7; }
__device__ __forceinline__ int win_addr(int wp, int chunk) { return wp * 128 + ((chunk ^ win_key(wp)) << 4); }

template <bool BNR>
__global__ __launch_bounds__(256, 1) void k_conv_win64(const Win64Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = a.W + 2, tile_px = WTH * a.W, win_px = (WTH + 2) * W2;
    // row tiles of 32 pixels: at most 7 real ones; wave w owns 2w and 2w + 1

    // ---- weights -> registers: A operand of step (tap, kc): lane holds channel nt*32 + lane%32, inputs kc*16 + 8*(lane/32) + 0..7
    bf16x8 wf[36][2];
    {
        const bf16* wp = static_cast<const bf16*>(a.w);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    wf[t * 4 + kc][nt] = *reinterpret_cast<const bf16x8*>(wp + ((nt * 32 + (lane & 31)) * 9 + t) * 64 + kc * 16 + 8 * (lane >> 5));
    }

    // ---- window DMA: piece pc = wave + 4 * i covers window pixels 8 pc .. 8 pc + 7; lane -> pixel 8 pc + lane / 8, chunk
    // position lane % 8 holding logical chunk (lane % 8) ^ win_key(pixel).  (The per-lane plan is recomputed per tile: a
    // division and a few integer operations per piece, instead of 22 registers next to the 288 of the weights.)
    const uint32_t lds0 = lds_address(smem);
    const int n_pieces = (win_px + 7) >> 3;
    // one piece (index i of this wave: window piece wave + 4 i) of the window of `tile` into buffer `buf`.  Called
    // from inside the MFMA loop of the previous tile, one piece per step: with one wave per SIMD the address
    // arithmetic only overlaps the matrix pipe if it sits between the MFMAs.  A piece index past the window repeats
    // the last piece (same bytes to the same place); a tile index past the end sends every lane out of range.
    auto issue_piece = [&](int tile, int buf, int i) {
        const uint32_t b = fdiv((uint32_t)tile, a.div_tpi);
        const int y0 = (tile - (int)b * a.tiles_per_image) * WTH;
        // descriptor based one row and one pixel ahead of the tile's first pixel (the halo): window pixel (wy, wx) is
        // byte (wy * W + wx) * 128 of it; lanes outside the image are masked
        const long first = ((long)b * a.H + y0) * a.W - (a.W + 1);
        const i32x4 rsrc = make_rsrc_words(static_cast<const char*>(a.x) + first * 128, (uint32_t)((WTH + 2) * a.W + 2) * 128u);
        int pc = wave + 4 * i;
        pc = pc < n_pieces ? pc : n_pieces - 1;
        const int wp = pc * 8 + (lane >> 3);
        const int wy = (int)fdiv((uint32_t)wp, a.div_w /* / (W + 2) */);
        const int wx = wp - wy * W2;
        const int yy = y0 - 1 + wy;
        const bool ok = tile < a.tiles && wp < win_px && wx >= 1 && wx <= a.W && yy >= 0 && yy < a.H;
        const int lch = (lane & 7) ^ win_key(wp);
        lds_dma16(rsrc, ok ? (uint32_t)((wy * a.W + wx) * 128 + lch * 16) : kOOB,
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)(buf * WIN_BYTES + pc * 1024))));
    };
    constexpr int PIECES_PER_WAVE = (WIN_BYTES / 1024 + 3) / 4;  // 11

    // ---- fragment read bases (tile-invariant): lane's pixel of row tile rt is pi = rt*32 + lane%32 -> window pixel of tap (0, 0)
    int wp0[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int pi = (wave * 2 + q) * 32 + (lane & 31);
        if (pi >= tile_px) pi = 0;  // dummy lanes read a valid pixel; their results are never stored
        const int py = pi / a.W, px = pi - py * a.W;
        wp0[q] = py * W2 + px;
    }
    const int khalf = lane >> 5;

    // store pass: thread -> (pixel = pass * 32 + tid / 8, chunk = tid % 8)
    const int och = tid & 7, orow = tid >> 3;
    float csum[8], csq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        csum[k] = 0.f;
        csq[k] = 0.f;
    }

    // One pass (32 pixels x 128 B) of the coalesced store of tile `pt` from staging buffer `sb`; the tile's pixels are
    // contiguous in memory.  Every thread issues exactly one store per call (out-of-range ones are dropped), so the
    // counted wait at the top of a tile knows how many operations follow the window DMA.  Called from inside the
    // MFMA loop of the NEXT tile (one wave per SIMD: only what sits between the MFMAs overlaps the matrix pipe).
    auto store_pass = [&](int pt, int sb, int p) {
        const bool live = pt >= 0;
        const uint32_t b = fdiv((uint32_t)(live ? pt : 0), a.div_tpi);
        const int y0 = ((live ? pt : 0) - (int)b * a.tiles_per_image) * WTH;
        const int rows = a.H - y0 < WTH ? a.H - y0 : WTH;
        const int valid_px = live ? rows * a.W : 0;
        const long pix0 = ((long)b * a.H + y0) * a.W;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(static_cast<char*>(a.y) + pix0 * 128, (uint32_t)valid_px * 128u);
        const char* Os = smem + STAGE_OFF + sb * STAGE_BYTES;
        const int pix = p * 32 + orow;
        const bool in = pix < valid_px;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(Os + pix * SPITCH + och * 16);
        if (BNR) {
            // sum g and sum g * x here; mean / invstd enter once, at the end: sum g * xhat = invstd * (sum g x - mean * sum g)
            // (16 registers the weights need; the fp32 cancellation is that of ~500 terms per thread, |mean| / std * 1e-7)
            bf16x8 xv;  // (zeros where nothing is stored: the staging buffer may hold anything there, 0 * NaN is NaN)
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = (bf16)0.f;
            unsigned bits = 0u;
            if (in) {
                xv = *reinterpret_cast<const bf16x8*>(static_cast<const bf16*>(a.bnr_x) + (pix0 + pix) * 64 + och * 8);
                bits = a.bnr_mask[(pix0 + pix) * 8 + och];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gg = ((bits >> k) & 1u) ? (float)v[k] : 0.f;
                csum[k] += gg;
                csq[k] += gg * (float)xv[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = in ? (float)v[k] : 0.f;
                csum[k] += f;
                csq[k] += f * f;
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, in ? (uint32_t)(pix * 128 + och * 16) : kOOB, 0, kAuxStream);
    };

    int tile = blockIdx.x;
    if (tile < a.tiles) {
#pragma unroll
        for (int i = 0; i < PIECES_PER_WAVE; ++i) issue_piece(tile, 0, i);
    }
    int buf = 0, sbuf = 0, prev = -1;
    bool first_tile = true;
    static_assert(NPASS == 7 && PIECES_PER_WAVE + NPASS <= 18, "the counted wait below; both fit the 18 steps of a tile");
    for (; tile < a.tiles; tile += gridDim.x, buf ^= 1, sbuf ^= 1) {
        // This tile's window has landed: its DMA was issued in steps 0-3 of the previous tile's loop, the NPASS stores
        // of this thread in steps 11-17, and vmcnt retires in issue order (the stores may still be in flight).
        if (first_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        first_tile = false;
        // the one barrier of a tile: window visible to all waves; every wave is done with the other window buffer
        // (about to be refilled), has staged the previous tile (about to be stored) and has stored the one before
        // (its staging buffer is rewritten at the end of this tile)
        __builtin_amdgcn_s_barrier();
        const int next = tile + gridDim.x;

        const char* win = smem + buf * WIN_BYTES;
        // (the 72 fragment addresses are tile-invariant; left to itself the compiler hoists them out of the tile loop
        // into 72 registers that the weights need: make their base opaque per tile, the few integer operations per
        // read are free next to the MFMAs)
        int wq[2] = {wp0[0], wp0[1]};
        asm volatile("" : "+v"(wq[0]), "+v"(wq[1]));
        f32x16_t acc[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[q][nt][k] = 0.f;
        // software-pipelined by hand (one wave per SIMD: nothing else hides the LDS latency): 18 steps of half a tap
        // (2 blocks of 16 channels); the 4 fragment reads of step i + 2 are issued before the 8 MFMAs of step i
        constexpr int PFD = BNR ? 1 : 2;  // steps of read-ahead (the fused-sums variant has 16 registers less to spare)
        bf16x8 pf[PFD + 1][2][2];
        auto read_step = [&](int i, bf16x8 (&dst)[2][2]) {
            const int t = i >> 1, r = t / 3, s = t - r * 3;
            const int shift = a.mirror ? (2 - r) * W2 + (2 - s) : r * W2 + s;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    dst[k2][q] = *reinterpret_cast<const bf16x8*>(win + win_addr(wq[q] + shift, ((i & 1) * 2 + k2) * 2 + khalf));
        };
#pragma unroll
        for (int i = 0; i < PFD; ++i) read_step(i, pf[i]);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            if (i + PFD < 18) read_step(i + PFD, pf[(i + PFD) % (PFD + 1)]);
            // the next tile's window in the first steps (it needs the rest of the tile to arrive), the previous tile's
            // output, one pass per step, in the last ones
            if (i < 4) {
#pragma unroll
                for (int j = i * 3; j < i * 3 + 3 && j < PIECES_PER_WAVE; ++j) issue_piece(next, buf ^ 1, j);
            } else if (i >= 18 - NPASS) {
                store_pass(prev, sbuf ^ 1, i - (18 - NPASS));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[q][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i * 2 + k2][nt], pf[i % (PFD + 1)][k2][q], acc[q][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- accumulators -> staging tile [pixel][64 ch] bf16: lane holds pixel rt*32 + lane%32, channels
        // nt*32 + 8*g + 4*(lane/32) + 0..3 for g = 0..3.  (A row tile past the end of the tile -- the 8th of 7 -- is
        // computed and staged like the others, in rows the store pass never reads: no branches in the MFMA stream.)
        char* Os = smem + STAGE_OFF + sbuf * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pix = (wave * 2 + q) * 32 + (lane & 31);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 pk = {(bf16)acc[q][nt][g * 4 + 0], (bf16)acc[q][nt][g * 4 + 1], (bf16)acc[q][nt][g * 4 + 2],
                                       (bf16)acc[q][nt][g * 4 + 3]};
                    *reinterpret_cast<bf16x4*>(Os + pix * SPITCH + (nt * 32 + g * 8 + khalf * 4) * 2) = pk;
                }
        }
        prev = tile;
    }
    // the last tile's output
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < NPASS; ++p) store_pass(prev, sbuf ^ 1, p);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the surplus window DMA of the last tile must not outlive the block)
    // ---- per-workgroup sums -> slab[blockIdx][2][64]
    if (a.slab != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + STAGE_OFF);  // [32 rows][8 chunks][16]
        if (BNR) {
#pragma unroll
            for (int k = 0; k < 8; ++k) csq[k] = a.bnr_invstd[och * 8 + k] * (csq[k] - a.bnr_mean[och * 8 + k] * csum[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(orow * 8 + och) * 16 + k] = csum[k];
            red[(orow * 8 + och) * 16 + 8 + k] = csq[k];
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float s = 0.f;
            for (int rrow = 0; rrow < 32; ++rrow) s += red[(rrow * 8 + (c >> 3)) * 16 + which * 8 + (c & 7)];
            a.slab[(long)blockIdx.x * 128 + tid] = s;
        }
    }
}

bool win64_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("PDT_CONV_WIN64");
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}

}  // namespace

bool conv_win64_ok(int C, int N, int KH, int KW, int stride, int pad, int H, int W) {
    return win64_enabled() && C == 64 && N == 64 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && W >= 2 &&
           (WTH + 2) * (W + 2) <= WIN_PX_MAX && WTH * W <= NPASS * 32 && H >= 1;
}

int conv_win64_grid(int B, int H) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n;
    }();
    const long tiles = (long)B * ((H + WTH - 1) / WTH);
    return (int)(tiles < cus ? tiles : cus);
}

// slab: [conv_win64_grid(B, H)][2][64] fp32 or null; bnr_*: all four or none (data gradient with the fused BN-backward sums)
void launch_conv_win64(const void* x, const void* w, void* y, float* slab, int B, int H, int W, int mirror,
                       const void* bnr_x, const uint8_t* bnr_mask, const float* bnr_mean, const float* bnr_invstd,
                       hipStream_t s) {
    if (!conv_win64_ok(64, 64, 3, 3, 1, 1, H, W) || (long)B * ((H + WTH - 1) / WTH) >= (1L << 31) || (bnr_x && !slab)) {
        fprintf(stderr, "launch_conv_win64: unsupported geometry %d x %d x %d\n", B, H, W);
        abort();
    }
    Win64Args a;
    a.x = x; a.w = w; a.y = y; a.slab = slab;
    a.bnr_x = bnr_x; a.bnr_mask = bnr_mask; a.bnr_mean = bnr_mean; a.bnr_invstd = bnr_invstd;
    a.B = B; a.H = H; a.W = W; a.mirror = mirror;
    a.tiles_per_image = (H + WTH - 1) / WTH;
    a.tiles = B * a.tiles_per_image;
    a.div_w = make_fastdiv((uint32_t)(W + 2));
    a.div_tpi = make_fastdiv((uint32_t)a.tiles_per_image);
    const int grid = conv_win64_grid(B, H);
    static bool attr_set = false;
    if (!attr_set) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_win64<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS));
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_win64<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS));
        attr_set = true;
    }
    if (bnr_x) hipLaunchKernelGGL((k_conv_win64<true>), dim3(grid), dim3(256), WIN_LDS, s, a);
    else hipLaunchKernelGGL((k_conv_win64<false>), dim3(grid), dim3(256), WIN_LDS, s, a);
}

}  // namespace pdt
