// Weight gradient of the 3x3 "same" convolution on MFMA (autodiff of Conv2D, KerasLayers.py:683,689,758):
//   dW[tap][ci][co] = sum over pixels p of X[p + off(tap)][ci] * dY[p][co]
// i.e. per tap a GEMM with M = ci, N = co and the PIXELS as the contraction dimension.
//
// One workgroup (4 waves) owns a (32 ci x 32 co) block of all 9 taps = 9 accumulator tiles per wave and
// walks pixel tiles (TH x TW = 256 pixels) split = blockIdx.x, split + nsplit, ...  Each wave contracts
// 64 of the tile's pixels.  Operands are staged NHWC ([pixel][32 channels]) in LDS:
//   bf16: the MFMA wants 8 consecutive k (= pixels) per lane for a fixed channel, which in NHWC is a
//         16-bit gather across rows -> ds_read_b64_tr_b16 (hardware 4x16 transpose read); with 64-byte
//         pixel rows the 4 pixels x 32 channels a half-wave reads are 256 contiguous bytes (no conflict);
//   f32 : v_mfma_f32_32x32x2_f32 takes one float per lane: lane = channel, plain ds_read_b32.
// The four waves' accumulators are folded through LDS in a fixed order, each workgroup writes one fp32
// slab [9][Cin][Cout], and a second kernel sums the slabs in split order: bitwise reproducible.
#include "rvip_common.h"
#include <atomic>
#include <type_traits>
#include <cstdlib>

namespace rvip {

struct WgArgs {
    const unsigned char* x0; const unsigned char* x1; const unsigned char* dy;
    float* slab;
    int c0, c1, up0;
    int n, h, w, cin, cout;
    int tiles_x, tiles_y, ntiles, nsplit;
    int zs;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

template <typename T, int TW>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(WgArgs a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HHT = TH + 2, NHALO = HWD * HHT;
    constexpr int VE = Vec<T>::VE;
    constexpr int ROWB = 32 * (int)sizeof(T);          // LDS bytes per pixel (32 channels)
    constexpr int PPR = ROWB / 16;                      // 16-byte pieces per pixel row
    constexpr int NXP = (NHALO * PPR + 255) / 256;
    constexpr int LDS_X = NHALO * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lx = smem;
    unsigned char* lg = smem + LDS_X;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, hf = lane >> 5;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int h0 = a.h >> a.up0, w0 = a.w >> a.up0;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = split; tile < a.ntiles; tile += a.nsplit) {
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int n = bx / a.tiles_y;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        __syncthreads();                                           // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int id = tid + 256 * i;
            if (id < NHALO * PPR) {
                const int hp = id / PPR, part = id % PPR;
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
                const int c = ci0 + part * VE;
                uint4 r = make_uint4(0, 0, 0, 0);
                if ((unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w) {
                    if (c < a.c0) {
                        const size_t pix = ((size_t)n * h0 + (gy >> a.up0)) * w0 + (gx >> a.up0);
                        if (!a.zs || ((gy & gx) & 1)) r = *reinterpret_cast<const uint4*>(a.x0 + (pix * a.c0 + c) * sizeof(T));
                    } else if (c < a.cin) {
                        const size_t pix = ((size_t)n * a.h + gy) * a.w + gx;
                        r = *reinterpret_cast<const uint4*>(a.x1 + (pix * a.c1 + (c - a.c0)) * sizeof(T));
                    }
                }
                *reinterpret_cast<uint4*>(lx + hp * ROWB + part * 16) = r;
            }
        }
#pragma unroll
        for (int i = 0; i < PPR; ++i) {
            const int id = tid + 256 * i;
            const int P = id / PPR, part = id % PPR;
            const int gy = ty0 + P / TW, gx = tx0 + P % TW;
            const int c = co0 + part * VE;
            uint4 r = make_uint4(0, 0, 0, 0);
            if (gy < a.h && gx < a.w && c < a.cout) {
                const size_t pix = ((size_t)n * a.h + gy) * a.w + gx;
                r = *reinterpret_cast<const uint4*>(a.dy + (pix * a.cout + c) * sizeof(T));
            }
            *reinterpret_cast<uint4*>(lg + P * ROWB + part * 16) = r;
        }
        __syncthreads();

        if constexpr (sizeof(T) == 2) {
            // lane (group G = lane>>4, i = lane&15 = 4q+p) supplies row q / columns 4p.. of its group's
            // 4-pixel x 16-channel block; it receives channel (lane&31), pixels 8*hf + 4u + {0..3}
            const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, grp = lane >> 4;
            const int kk = 8 * (grp >> 1) + q;
            const int cb = (16 * (grp & 1) + 4 * p4) * 2;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int gaddr[2], xaddr[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int P = wv * 64 + s * 16 + kk + 4 * u;
                    gaddr[u] = P * ROWB + cb;
                    xaddr[u] = ((P / TW) * HWD + (P % TW)) * ROWB + cb;
                }
                const s16x4 g0 = tr_read(lg + gaddr[0]);
                const s16x4 g1 = tr_read(lg + gaddr[1]);
                const uint4 fb = __builtin_bit_cast(uint4, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int toff = ((t / 3) * HWD + (t % 3)) * ROWB;
                    const s16x4 x0 = tr_read(lx + xaddr[0] + toff);
                    const s16x4 x1 = tr_read(lx + xaddr[1] + toff);
                    const uint4 fa = __builtin_bit_cast(uint4, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[t] = mfma16<T>(fa, fb, acc[t]);
                }
            }
        } else {
#pragma unroll 2
            for (int s = 0; s < 32; ++s) {
                const int P = wv * 64 + s * 2 + hf;
                const float g = *reinterpret_cast<const float*>(lg + P * ROWB + j * 4);
                const int xb = ((P / TW) * HWD + (P % TW)) * ROWB + j * 4;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float x = *reinterpret_cast<const float*>(lx + xb + ((t / 3) * HWD + (t % 3)) * ROWB);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, g, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // fold the four waves in a fixed order, then one coalesced slab write
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                  // [9][32 ci][32 co]
    for (int wsel = 0; wsel < 4; ++wsel) {
        if (wv == wsel) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * hf;
                    const int idx = (t * 32 + ci) * 32 + j;
                    red[idx] = (wsel == 0 ? 0.f : red[idx]) + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = a.slab + (size_t)split * 9 * a.cin * a.cout;
    for (int e = tid; e < 9 * 32 * 32; e += 256) {
        const int t = e >> 10, ci = ci0 + ((e >> 5) & 31), co = co0 + (e & 31);
        if (ci < a.cin && co < a.cout) out[((size_t)t * a.cin + ci) * a.cout + co] = red[e];
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad v2: LDS-DMA staging (buffer_load ... lds, zeros for out-of-range lanes = halo / tile overhang /
// channel tails), two stages so the next pixel tile streams in while the current one is contracted, and a
// (CIB x COB) block of (ci, co) per workgroup: with 64 x 64 the four waves own one 32 x 32 pair each and
// contract all 256 pixels of the tile (no cross-wave fold); with fewer pairs the waves split the pixels and
// are folded through LDS in a fixed order.  Rows are 64 B (32 bf16 / 16 f32 channels... see RB) or 128 B;
// 128-byte rows store their two 64-byte halves swapped when (row >> 1) & 1 (source-side swizzle) so that the
// 4 consecutive pixel rows a transposed read touches fall on 4 different 64-byte bank segments.
// ---------------------------------------------------------------------------------------------
typedef int i32x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4w make_rsrc_w(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4w r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16w(i32x4w rsrc, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}

__device__ __forceinline__ void dma16w_nt(i32x4w rsrc, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}

struct WgArgs2 {
    const unsigned char* x0; const unsigned char* x1; const unsigned char* dy;
    float* slab;
    unsigned x0_bytes, x1_bytes, dy_bytes;
    int c0, c1, up0;
    int n, h, w, cin, cout;
    int tiles_x, tiles_y, ntiles, nsplit;
    int zs;
    int depth, dshift;                        // Conv3D depth tap: X is read from image n + dshift of the same volume (zeros outside)
    int dbg;                                  // ablation only (RVIP_DBG): 1 = no DMA after the first tile, 2 = no MFMA, 4 = DMAs fetch nothing
    int nt_slab;                              // slabs leave with the non-temporal hint (deferred fold: their reader runs milliseconds later)
    int nt_x;                                 // X tiles are fetched with the non-temporal hint: the activations of the forward pass are read here for the last time,
                                              // and the gradient tensor the data gradient reads next stays cached (same box: 4.836 -> 4.815, 4.947 -> 4.894 ms per step;
                                              // the same hint on dY, or on X of only the one-block / only the many-block layers, measured slower)
    int sp;                                   // sub-pixel form of the up-sampled layer (wgrad3x3_ws<..., TAPS = 4>): h, w = the low-resolution grid, nsplit = 4 phases x pixel splits
};

template <typename T, int TW, int CIB, int COB>
__global__ __launch_bounds__(256, 1) void wgrad3x3_dma(WgArgs2 a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HHT = TH + 2, NHALO = HWD * HHT;
    constexpr int NHROWS = (NHALO + 15) / 16 * 16;
    constexpr int ESZ = (int)sizeof(T), VE = Vec<T>::VE;
    constexpr int RBX = CIB * ESZ, RBG = COB * ESZ;                 // row bytes: 64 or 128
    static_assert((RBX == 64 || RBX == 128) && (RBG == 64 || RBG == 128), "row bytes");
    constexpr int X_BYTES = NHROWS * RBX, G_BYTES = 256 * RBG, ST_BYTES = X_BYTES + G_BYTES;
    constexpr int NQX = X_BYTES / 1024, NQG = G_BYTES / 1024;
    constexpr int QX = (NQX + 3) / 4, QG = (NQG + 3) / 4;
    constexpr int NPAIR = (CIB / 32) * (COB / 32) > 4 ? 4 : (CIB * ESZ / 64) * 0 + ((CIB / 32) * (COB / 32));
    constexpr int NCI = (ESZ == 2) ? CIB / 32 : 1, NCO = (ESZ == 2) ? COB / 32 : 1;   // 32-channel MFMA tiles per block
    constexpr int PAIRS = NCI * NCO;                                  // 1, 2 or 4
    constexpr int PSPLIT = 4 / PAIRS;
    constexpr unsigned OOB = 0x80000000u;
    (void)NPAIR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, ci0 = blockIdx.y * CIB, co0 = blockIdx.z * COB;
    const int h0 = a.h >> a.up0, w0 = a.w >> a.up0;
    const int pair = wv % PAIRS, part = wv / PAIRS;
    const int ci_t = pair % NCI, co_t = pair / NCI;

    const bool from0 = ci0 < a.c0;                                    // a block never straddles the concat (host)
    const i32x4w rsx = from0 ? make_rsrc_w(a.x0, a.x0_bytes) : make_rsrc_w(a.x1, a.x1_bytes);
    const i32x4w rsg = make_rsrc_w(a.dy, a.dy_bytes);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;

    // Tile-independent lane geometry of the DMA pieces this wave issues.  A piece's byte offset is
    //   tile_base (wave-uniform) + rel (per lane, precomputed)      -- also for the x2 reads: tiles start on even
    // coordinates, so ((ty0 - 1 + hy) >> 1) = ty0/2 + ((hy - 1) >> 1) -- and validity is a handful of compares
    // folded into one select (no branches in the per-tile issue code).
    constexpr int SLX = RBX / 16, RPPX = 1024 / RBX, SLG = RBG / 16, RPPG = 1024 / RBG;
    const int csrc = from0 ? a.c0 : a.c1, cb0 = from0 ? ci0 : ci0 - a.c0;
    const int hs = from0 ? h0 : a.h, wsrc = from0 ? w0 : a.w, shf = from0 ? a.up0 : 0;
    const bool zsx = from0 && a.zs;
    int xhy[QX], xhx[QX], xrel[QX], grel[QG], gpy[QG], gpx[QG];
#pragma unroll
    for (int i = 0; i < QX; ++i) {
        const int row = (wv + 4 * i) * RPPX + lane / SLX, slot = lane % SLX;
        const int hy = row / HWD, hx = row - hy * HWD;
        const int p = (RBX == 128) ? ((((slot >> 2) ^ ((hx >> 1) & 1)) << 2) | (slot & 3)) : slot;   // swizzle by the halo x coordinate
        const int c = cb0 + p * VE;
        const bool ok = row < NHALO && c < csrc;
        xhy[i] = ok ? hy - 1 : -100000;                           // a statically dead piece fails the range test below
        xhx[i] = hx - 1;
        xrel[i] = ((((hy - 1) >> shf) * wsrc + ((hx - 1) >> shf)) * csrc + c) * ESZ;
    }
#pragma unroll
    for (int i = 0; i < QG; ++i) {
        const int P = (wv + 4 * i) * RPPG + lane / SLG, slot = lane % SLG;
        const int p = (RBG == 128) ? ((((slot >> 2) ^ ((P >> 1) & 1)) << 2) | (slot & 3)) : slot;
        const int c = co0 + p * VE;
        gpy[i] = (c < a.cout) ? P / TW : -100000;
        gpx[i] = P % TW;
        grel[i] = (((P / TW) * a.w + (P % TW)) * a.cout + c) * ESZ;
    }
    auto issue = [&](int tile, int stage) __attribute__((always_inline)) {
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int n = bx / a.tiles_y;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        const bool dok = (unsigned)(n % a.depth + a.dshift) < (unsigned)a.depth;
        const int xbase = (((n + a.dshift) * hs + (ty0 >> shf)) * wsrc + (tx0 >> shf)) * csrc * ESZ;
        const int gbase = ((n * a.h + ty0) * a.w + tx0) * a.cout * ESZ;
#pragma unroll
        for (int i = 0; i < QX; ++i) {
            const int q = wv + 4 * i;
            if (q < NQX) {
                const int gy = ty0 + xhy[i], gx = tx0 + xhx[i];
                bool ok = dok && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
                if (zsx) ok = ok && ((gy & gx) & 1);
                const unsigned off = (ok && !(a.dbg & 4)) ? (unsigned)(xbase + xrel[i]) : OOB;
                dma16w(rsx, off, lds_base + stage * ST_BYTES + q * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < QG; ++i) {
            const int q = wv + 4 * i;
            if (q < NQG) {
                const int gy = ty0 + gpy[i], gx = tx0 + gpx[i];
                const bool ok = (unsigned)gy < (unsigned)a.h && gx < a.w;
                const unsigned off = (ok && !(a.dbg & 4)) ? (unsigned)(gbase + grel[i]) : OOB;
                dma16w(rsg, off, lds_base + stage * ST_BYTES + X_BYTES + q * 1024);
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (split < a.ntiles) issue(split, 0);
    int it = 0;
    for (int tile = split; tile < a.ntiles; tile += a.nsplit, ++it) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (tile + a.nsplit < a.ntiles && !(a.dbg & 1)) issue(tile + a.nsplit, (it + 1) & 1);
        if (a.dbg & 2) continue;
        const unsigned char* lx = smem + (it & 1) * ST_BYTES;
        const unsigned char* lg = lx + X_BYTES;
        if constexpr (ESZ == 2) {
            // Every transposed read of the tile = one of 6 (X) / 2 (G) per-lane base addresses + a compile-time offset:
            // the 64-byte-half swizzle of a 128-byte row depends only on bit 1 of the halo x coordinate, which is the
            // same for every k-step (k-steps start on multiples of 16 pixels) and every tap ROW; only the tap COLUMN
            // (tx = 0..2) changes it.  No address arithmetic is left between the MFMAs.
            constexpr int STEPS = 16 / PSPLIT;
            const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, grp = lane >> 4;
            const int kk = 8 * (grp >> 1) + q4;
            const int cb = (16 * (grp & 1) + 4 * p4) * 2;
            const int wave_px = part * STEPS * 16;                       // first pixel of this wave's share (multiple of 64)
            const unsigned char* xp[3][2];
            const unsigned char* gp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pxl = kk + 4 * u;                              // x within the k-step (0..15)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int hx = pxl + tx;
                    xp[tx][u] = lx + ((wave_px / TW) * HWD + hx) * RBX + ((RBX == 128) ? ((ci_t ^ ((hx >> 1) & 1)) << 6) : 0) + cb;
                }
                gp[u] = lg + (wave_px + pxl) * RBG + ((RBG == 128) ? ((co_t ^ ((pxl >> 1) & 1)) << 6) : 0) + cb;
            }
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                constexpr int dummy = 0; (void)dummy;
                const int srow = (s * 16) / TW, scol = (s * 16) % TW;     // compile-time after unrolling
                const s16x4 g0 = tr_read(gp[0] + s * 16 * RBG);
                const s16x4 g1 = tr_read(gp[1] + s * 16 * RBG);
                const uint4 fb = __builtin_bit_cast(uint4, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int off = ((srow + t / 3) * HWD + scol) * RBX;
                    const s16x4 x0 = tr_read(xp[t % 3][0] + off);
                    const s16x4 x1 = tr_read(xp[t % 3][1] + off);
                    const uint4 fa = __builtin_bit_cast(uint4, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[t] = mfma16<T>(fa, fb, acc[t]);
                }
            }
        } else {
            // f32: 32 channels per operand (128-byte rows), lane = channel, one pixel per half-wave
            constexpr int STEPS = 128 / PSPLIT;
            const int choff = (j & 15) * 4, chalf = j >> 4;
#pragma unroll 2
            for (int s = 0; s < STEPS; ++s) {
                const int P = (part * STEPS + s) * 2 + hf;
                const int py = P / TW, px = P % TW;
                const float g = *reinterpret_cast<const float*>(lg + P * RBG + ((chalf ^ ((px >> 1) & 1)) << 6) + choff);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hx = px + t % 3;
                    const int rr = (py + t / 3) * HWD + hx;
                    const float x = *reinterpret_cast<const float*>(lx + rr * RBX + ((chalf ^ ((hx >> 1) & 1)) << 6) + choff);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, g, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // results: one 9 x 32 x 32 fp32 block per (ci_t, co_t) pair; waves that split the pixels fold through LDS
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    float* out = a.slab + (size_t)split * 9 * a.cin * a.cout;
    if constexpr (PSPLIT == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + ci_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf, co = co0 + co_t * 32 + j;
                if (ci < a.cin && co < a.cout) out[((size_t)t * a.cin + ci) * a.cout + co] = acc[t][r];
            }
    } else {
        float* red = reinterpret_cast<float*>(smem) + pair * 9 * 32 * 32;   // [PAIRS][9][32][32]
        for (int psel = 0; psel < PSPLIT; ++psel) {
            if (part == psel) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int idx = (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + j;
                        red[idx] = (psel == 0 ? 0.f : red[idx]) + acc[t][r];
                    }
            }
            __syncthreads();
        }
        const float* redall = reinterpret_cast<const float*>(smem);
        for (int e = tid; e < PAIRS * 9 * 32 * 32; e += 256) {
            const int pr = e / (9 * 32 * 32), rem = e % (9 * 32 * 32);
            const int t = rem >> 10, ci = ci0 + (pr % NCI) * 32 + ((rem >> 5) & 31), co = co0 + (pr / NCI) * 32 + (rem & 31);
            if (ci < a.cin && co < a.cout) out[((size_t)t * a.cin + ci) * a.cout + co] = redall[e];
        }
    }
}

// wgrad v3: the same kernel with the producer / consumer split of igemm v3.  A workgroup is 8 waves: waves 0..3 contract
// (exactly the four waves of v2), waves 4..7 only issue the LDS-DMA pieces of the next pixel tile; one s_barrier per tile
// joins them.  In v2 every wave issues ~19 DMA instructions (~150 issue cycles each) in front of its 144 MFMAs per tile.
// NST = LDS stages: 3 where they fit (the 32 x 32 block variant, 38 KiB per stage - the full-resolution layers, which are bound
// by the latency of the input stream: a second tile in flight per CU), else 2.
// TAPS = 4: the weight gradient of UpSampling2D(2) -> conv in its sub-pixel form (16-bit types).  Output pixel (2i+pa, 2j+pb) of the
// up-sampled convolution only sees the low-resolution pixels (i+pa-1..i+pa) x (j+pb-1..j+pb), each through a SUM of the 3x3 taps
// that land on it (rvip_pack_subpixel_weights has the table), so per output phase (pa, pb) the gradient of those four summed taps is
// a 2x2-tap contraction of the low-resolution X with the phase image dY[2i+pa][2j+pb]: 16 instead of 36 multiply-adds per
// low-resolution pixel.  blockIdx.x = 4 * pixel split + phase; X tiles are read from the low-resolution tensor as they lie (a.h, a.w
// = ITS grid), dY tiles with stride 2.  d/dW[kh][kw] = the sum over the phases of the summed-tap gradient that contains (kh, kw) --
// exactly one per phase -- so every workgroup writes a full nine-tap slab (its four blocks repeated where they belong) and the slab
// fold stays what it is.
// PB = 1 (with TAPS = 4): a workgroup contracts BOTH column phases pb = 0, 1 of its row phase pa against one staged X tile -- two dY
// phase tiles per stage, two accumulator sets, their blocks added before the slab leaves.  A tile's DMA pieces need ~2.2 us to issue
// and land whatever the MFMA count (one tile in flight), so the form pays where the stage still fits two buffers: 64 x 32 blocks
// (X 45 KB + 2 x 16 KB), i.e. the full-resolution layer 64 -> 32 at 256^2, which the four-phase form could not speed up.
// (a device function of the workgroup's coordinates: bx_ = pixel split [x phase], by_ / bz_ = input / output channel block, gdx_ = the
//  grid's x extent -- rvip_pair.hip runs it in a part of the grid of its weight / data gradient pair kernel)
// TS = 2 (16-bit types): EIGHT compute waves -- every (pair, pixel part) of the four-wave form twice, once for
// taps 0..4 and once for taps 5..8 of the nine (80 / 64 accumulator registers instead of 144), 0..1 and 2..3 of the sub-pixel form's four -- so that the workgroup is 12 waves at <= 168 VGPRs, the shape of
// the eight-compute-wave igemm: the two can then be the two parts of one grid (rvip_pair.hip).  Both tap halves read the same dY
// fragments; the X fragments, the MFMA count and the slab are those of the four-wave form.
template <typename T, int TW, int CIB, int COB, int NST = 2, int TAPS = 9, int PB = 0, int TS = 1>
__device__ __forceinline__ void wgrad3x3_ws_body(const WgArgs2& a, const unsigned bx_, const unsigned by_, const unsigned bz_, const unsigned gdx_) {
    static_assert(TAPS == 9 || (TAPS == 4 && sizeof(T) == 2), "taps");
    static_assert(PB == 0 || TAPS == 4, "phase pairs belong to the sub-pixel form");
    static_assert(TS == 1 || (TS == 2 && sizeof(T) == 2), "tap halves: 16-bit types");
    constexpr int HT = (TAPS + 1) / 2;                                // taps of the first half (TS = 2): 5 of 9, 2 of 4
    constexpr int NCWV = 4 * TS;                                      // compute waves
    constexpr bool SP = TAPS == 4;
    constexpr int NG = PB ? 2 : 1;                                    // dY phase tiles per stage
    constexpr int TH = 256 / TW, HWD = TW + 2, HHT = TH + 2, NHALO = HWD * HHT;
    constexpr int NHROWS = (NHALO + 15) / 16 * 16;
    constexpr int ESZ = (int)sizeof(T), VE = Vec<T>::VE;
    constexpr int RBX = CIB * ESZ, RBG = COB * ESZ;                 // row bytes: 64 or 128
    static_assert((RBX == 64 || RBX == 128) && (RBG == 64 || RBG == 128), "row bytes");
    constexpr int X_BYTES = NHROWS * RBX, G_BYTES = 256 * RBG, ST_BYTES = X_BYTES + NG * G_BYTES;
    constexpr int NQX = X_BYTES / 1024, NQG = G_BYTES / 1024;
    constexpr int QX = (NQX + 3) / 4, QG = (NQG + 3) / 4;
    constexpr int NPAIR = (CIB / 32) * (COB / 32) > 4 ? 4 : (CIB * ESZ / 64) * 0 + ((CIB / 32) * (COB / 32));
    constexpr int NCI = (ESZ == 2) ? CIB / 32 : 1, NCO = (ESZ == 2) ? COB / 32 : 1;   // 32-channel MFMA tiles per block
    constexpr int PAIRS = NCI * NCO;                                  // 1, 2 or 4
    constexpr int PSPLIT = 4 / PAIRS;
    constexpr unsigned OOB = 0x80000000u;
    (void)NPAIR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
    const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv8 >= NCWV;
    const int wv = wv8 & 3;                                           // compute wave id within its tap half / loader wave id
    const int th = TS == 2 ? wv8 >> 2 : 0;                            // tap half of a compute wave (TS = 2: 0 -> taps 0..HT-1, 1 -> the rest)
    // Sub-pixel form: the four phases of a pixel split read the same X tiles and the four interleaved quarters of the same dY lines
    // (a pixel of a 32-channel dY is half a 128-byte line).  Workgroups go to the 8 XCDs round-robin, so with a grid of whole
    // groups of 32 the phases of split s are the workgroups 8 apart -- same XCD, same L2, running side by side; otherwise every line
    // of dY comes from memory twice (64 -> 32 at 256^2: 85 us against 74 for the nine-tap form it replaces).
    constexpr int NPHB = PB ? 1 : 2;                                  // log2 of the phases that are separate workgroups (PB: only pa)
    const bool xcd_map = SP && gdx_ % (8 << NPHB) == 0;
    const int ph = !SP ? 0 : xcd_map ? (int)((bx_ >> 3) & ((1 << NPHB) - 1)) : (int)(bx_ & ((1 << NPHB) - 1));
    const int split = !SP ? (int)bx_ : xcd_map ? (int)((bx_ & 7) | ((bx_ >> (3 + NPHB)) << 3)) : (int)(bx_ >> NPHB);
    const int ci0 = by_ * CIB, co0 = bz_ * COB;
    const int pa = PB ? ph : ph >> 1, pb = PB ? 0 : ph & 1;                                             // output phase (wave-uniform); PB: pb = 0 and 1
    const int nsplit = SP ? a.nsplit >> NPHB : a.nsplit;                                                // pixel splits
    const int h0 = a.h >> a.up0, w0 = a.w >> a.up0;
    const int pair = wv % PAIRS, part = wv / PAIRS;
    const int ci_t = pair % NCI, co_t = pair / NCI;

    const bool from0 = ci0 < a.c0;                                    // a block never straddles the concat (host)
    const i32x4w rsx = from0 ? make_rsrc_w(a.x0, a.x0_bytes) : make_rsrc_w(a.x1, a.x1_bytes);
    const i32x4w rsg = make_rsrc_w(a.dy, a.dy_bytes);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;

    // Tile-independent lane geometry of the DMA pieces this wave issues.  A piece's byte offset is
    //   tile_base (wave-uniform) + rel (per lane, precomputed)      -- also for the x2 reads: tiles start on even
    // coordinates, so ((ty0 - 1 + hy) >> 1) = ty0/2 + ((hy - 1) >> 1) -- and validity is a handful of compares
    // folded into one select (no branches in the per-tile issue code).
    constexpr int SLX = RBX / 16, RPPX = 1024 / RBX, SLG = RBG / 16, RPPG = 1024 / RBG;
    const int csrc = from0 ? a.c0 : a.c1, cb0 = from0 ? ci0 : ci0 - a.c0;
    const int hs = from0 ? h0 : a.h, wsrc = from0 ? w0 : a.w, shf = from0 ? a.up0 : 0;
    const bool zsx = from0 && a.zs;
    int xhy[QX], xhx[QX], xrel[QX], grel[QG], gpy[QG], gpx[QG];
#pragma unroll
    for (int i = 0; i < QX; ++i) {
        const int row = (wv + 4 * i) * RPPX + lane / SLX, slot = lane % SLX;
        const int hy = row / HWD, hx = row - hy * HWD;
        const int p = (RBX == 128) ? ((((slot >> 2) ^ ((hx >> 1) & 1)) << 2) | (slot & 3)) : slot;   // swizzle by the halo x coordinate
        const int c = cb0 + p * VE;
        const bool ok = row < NHALO && c < csrc;
        xhy[i] = ok ? hy - 1 : -100000;                           // a statically dead piece fails the range test below
        xhx[i] = hx - 1;
        xrel[i] = ((((hy - 1) >> shf) * wsrc + ((hx - 1) >> shf)) * csrc + c) * ESZ;
    }
#pragma unroll
    for (int i = 0; i < QG; ++i) {
        const int P = (wv + 4 * i) * RPPG + lane / SLG, slot = lane % SLG;
        const int p = (RBG == 128) ? ((((slot >> 2) ^ ((P >> 1) & 1)) << 2) | (slot & 3)) : slot;
        const int c = co0 + p * VE;
        gpy[i] = (c < a.cout) ? P / TW : -100000;
        gpx[i] = P % TW;
        grel[i] = SP ? ((2 * (P / TW) * 2 * a.w + 2 * (P % TW)) * a.cout + c) * ESZ : (((P / TW) * a.w + (P % TW)) * a.cout + c) * ESZ;
    }
    auto issue = [&](int tile, int stage) __attribute__((always_inline)) -> int {     // returns the VMEM instructions this wave issued
        int nvm = 0;
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int n = bx / a.tiles_y;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        const bool dok = (unsigned)(n % a.depth + a.dshift) < (unsigned)a.depth;
        const int xbase = (((n + a.dshift) * hs + (ty0 >> shf)) * wsrc + (tx0 >> shf)) * csrc * ESZ;
        const int gbase = SP ? ((n * 2 * a.h + 2 * ty0 + pa) * 2 * a.w + 2 * tx0 + pb) * a.cout * ESZ : ((n * a.h + ty0) * a.w + tx0) * a.cout * ESZ;
        const int gstep = a.cout * ESZ;                                  // PB: the pb = 1 phase image is one full-resolution pixel to the right
#pragma unroll
        for (int i = 0; i < QX; ++i) {
            const int q = wv + 4 * i;
            if (q < NQX) {
                const int gy = ty0 + xhy[i], gx = tx0 + xhx[i];
                bool ok = dok && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
                if (zsx) ok = ok && ((gy & gx) & 1);
                const unsigned off = (ok && !(a.dbg & 4)) ? (unsigned)(xbase + xrel[i]) : OOB;
                if (a.nt_x) dma16w_nt(rsx, off, lds_base + stage * ST_BYTES + q * 1024);
                else dma16w(rsx, off, lds_base + stage * ST_BYTES + q * 1024);
                ++nvm;
            }
        }
#pragma unroll
        for (int gph = 0; gph < NG; ++gph)
#pragma unroll
        for (int i = 0; i < QG; ++i) {
            const int q = wv + 4 * i;
            if (q < NQG) {
                const int gy = ty0 + gpy[i], gx = tx0 + gpx[i];
                const bool ok = (unsigned)gy < (unsigned)a.h && gx < a.w;
                const unsigned off = (ok && !(a.dbg & 4)) ? (unsigned)(gbase + gph * gstep + grel[i]) : OOB;
                dma16w(rsg, off, lds_base + stage * ST_BYTES + X_BYTES + gph * G_BYTES + q * 1024);
                ++nvm;
            }
        }
        return nvm;
    };

    if (loader) {
        __builtin_amdgcn_s_setprio(3);                       // the loaders' few instructions go ahead of the compute waves' streams
        if (split < a.ntiles) issue(split, 0);
        // VMEM instructions of the tile BEHIND the one the next barrier hands over, as its request counted them (the pieces are dealt
        // round-robin to the four loader waves): they may stay in flight, loads return in order
        int young = 0;
        if constexpr (NST == 3) { if (split + nsplit < a.ntiles) young = issue(split + nsplit, 1); }
        int it = 0;
        for (int tile = split; tile < a.ntiles; tile += nsplit, ++it) {
            // my pieces of this tile have landed; after the barrier everybody's have, and the compute waves are done
            // with the previous tile, whose stage the next one may overwrite
            if constexpr (NST == 3) {
                static_assert(NG == 1 || NST == 2, "the counted wait below knows one dY tile per stage");
                // any count the cases below do not name (none in flight included) waits for everything
                if (young == QX + QG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QX + QG) : "memory");
                else if (young == QX + QG - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QX + QG - 1) : "memory");
                else if (QX + QG >= 3 && young == QX + QG - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QX + QG >= 2 ? QX + QG - 2 : 0) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                young = (tile + 2 * nsplit < a.ntiles && !(a.dbg & 1)) ? issue(tile + 2 * nsplit, (it + 2) % 3) : 0;
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (tile + nsplit < a.ntiles && !(a.dbg & 1)) issue(tile + nsplit, (it + 1) & 1);
            }
        }
        asm volatile("s_barrier" ::: "memory");                       // matches the compute waves' barrier before the fold
        if constexpr (PSPLIT > 1) __syncthreads();                     // the compute waves' blocks are in LDS
        return;
    }

    // (T0, T1): this wave's taps.  acc[lt] belongs to tap T0 + lt (TS = 1: all of them, lt = gph * TAPS + t)
    constexpr int MYT = TS == 2 ? NG * HT : NG * TAPS;          // TS = 2: acc[gph * HT + (tap - T0)]
    f32x16 acc[MYT];
#pragma unroll
    for (int t = 0; t < MYT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto tile_loop = [&](auto T0c, auto T1c) __attribute__((always_inline)) {
    constexpr int T0 = decltype(T0c)::value, T1 = decltype(T1c)::value;

    int it = 0;
    for (int tile = split; tile < a.ntiles; tile += nsplit, ++it) {
        asm volatile("s_barrier" ::: "memory");                       // the tile is in LDS (the loaders waited for their DMAs)
        if (a.dbg & 2) continue;
        const unsigned char* lx = smem + (NST == 3 ? it % 3 : (it & 1)) * ST_BYTES;
        const unsigned char* lg = lx + X_BYTES;
        if constexpr (ESZ == 2) {
            // Every transposed read of the tile = one of 6 (X) / 2 (G) per-lane base addresses + a compile-time offset:
            // the 64-byte-half swizzle of a 128-byte row depends only on bit 1 of the halo x coordinate, which is the
            // same for every k-step (k-steps start on multiples of 16 pixels) and every tap ROW; only the tap COLUMN
            // (tx = 0..2) changes it.  No address arithmetic is left between the MFMAs.
            constexpr int STEPS = 16 / PSPLIT;
            const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, grp = lane >> 4;
            const int kk = 8 * (grp >> 1) + q4;
            const int cb = (16 * (grp & 1) + 4 * p4) * 2;
            const int wave_px = part * STEPS * 16;                       // first pixel of this wave's share (multiple of 64)
            const unsigned char* xp[3][2];
            const unsigned char* gp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pxl = kk + 4 * u;                              // x within the k-step (0..15)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int hx = pxl + tx;
                    xp[tx][u] = lx + ((wave_px / TW) * HWD + hx) * RBX + ((RBX == 128) ? ((ci_t ^ ((hx >> 1) & 1)) << 6) : 0) + cb;
                }
                gp[u] = lg + (wave_px + pxl) * RBG + ((RBG == 128) ? ((co_t ^ ((pxl >> 1) & 1)) << 6) : 0) + cb;
            }
            // sub-pixel form: tap (tr, tc) of phase (pa, pb) reads halo row + pa + tr, halo column + pb + tc
            const unsigned char* xq[SP ? NG : 1][SP ? 2 : 1][2];
            if constexpr (SP) {
#pragma unroll
                for (int gph = 0; gph < NG; ++gph)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int hx = kk + 4 * u + tc + (PB ? gph : pb);
                        xq[gph][tc][u] = lx + ((wave_px / TW + pa) * HWD + hx) * RBX + ((RBX == 128) ? ((ci_t ^ ((hx >> 1) & 1)) << 6) : 0) + cb;
                    }
            }
#pragma unroll
            for (int gph = 0; gph < NG; ++gph)
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                constexpr int dummy = 0; (void)dummy;
                const int srow = (s * 16) / TW, scol = (s * 16) % TW;     // compile-time after unrolling
                const s16x4 g0 = tr_read(gp[0] + gph * G_BYTES + s * 16 * RBG);
                const s16x4 g1 = tr_read(gp[1] + gph * G_BYTES + s * 16 * RBG);
                const uint4 fb = __builtin_bit_cast(uint4, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int t = T0; t < T1; ++t) {
                    const int off = ((srow + (SP ? t >> 1 : t / 3)) * HWD + scol) * RBX;
                    const s16x4 x0 = tr_read((SP ? xq[SP ? gph : 0][t & 1][0] : xp[t % 3][0]) + off);
                    const s16x4 x1 = tr_read((SP ? xq[SP ? gph : 0][t & 1][1] : xp[t % 3][1]) + off);
                    const uint4 fa = __builtin_bit_cast(uint4, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
                    const int ai = TS == 2 ? gph * HT + (t - T0) : gph * TAPS + t;      // compile-time after unrolling
                    acc[ai] = mfma16<T>(fa, fb, acc[ai]);
                }
            }
        } else {
            // f32: 32 channels per operand (128-byte rows), lane = channel, one pixel per half-wave
            constexpr int STEPS = 128 / PSPLIT;
            const int choff = (j & 15) * 4, chalf = j >> 4;
#pragma unroll 2
            for (int s = 0; s < STEPS; ++s) {
                const int P = (part * STEPS + s) * 2 + hf;
                const int py = P / TW, px = P % TW;
                const float g = *reinterpret_cast<const float*>(lg + P * RBG + ((chalf ^ ((px >> 1) & 1)) << 6) + choff);
#pragma unroll
                for (int t = 0; t < (SP ? 0 : 9); ++t) {
                    const int hx = px + t % 3;
                    const int rr = (py + t / 3) * HWD + hx;
                    const float x = *reinterpret_cast<const float*>(lx + rr * RBX + ((chalf ^ ((hx >> 1) & 1)) << 6) + choff);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, g, acc[t], 0, 0, 0);
                }
            }
        }
    }
    };
    if (TS == 2 && th == 1) tile_loop(std::integral_constant<int, HT>{}, std::integral_constant<int, TAPS>{});
    else tile_loop(std::integral_constant<int, 0>{}, std::integral_constant<int, TS == 2 ? HT : TAPS>{});

    // results: one 9 x 32 x 32 fp32 block per (ci_t, co_t) pair; waves that split the pixels fold through LDS
    asm volatile("s_barrier" ::: "memory");                           // every stage has been consumed by every compute wave
    float* out = a.slab + (size_t)bx_ * 9 * a.cin * a.cout;
    // sub-pixel form: which of the phase's two summed taps per axis holds 3x3 tap row kh / column kw
    auto tap_of = [](int phase, int k3) { return phase ? (k3 == 2 ? 1 : 0) : (k3 != 0 ? 1 : 0); };
    if constexpr (PSPLIT == 1) {
        static_assert(PB == 0, "phase pairs are folded through LDS (PSPLIT > 1 blocks)");
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            if (SP ? (2 * tap_of(pa, t9 / 3) + tap_of(pb, t9 % 3) != t) : (t != t9)) continue;           // wave-uniform
            if (TS == 2 && (t < HT) != (th == 0)) continue;                                              // the other tap half's
            const int lt = TS == 2 ? (t < HT ? t : t - HT) : t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + ci_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf, co = co0 + co_t * 32 + j;
                // the slab is read back by the batched fold at the end of the gradient bucket, milliseconds later: non-temporal
                // stores keep its 37.7 MB per layer from displacing the activations (measured: -0.07 ms per step)
                if (ci < a.cin && co < a.cout) {
                    if (a.nt_slab) __builtin_nontemporal_store(acc[lt][r], &out[((size_t)t9 * a.cin + ci) * a.cout + co]);
                    else out[((size_t)t9 * a.cin + ci) * a.cout + co] = acc[lt][r];      // folded by the next launch: keep it cached
                }
            }
          }
    } else {
        // every compute wave parks its block in LDS (the stages are free), one barrier, then 256 threads add the PSPLIT copies in
        // a fixed order and write 16 bytes each (the first form took turns: PSPLIT read-modify-write rounds behind barriers and
        // 4-byte stores -- 10-20 us at the end of the 32-channel launches, with nothing to overlap them)
        constexpr int BLK = NG * TAPS * 32 * 32, N4 = PAIRS * BLK / 4, BLK9 = 9 * 32 * 32;
        float* red = reinterpret_cast<float*>(smem) + (part * PAIRS + pair) * BLK;   // [PSPLIT][PAIRS][9][32][32]
#pragma unroll
        for (int lt = 0; lt < MYT; ++lt) {
            const int tl = TS == 2 ? lt % HT + HT * th : lt;                  // tap of this accumulator (TS = 2: within its phase tile)
            const int t = TS == 2 ? (lt / HT) * TAPS + tl : lt;               // its index in the four-wave form's accumulator order
            if (TS == 2 ? tl < TAPS : t < NG * TAPS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + j] = acc[lt][r];
            }
        }
        __syncthreads();
        const f32x4* r4 = reinterpret_cast<const f32x4*>(smem);
        for (int o4 = tid; o4 < PAIRS * BLK9 / 4; o4 += 256 * TS) {
            const int e = 4 * o4, pr = e / BLK9, rem = e % BLK9;
            const int t = rem >> 10;                                             // 3x3 tap of the slab element
            const int e4 = SP ? (pr * BLK + (2 * tap_of(pa, t / 3) + tap_of(pb, t % 3)) * 1024 + (rem & 1023)) / 4 : o4;
            f32x4 sum = r4[e4];
#pragma unroll
            for (int p = 1; p < PSPLIT; ++p) sum += r4[p * N4 + e4];
            if constexpr (PB) {                                           // + the pb = 1 phase's block that holds this tap
                const int e41 = (pr * BLK + (TAPS + 2 * tap_of(pa, t / 3) + tap_of(1, t % 3)) * 1024 + (rem & 1023)) / 4;
#pragma unroll
                for (int p = 0; p < PSPLIT; ++p) sum += r4[p * N4 + e41];
            }
            const int ci = ci0 + (pr % NCI) * 32 + ((rem >> 5) & 31), co = co0 + (pr / NCI) * 32 + (rem & 31);
            if (ci < a.cin && co + 3 < a.cout) {
                if (a.nt_slab) __builtin_nontemporal_store(sum, reinterpret_cast<f32x4*>(&out[((size_t)t * a.cin + ci) * a.cout + co]));
                else *reinterpret_cast<f32x4*>(&out[((size_t)t * a.cin + ci) * a.cout + co]) = sum;
            } else if (ci < a.cin) {
                for (int q = 0; q < 4; ++q) if (co + q < a.cout) __builtin_nontemporal_store(sum[q], &out[((size_t)t * a.cin + ci) * a.cout + co + q]);
            }
        }
    }
}

template <typename T, int TW, int CIB, int COB, int NST = 2, int TAPS = 9, int PB = 0>
__global__ __launch_bounds__(512, 1) void wgrad3x3_ws(WgArgs2 a) {
    wgrad3x3_ws_body<T, TW, CIB, COB, NST, TAPS, PB>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

template <typename T, int TW, int CIB, int COB, int NST = 2, int TAPS = 9, int PB = 0>
__global__ __launch_bounds__(768, 1) void wgrad3x3_ws12(WgArgs2 a) {
    wgrad3x3_ws_body<T, TW, CIB, COB, NST, TAPS, PB, 2>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// dw[i] = sum_k slab[k][i] in a fixed order (reproducible).  A thread owns four consecutive elements (one 16-byte load per slab);
// G split groups of 32 such threads share a segment, every thread has U = 8 loads in flight before the first add, and R segments
// make up the workgroup.  G is a function of the slab count alone (fold_groups): about eight slabs per thread -- the round-3 geometry
// (32 groups above 32 slabs, 8 groups above 4, 4 below) left a thread of the 64-, 8- and 4-slab layers with one or two loads in
// flight and thousands of tiny workgroups (2.2-2.4 TB/s where the 256-slab layers reached 3.4).  The order of the sum -- per group
// its slabs in steps of G, eight at a time; then the groups in order -- depends on the slab count only, and both folds below use it:
// same bits from either.
static int fold_groups(int nsplit) {
    int g = 1;
    while (g < 32 && g * 8 < nsplit) g <<= 1;
    return g;
}
template <int G, int U, int R>
__global__ __launch_bounds__(32 * G * R) void wgrad_fold_kernel(const float4* __restrict__ slab, int nsplit, long long count4, float4* __restrict__ dw) {
    __shared__ float4 sh[R][G][32];
    const int e = threadIdx.x & 31, g = (threadIdx.x >> 5) % G, r = threadIdx.x / (32 * G);
    const long long i = ((long long)blockIdx.x * R + r) * 32 + e;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < count4) {
        for (int k0 = g; k0 < nsplit; k0 += G * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * G;
                v[u] = k < nsplit ? slab[(size_t)k * count4 + i] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if constexpr (G == 1) {
        if (i < count4) dw[i] = acc;
        return;
    }
    sh[r][g][e] = acc;
    __syncthreads();
    if (g == 0 && i < count4) {
        float4 t = sh[r][0][e];
#pragma unroll
        for (int gg = 1; gg < G; ++gg) { t.x += sh[r][gg][e].x; t.y += sh[r][gg][e].y; t.z += sh[r][gg][e].z; t.w += sh[r][gg][e].w; }
        dw[i] = t;
    }
}

static int launch_wgrad_fold(const float* slab, int nsplit, long long count, float* dw, hipStream_t s) {
    if (count % 4 || ((uintptr_t)slab & 15) || ((uintptr_t)dw & 15)) return RVIP_EINVAL;
    const long long count4 = count / 4;
    const float4* sl = reinterpret_cast<const float4*>(slab);
    float4* out = reinterpret_cast<float4*>(dw);
    const long long segs = cdiv(count4, 32);
    switch (fold_groups(nsplit)) {
        case 1: hipLaunchKernelGGL((wgrad_fold_kernel<1, 8, 8>), dim3((unsigned)cdiv(segs, 8)), dim3(256), 0, s, sl, nsplit, count4, out); break;
        case 2: hipLaunchKernelGGL((wgrad_fold_kernel<2, 8, 4>), dim3((unsigned)cdiv(segs, 4)), dim3(256), 0, s, sl, nsplit, count4, out); break;
        case 4: hipLaunchKernelGGL((wgrad_fold_kernel<4, 8, 2>), dim3((unsigned)cdiv(segs, 2)), dim3(256), 0, s, sl, nsplit, count4, out); break;
        case 8: hipLaunchKernelGGL((wgrad_fold_kernel<8, 8, 1>), dim3((unsigned)segs), dim3(256), 0, s, sl, nsplit, count4, out); break;
        case 16: hipLaunchKernelGGL((wgrad_fold_kernel<16, 8, 1>), dim3((unsigned)segs), dim3(512), 0, s, sl, nsplit, count4, out); break;
        default: hipLaunchKernelGGL((wgrad_fold_kernel<32, 8, 1>), dim3((unsigned)segs), dim3(1024), 0, s, sl, nsplit, count4, out); break;
    }
    return check_launch();
}

// The same fold, row-aligned, with the contraction against the layer's own kernel riding along:
//   dw[i] = sum_k slab[k][i]   and   rows[(tap * nchunk + chunk)][ci] = sum over the chunk's output channels of Wr[tap][ci][o] * dw[tap][ci][o]
// (Wr = the fp32 master rounded to the activation type, i.e. the values the data-gradient kernel multiplies with).  Summed over
// all rows, column ci is  T2[ci] = sum_{t,o} W dW = sum_pixels X[., ci] * dX[., ci]  -- the conv is linear in its input, so the
// inner product of an input channel with its own gradient can be read off the weight gradient (rvip_bn_bwd_coef uses it as the
// sum g*y of the producer's BatchNormalization backward; no pass over g and y).
// A (tap, ci) row of the kernel = cout4 float4s handled by LW = min(32, pow2 >= cout4) lanes of the 32 "element" lanes; 32 / LW
// rows per segment, nchunk = ceil(cout4 / 32) segments per row.  G split groups, U loads in flight and R segments per workgroup as in
// wgrad_fold_kernel.
// SPD (the layer's data gradient runs in sub-pixel form; the slabs come from the four-phase weight-gradient form, slab k = phase
// (k >> 3) & 3 when the grid was XCD-mapped, k & 3 otherwise): the rows are dotted against the PHASE kernels wph[phase][u][v][ci][co]
// of rvip_pack_subpixel_dgrad_weights instead of the rounded taps.  A phase's slab holds its summed-tap block (u, v) at every 3x3 tap
// position the block covers; the first such position (in kh, kw order) represents it.
template <typename T, int G, int U, int R, int SPD = 0>
__global__ __launch_bounds__(32 * G * R) void wgrad_fold_dot_kernel(const float4* __restrict__ slab, int nsplit, long long count4, float4* __restrict__ dw,
                                                                    const float4* __restrict__ w, int cout4, int cin, int nrow, int lw, int nchunk,
                                                                    int nseg, double* __restrict__ rows, const T* __restrict__ wph = nullptr, int xcd_map = 0) {
    // The dot product is carried in DOUBLE from the slabs on: sum W*dW is a small difference of large terms whenever the
    // gradient reaching the producer is mostly common-mode (BatchNormalization removes that part), and float partial sums over
    // the 9 * Cout terms of a channel would lose to that cancellation what the pass over (g, xhat) it replaces does not.  dw
    // itself is the float sum in the plain fold's order (bit-identical to wgrad_fold_kernel).
    __shared__ float4 sh[R][G][32];
    __shared__ double shd[R][G][32][SPD ? 1 : 4];
    const int e = threadIdx.x & 31, g = (threadIdx.x >> 5) % G, r = threadIdx.x / (32 * G);
    const int seg = blockIdx.x * R + r;
    const int rpw = 32 / lw, rl = e / lw, l = e - rl * lw;
    const int chunk = seg % nchunk, row = (seg / nchunk) * rpw + rl, col4 = chunk * 32 + l;
    const bool valid = seg < nseg && row < nrow && col4 < cout4;
    const long long i = (long long)row * cout4 + col4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int NPH = SPD ? 4 : 1;
    double da[NPH][4];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int q = 0; q < 4; ++q) da[ph][q] = 0.0;
    if (valid) {
        for (int k0 = g; k0 < nsplit; k0 += G * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * G;
                v[u] = k < nsplit ? slab[(size_t)k * count4 + i] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                if constexpr (SPD) {
                    const int k = k0 + u * G, phk = xcd_map ? (k >> 3) & 3 : k & 3;      // (slabs past nsplit carry zeros)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        const bool m = phk == ph;
                        da[ph][0] += m ? (double)v[u].x : 0.0; da[ph][1] += m ? (double)v[u].y : 0.0;
                        da[ph][2] += m ? (double)v[u].z : 0.0; da[ph][3] += m ? (double)v[u].w : 0.0;
                    }
                } else {
                    da[0][0] += (double)v[u].x; da[0][1] += (double)v[u].y; da[0][2] += (double)v[u].z; da[0][3] += (double)v[u].w;
                }
            }
        }
    }
    // SPD: this thread's slabs against the phase kernels, before the groups are folded (the product is linear in the slabs)
    double pt = 0.0;
    if constexpr (SPD) {
        if (valid) {
            const int t9 = row / cin, ci = row - t9 * cin, kh = t9 / 3, kw = t9 - 3 * kh;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int pa = ph >> 1, pb = ph & 1;
                // forward convention: phase 0 covers tap 0 | taps 1, 2;  phase 1 covers taps 0, 1 | tap 2  (per axis)
                const int uf = pa ? (kh == 2 ? 1 : 0) : (kh != 0 ? 1 : 0), vf = pb ? (kw == 2 ? 1 : 0) : (kw != 0 ? 1 : 0);
                const bool rep = (pa ? kh != 1 : kh != 2) && (pb ? kw != 1 : kw != 2);       // first tap position of its block, per axis
                if (rep) {
                    // the data-gradient phase kernels index the window mirrored: (u, v)_d = (1 - u, 1 - v)_f
                    const T* wp = wph + (((size_t)(ph * 4 + 2 * (1 - uf) + (1 - vf)) * cin + ci) * (size_t)(cout4 * 4) + (size_t)col4 * 4);
                    const uint2 w2 = *reinterpret_cast<const uint2*>(wp);
                    pt += (double)Vec<T>::dec((uint16_t)(w2.x & 0xffffu)) * da[ph][0] + (double)Vec<T>::dec((uint16_t)(w2.x >> 16)) * da[ph][1]
                        + (double)Vec<T>::dec((uint16_t)(w2.y & 0xffffu)) * da[ph][2] + (double)Vec<T>::dec((uint16_t)(w2.y >> 16)) * da[ph][3];
                }
            }
        }
    }
    double p = 0.0;
    if constexpr (G == 1) {
        if (valid) {
            dw[i] = acc;
            if constexpr (SPD) p = pt;
            else {
                const float4 wv = w[i];
                p = (double)Vec<T>::round(wv.x) * da[0][0] + (double)Vec<T>::round(wv.y) * da[0][1] + (double)Vec<T>::round(wv.z) * da[0][2] + (double)Vec<T>::round(wv.w) * da[0][3];
            }
        }
    } else {
        sh[r][g][e] = acc;
        if constexpr (SPD) shd[r][g][e][0] = pt;
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) shd[r][g][e][q] = da[0][q];
        }
        __syncthreads();
        if (g != 0) return;                                   // group 0 of every segment finishes (its 32 element lanes = one half wave)
        if (valid) {
            float4 t = sh[r][0][e];
            double d[SPD ? 1 : 4];
#pragma unroll
            for (int q = 0; q < (SPD ? 1 : 4); ++q) d[q] = shd[r][0][e][q];
#pragma unroll
            for (int gg = 1; gg < G; ++gg) {
                t.x += sh[r][gg][e].x; t.y += sh[r][gg][e].y; t.z += sh[r][gg][e].z; t.w += sh[r][gg][e].w;
#pragma unroll
                for (int q = 0; q < (SPD ? 1 : 4); ++q) d[q] += shd[r][gg][e][q];
            }
            dw[i] = t;
            if constexpr (SPD) p = d[0];
            else {
                const float4 wv = w[i];
                p = (double)Vec<T>::round(wv.x) * d[0] + (double)Vec<T>::round(wv.y) * d[1] + (double)Vec<T>::round(wv.z) * d[2] + (double)Vec<T>::round(wv.w) * d[3];
            }
        }
    }
    // lw is a power of two <= 32 and a row's lanes are lw consecutive lanes of one 32-lane half wave: the butterfly stays inside them
    for (int o = 1; o < lw; o <<= 1) p += __shfl_xor(p, o);
    if (l == 0 && seg < nseg && row < nrow) {
        const int tap = row / cin, ci = row - tap * cin;
        rows[((size_t)tap * nchunk + chunk) * cin + ci] = p;
    }
}

struct DotGeom { int cout4, lw, nchunk, nrow; };
static DotGeom dot_geometry(int taps, int cin, int cout) {
    DotGeom g;
    g.cout4 = cout / 4;
    int lw = 1;
    while (lw < g.cout4 && lw < 32) lw <<= 1;
    g.lw = lw;
    g.nchunk = (int)cdiv(g.cout4, 32);
    g.nrow = taps * cin;
    return g;
}

template <typename T>
static int launch_wgrad_fold_dot(const float* slab, int nsplit, int cin, int cout, float* dw, const float* w, double* rows, hipStream_t s,
                                 const void* w_phase = nullptr, int xcd_map = 0) {
    const long long count = 9LL * cin * cout;
    if (cout % 4 || ((uintptr_t)slab & 15) || ((uintptr_t)dw & 15) || ((uintptr_t)w & 15)) return RVIP_EINVAL;
    const DotGeom g = dot_geometry(9, cin, cout);
    const int nseg = (int)(cdiv(g.nrow, 32 / g.lw) * g.nchunk);
    const float4* sl = reinterpret_cast<const float4*>(slab);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float4* out = reinterpret_cast<float4*>(dw);
#define RVIP_FOLD_DOT(G_, R_) do { \
        if constexpr (sizeof(T) == 2) { \
            if (w_phase) { hipLaunchKernelGGL((wgrad_fold_dot_kernel<T, G_, 8, R_, 1>), dim3((unsigned)cdiv(nseg, R_)), dim3(32 * G_ * R_), 0, s, sl, nsplit, count / 4, out, w4, g.cout4, cin, g.nrow, g.lw, g.nchunk, nseg, rows, (const T*)w_phase, xcd_map); break; } \
        } \
        hipLaunchKernelGGL((wgrad_fold_dot_kernel<T, G_, 8, R_>), dim3((unsigned)cdiv(nseg, R_)), dim3(32 * G_ * R_), 0, s, sl, nsplit, count / 4, out, w4, g.cout4, cin, g.nrow, g.lw, g.nchunk, nseg, rows, (const T*)nullptr, 0); \
    } while (0)
    switch (fold_groups(nsplit)) {
        case 1: RVIP_FOLD_DOT(1, 8); break;
        case 2: RVIP_FOLD_DOT(2, 4); break;
        case 4: RVIP_FOLD_DOT(4, 2); break;
        case 8: RVIP_FOLD_DOT(8, 1); break;
        case 16: RVIP_FOLD_DOT(16, 1); break;
        default: RVIP_FOLD_DOT(32, 1); break;
    }
#undef RVIP_FOLD_DOT
    return check_launch();
}

static void wgrad_geometry(int n, int h, int w, int cin, int cout, int& tw, int& tiles_x, int& tiles_y, int& ntiles, int& nsplit) {
    tw = w > 16 ? 32 : 16;
    const int th = 256 / tw;
    tiles_x = (int)cdiv(w, tw);
    tiles_y = (int)cdiv(h, th);
    ntiles = n * tiles_x * tiles_y;
    const long long pairs = cdiv(cin, 32) * cdiv(cout, 32);
    long long s = 512 / pairs;                 // ~2 workgroups per CU in total
    if (s < 1) s = 1;
    if (s > ntiles) s = ntiles;
    nsplit = (int)s;
}

struct Wg2Geom { int tw, cib, cob, tiles_x, tiles_y, ntiles, nsplit; bool ok; bool pair = false; };     // pair: sub-pixel form with both column phases per workgroup

static int wg_cus(const rvip_wgrad3x3_desc* d) { return (d->cu_limit > 0 && d->cu_limit < 256) ? d->cu_limit : 256; }

static Wg2Geom wgrad2_geometry(int n, int h, int w, int c0, int c1, int cout, int dtype, int cus = 256) {
    Wg2Geom g;
    const int cin = c0 + c1;
    const int esz = RVIP_ESZ(dtype);
    const int wide = 128 / esz, narrow = 64 / esz;             // channels in a 128-byte / 64-byte row
    g.tw = w > 16 ? 32 : 16;
    const int th = 256 / g.tw;
    g.tiles_x = (int)cdiv(w, g.tw); g.tiles_y = (int)cdiv(h, th);
    g.ntiles = n * g.tiles_x * g.tiles_y;
    if (dtype != RVIP_F32) {
        g.cib = (cin >= wide && (c1 == 0 || c0 % wide == 0)) ? wide : narrow;
        g.cob = cout >= wide ? wide : narrow;
    } else {
        g.cib = wide; g.cob = wide;                              // f32: 32 channels = 128-byte rows
    }
    g.ok = (c1 == 0 || c0 % g.cib == 0);
    const long long blocks = cdiv(cin, g.cib) * cdiv(cout, g.cob);
    long long s = cus / blocks;                                  // one workgroup per CU (LDS-limited)
    if (s < 1) s = 1;
    if (s > g.ntiles) s = g.ntiles;
    g.nsplit = (int)s;
    return g;
}

template <typename T, int TW, int CIB, int COB, bool WS, int TAPS = 9, int PB = 0>
static int launch_wgrad2x(const WgArgs2& a, hipStream_t s) {
    constexpr int TH = 256 / TW;
    constexpr int NHROWS = ((TW + 2) * (TH + 2) + 15) / 16 * 16;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int ST = NHROWS * CIB * ESZ + (PB ? 2 : 1) * 256 * COB * ESZ;
    constexpr int NST = (WS && 3 * ST <= 160 * 1024) ? 3 : 2;
    // the wave-specialised kernel folds the pixel-split copies of a block through LDS: 4 x [9][32][32] floats
    constexpr int FOLD = (WS && (ESZ == 4 || (CIB / 32) * (COB / 32) < 4)) ? 4 * (PB ? 2 : 1) * TAPS * 32 * 32 * 4 : 0;
    constexpr int lds = NST * ST > FOLD ? NST * ST : FOLD;
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<bool> attr_done{false};      // idempotent attribute call; atomic so concurrent host threads do not race on the flag
    if (!attr_done) {
        hipError_t e;
        if constexpr (WS) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_ws<T, TW, CIB, COB, NST, TAPS, PB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        else e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_dma<T, TW, CIB, COB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
        attr_done = true;
    }
    dim3 grid((unsigned)a.nsplit, (unsigned)cdiv(a.cin, CIB), (unsigned)cdiv(a.cout, COB));
    if constexpr (WS && sizeof(T) == 2) {
        // (A/B: the eight-compute-wave form -- the taps split between two waves -- as a launch of its own)
        static const bool ws12 = [] { const char* e = getenv("RVIP_WGRAD_WS12"); return e && e[0] == '1'; }();
        if (ws12) {
            static std::atomic<bool> attr12{false};
            if (!attr12) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_ws12<T, TW, CIB, COB, NST, TAPS, PB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
                attr12 = true;
            }
            hipLaunchKernelGGL((wgrad3x3_ws12<T, TW, CIB, COB, NST, TAPS, PB>), grid, dim3(768), lds, s, a);
            return check_launch();
        }
    }
    if constexpr (WS) hipLaunchKernelGGL((wgrad3x3_ws<T, TW, CIB, COB, NST, TAPS, PB>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((wgrad3x3_dma<T, TW, CIB, COB>), grid, dim3(256), lds, s, a);
    return check_launch();
}

// the wave-specialised kernel for the 16-bit types, the four-wave LDS-DMA kernel for f32
template <typename T, int TW, int CIB, int COB>
static int launch_wgrad2(const WgArgs2& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if constexpr (CIB == 64 && COB == 32) { if (a.sp == 2) return launch_wgrad2x<T, TW, CIB, COB, true, 4, 1>(a, s); }
        if (a.sp) return launch_wgrad2x<T, TW, CIB, COB, true, 4>(a, s);
    }
    return launch_wgrad2x<T, TW, CIB, COB, sizeof(T) == 2>(a, s);
}

// UpSampling2D -> conv, 16-bit types, 2-D: geometry of the sub-pixel form (tiles of the LOW-resolution grid, 4 phases x pixel splits)
static bool wgrad_subpixel_geometry(const rvip_wgrad3x3_desc* d, Wg2Geom& g) {
    const char* e = getenv("RVIP_SUBPIX_WGRAD");                  // read per call (host side of a launch; tests flip it): 0 never, 2 wherever eligible
    const bool on = !(e && e[0] == '0'), force = e && e[0] == '2';
    const int kd = d->kd > 0 ? d->kd : 1;
    if (!on || d->up0 != 1 || d->c1 != 0 || d->dtype == RVIP_F32 || kd != 1 || ((d->h | d->w) & 1)) return false;
    g = wgrad2_geometry(d->n, d->h / 2, d->w / 2, d->c0, 0, d->cout, d->dtype, wg_cus(d));
    if (!g.ok) return false;
    // The form saves matrix work (16 / 36), not staging: every phase stages the X tile again.  It pays where the nine-tap kernel is
    // bound by its MFMA / LDS-read side -- the 64 x 64 blocks (512 -> 256 .. 128 -> 64: 64 -> 45 us) -- and not on the 32-wide
    // blocks of the full-resolution layer, whose tiles carry half the matrix work per staged byte (64 -> 32 at 256^2: 74 -> 77 us).
    const bool pair = g.cib == 64 && g.cob == 32;                 // both column phases per workgroup (wgrad3x3_ws<..., PB = 1>): fits LDS there
    if (!force && !(g.cib == 64 && g.cob == 64) && !pair) return false;
    const long long blocks = cdiv(d->c0, g.cib) * cdiv(d->cout, g.cob);
    const int nph = pair ? 2 : 4;                                 // phases that are separate workgroups
    long long sp = wg_cus(d) / nph / blocks;                      // phases x sp pixel splits x blocks ~ one workgroup per CU
    if (sp < 1) sp = 1;
    if (sp > g.ntiles) sp = g.ntiles;
    g.nsplit = (int)(nph * sp);
    g.pair = pair;
    return true;
}


template <typename T, int TW>
static int launch_wgrad(const WgArgs& a, hipStream_t s) {
    constexpr int TH = 256 / TW;
    constexpr int rowb = 32 * (int)sizeof(T);
    constexpr int lds_tiles = (TW + 2) * (TH + 2) * rowb + 256 * rowb;
    constexpr int lds = lds_tiles > 9 * 32 * 32 * 4 ? lds_tiles : 9 * 32 * 32 * 4;
    static std::atomic<bool> attr_done{false};      // idempotent attribute call; atomic so concurrent host threads do not race on the flag
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel<T, TW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
        attr_done = true;
    }
    dim3 grid((unsigned)a.nsplit, (unsigned)cdiv(a.cin, 32), (unsigned)cdiv(a.cout, 32));
    hipLaunchKernelGGL((wgrad3x3_kernel<T, TW>), grid, dim3(256), lds, s, a);
    return check_launch();
}

}  // namespace rvip

extern "C" int rvip_conv3x3_wgrad_dot_rows(const rvip_wgrad3x3_desc* d);
extern "C" int rvip_conv3x3_wgrad_form(const rvip_wgrad3x3_desc* d);

namespace rvip {
// The arguments rvip_conv3x3_wgrad hands to wgrad3x3_ws for a 2-D layer of a 16-bit type (the case the pair kernel of rvip_pair.hip
// serves), without launching; RVIP_EUNSUPPORTED for everything else.  Mirrors the checks of rvip_conv3x3_wgrad.
struct WgradPlan { WgArgs2 b; Wg2Geom g; };
static int wgrad_plan(const rvip_wgrad3x3_desc* d, WgradPlan& p) {
    if (!d || !d->x0 || !d->dy || !d->dw || !d->workspace) return RVIP_EINVAL;
    if (d->dtype != RVIP_BF16 && d->dtype != RVIP_F16) return RVIP_EUNSUPPORTED;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->c0 <= 0) return RVIP_EINVAL;
    if (d->c0 % 8 || d->c1 % 8 || d->cout % 8) return RVIP_EINVAL;
    if ((d->c1 > 0) != (d->x1 != nullptr)) return RVIP_EINVAL;
    if (d->up0 < 0 || d->up0 > 2 || (d->up0 == 2 && d->c1 > 0)) return RVIP_EINVAL;
    if (d->up0 && ((d->h | d->w) & 1)) return RVIP_EINVAL;
    const int depth = d->depth > 0 ? d->depth : 1, kd = d->kd > 0 ? d->kd : 1;
    if (kd != 1 || depth != 1) return RVIP_EUNSUPPORTED;
    if (d->w_phase && !d->dot_rows) return RVIP_EINVAL;
    if (d->w_phase && rvip_conv3x3_wgrad_form(d) != 1) return RVIP_EUNSUPPORTED;
    const int cin = d->c0 + d->c1;
    if (d->dot_rows) {
        if (!d->w_master || d->defer_fold) return RVIP_EINVAL;
        if (d->dot_rows_bytes < (size_t)rvip_conv3x3_wgrad_dot_rows(d) * cin * sizeof(double) || ((uintptr_t)d->dot_rows & 7)) return RVIP_EWORKSPACE;
    }
    const int up = d->up0 ? 1 : 0;
    const long long x0b = (long long)d->n * (d->h >> up) * (d->w >> up) * d->c0 * 2, x1b = (long long)d->n * d->h * d->w * d->c1 * 2;
    const long long dyb = (long long)d->n * d->h * d->w * d->cout * 2;
    Wg2Geom& g2 = p.g;
    const bool sp = wgrad_subpixel_geometry(d, g2);
    if (!sp) g2 = wgrad2_geometry(d->n, d->h, d->w, d->c0, d->c1, d->cout, d->dtype, wg_cus(d));
    if (!(g2.ok && x0b < (1LL << 31) && x1b < (1LL << 31) && dyb < (1LL << 31))) return RVIP_EUNSUPPORTED;
    WgArgs2& b = p.b;
    b.x0 = (const unsigned char*)d->x0; b.x1 = (const unsigned char*)d->x1; b.dy = (const unsigned char*)d->dy; b.slab = (float*)d->workspace;
    b.x0_bytes = (unsigned)x0b; b.x1_bytes = (unsigned)x1b; b.dy_bytes = (unsigned)dyb;
    b.c0 = d->c0; b.c1 = d->c1; b.up0 = up; b.zs = d->up0 == 2; b.n = d->n; b.h = d->h; b.w = d->w; b.cin = cin; b.cout = d->cout;
    b.sp = sp ? (g2.pair ? 2 : 1) : 0;
    if (sp) { b.up0 = 0; b.h = d->h / 2; b.w = d->w / 2; }
    b.tiles_x = g2.tiles_x; b.tiles_y = g2.tiles_y; b.ntiles = g2.ntiles; b.nsplit = g2.nsplit;
    { static const int dbg = [] { const char* e = getenv("RVIP_DBG"); return e ? atoi(e) : 0; }(); b.dbg = dbg; }
    { static const bool ntx = [] { const char* e = getenv("RVIP_NT_WGRAD"); return !(e && e[0] == '0'); }(); b.nt_x = ntx ? 1 : 0; }
    b.nt_slab = d->defer_fold ? 1 : 0;
    b.depth = 1; b.dshift = 0;
    if (d->workspace_bytes < (size_t)b.nsplit * 9 * cin * d->cout * sizeof(float)) return RVIP_EWORKSPACE;
    return RVIP_OK;
}
// what follows the kernel: the slab fold (with the dot rows), or nothing when the fold is deferred (defined once, in rvip_wgrad.hip's
// own translation unit: it launches the fold kernels)
int wgrad_finish(const rvip_wgrad3x3_desc* d, int nsplit, int sp, hipStream_t s);
}  // namespace rvip

#ifndef RVIP_KERNELS_ONLY        /* rvip_pair.hip includes this file for its kernels and launch geometry only */
namespace rvip {
int wgrad_finish(const rvip_wgrad3x3_desc* d, int nsplit, int sp, hipStream_t s) {
    if (d->defer_fold) return RVIP_OK;
    const int cin = d->c0 + d->c1;
    const long long count2 = 9LL * cin * d->cout;
    if (d->dot_rows) {
        return by_dtype(d->dtype, [&](auto t) {
            return launch_wgrad_fold_dot<decltype(t)>((const float*)d->workspace, nsplit, cin, d->cout, d->dw, d->w_master, d->dot_rows, s,
                                                      d->w_phase, (sp == 1 && nsplit % 32 == 0) ? 1 : 0);
        });
    }
    return launch_wgrad_fold((const float*)d->workspace, nsplit, count2, d->dw, s);
}
}  // namespace rvip
using namespace rvip;

extern "C" size_t rvip_conv3x3_wgrad_workspace(int n, int h, int w, int cin, int cout) {
    int tw, tx, ty, nt, ns;
    wgrad_geometry(n, h, w, cin, cout, tw, tx, ty, nt, ns);
    int ns2 = 4 * nt < 256 ? 4 * nt : 256;                      // upper bound of the LDS-DMA kernels' split count (sub-pixel form: 4 phases x <= nt pixel splits)
    if (ns2 > ns) ns = ns2;
    return (size_t)ns * 9 * cin * cout * sizeof(float);
}

extern "C" int rvip_conv3x3_wgrad(const rvip_wgrad3x3_desc* d, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->x0 || !d->dy || !d->dw || !d->workspace) return RVIP_EINVAL;
    if (!RVIP_DT_OK(d->dtype)) return RVIP_EINVAL;
    const int ve = RVIP_VE(d->dtype);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->c0 <= 0) return RVIP_EINVAL;
    if (d->c0 % ve || d->c1 % ve || d->cout % ve) return RVIP_EINVAL;
    if ((d->c1 > 0) != (d->x1 != nullptr)) return RVIP_EINVAL;
    if (d->up0 < 0 || d->up0 > 2 || (d->up0 == 2 && d->c1 > 0)) return RVIP_EINVAL;
    if (d->up0 && ((d->h | d->w) & 1)) return RVIP_EINVAL;
    WgArgs a;
    a.x0 = (const unsigned char*)d->x0; a.x1 = (const unsigned char*)d->x1; a.dy = (const unsigned char*)d->dy;
    a.slab = (float*)d->workspace;
    a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0 ? 1 : 0; a.zs = d->up0 == 2;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->c0 + d->c1; a.cout = d->cout;
    hipStream_t s = (hipStream_t)stream;
    int rc = RVIP_OK;
    const int depth = d->depth > 0 ? d->depth : 1, kd = d->kd > 0 ? d->kd : 1;
    if ((kd != 1 && kd != 3) || d->n % depth) return RVIP_EINVAL;
    if (d->w_phase && !d->dot_rows) return RVIP_EINVAL;
    if (d->w_phase && (d->dtype == RVIP_F32 || rvip_conv3x3_wgrad_form(d) != 1)) return RVIP_EUNSUPPORTED;
    if (d->dot_rows) {
        if (!d->w_master || d->defer_fold) return RVIP_EINVAL;
        if (d->dot_rows_bytes < (size_t)rvip_conv3x3_wgrad_dot_rows(d) * (d->c0 + d->c1) * sizeof(double) || ((uintptr_t)d->dot_rows & 7)) return RVIP_EWORKSPACE;
    }
    const long long esz = RVIP_ESZ(d->dtype);
    const long long x0b = (long long)a.n * (a.h >> a.up0) * (a.w >> a.up0) * a.c0 * esz, x1b = (long long)a.n * a.h * a.w * a.c1 * esz;
    const long long dyb = (long long)a.n * a.h * a.w * a.cout * esz;
    Wg2Geom g2;
    const bool sp = wgrad_subpixel_geometry(d, g2);
    if (!sp) g2 = wgrad2_geometry(a.n, a.h, a.w, a.c0, a.c1, a.cout, d->dtype, wg_cus(d));
    if (g2.ok && x0b < (1LL << 31) && x1b < (1LL << 31) && dyb < (1LL << 31)) {
        WgArgs2 b;
        b.x0 = a.x0; b.x1 = a.x1; b.dy = a.dy; b.slab = a.slab;
        b.x0_bytes = (unsigned)x0b; b.x1_bytes = (unsigned)x1b; b.dy_bytes = (unsigned)dyb;
        b.c0 = a.c0; b.c1 = a.c1; b.up0 = a.up0; b.zs = a.zs; b.n = a.n; b.h = a.h; b.w = a.w; b.cin = a.cin; b.cout = a.cout;
        b.sp = sp ? (g2.pair ? 2 : 1) : 0;                       // 2: both column phases per workgroup
        if (sp) { b.up0 = 0; b.h = a.h / 2; b.w = a.w / 2; }     // X is read as it lies; tiles, borders and splits are those of its grid
        b.tiles_x = g2.tiles_x; b.tiles_y = g2.tiles_y; b.ntiles = g2.ntiles; b.nsplit = g2.nsplit;
        { static const int dbg = [] { const char* e = getenv("RVIP_DBG"); return e ? atoi(e) : 0; }(); b.dbg = dbg; }
        { static const bool ntx = [] { const char* e = getenv("RVIP_NT_WGRAD"); return !(e && e[0] == '0'); }(); b.nt_x = ntx ? 1 : 0; }
        b.nt_slab = d->defer_fold ? 1 : 0;        // (measured equal either way for the fold that follows at once: 6 452 vs 6 451 slices/s)
        if (d->workspace_bytes < (size_t)b.nsplit * 9 * a.cin * a.cout * sizeof(float)) return RVIP_EWORKSPACE;
        // Conv3D: one pass per depth tap (X shifted by kdi - 1 images inside the volume) into dw[kdi][9][Cin][Cout]
        const long long count2 = 9LL * a.cin * a.cout;
        for (int kdi = 0; kdi < kd; ++kdi) {
            b.depth = depth; b.dshift = kdi - (kd >> 1);
            rc = by_dtype(d->dtype, [&](auto t) {
                using T = decltype(t);
                if constexpr (sizeof(T) == 4) {
                    return g2.tw == 32 ? launch_wgrad2<float, 32, 32, 32>(b, s) : launch_wgrad2<float, 16, 32, 32>(b, s);
                } else if (g2.tw == 32) {
                    if (g2.cib == 64 && g2.cob == 64) return launch_wgrad2<T, 32, 64, 64>(b, s);
                    if (g2.cib == 64) return launch_wgrad2<T, 32, 64, 32>(b, s);
                    if (g2.cob == 64) return launch_wgrad2<T, 32, 32, 64>(b, s);
                    return launch_wgrad2<T, 32, 32, 32>(b, s);
                } else {
                    if (g2.cib == 64 && g2.cob == 64) return launch_wgrad2<T, 16, 64, 64>(b, s);
                    if (g2.cib == 64) return launch_wgrad2<T, 16, 64, 32>(b, s);
                    if (g2.cob == 64) return launch_wgrad2<T, 16, 32, 64>(b, s);
                    return launch_wgrad2<T, 16, 32, 32>(b, s);
                }
            });
            if (rc) return rc;
            if (d->defer_fold && kd == 1) return RVIP_OK;      // slabs stay in the caller's workspace for rvip_fold_rows_batch
            if (d->dot_rows) {
                const int rpp = 9 * dot_geometry(9, a.cin, a.cout).nchunk;            // rows per depth-tap pass
                rc = by_dtype(d->dtype, [&](auto t) {
                    return launch_wgrad_fold_dot<decltype(t)>(a.slab, b.nsplit, a.cin, a.cout, d->dw + (size_t)kdi * count2,
                                                              d->w_master + (size_t)kdi * count2, d->dot_rows + (size_t)kdi * rpp * a.cin, s,
                                                              d->w_phase, (b.sp == 1 && b.nsplit % 32 == 0) ? 1 : 0);
                });
            } else rc = launch_wgrad_fold(a.slab, b.nsplit, count2, d->dw + (size_t)kdi * count2, s);
            if (rc) return rc;
        }
        return RVIP_OK;
    }
    if (kd > 1) return RVIP_EUNSUPPORTED;         // the register-staged fallback is 2-D only
    int tw;
    wgrad_geometry(a.n, a.h, a.w, a.cin, a.cout, tw, a.tiles_x, a.tiles_y, a.ntiles, a.nsplit);
    const size_t need = (size_t)a.nsplit * 9 * a.cin * a.cout * sizeof(float);
    if (d->workspace_bytes < need) return RVIP_EWORKSPACE;
    rc = by_dtype(d->dtype, [&](auto t) { return tw == 32 ? launch_wgrad<decltype(t), 32>(a, s) : launch_wgrad<decltype(t), 16>(a, s); });
    if (rc || d->defer_fold) return rc;
    const long long count = 9LL * a.cin * a.cout;
    if (d->dot_rows) return by_dtype(d->dtype, [&](auto t) { return launch_wgrad_fold_dot<decltype(t)>(a.slab, a.nsplit, a.cin, a.cout, d->dw, d->w_master, d->dot_rows, s); });
    return launch_wgrad_fold(a.slab, a.nsplit, count, d->dw, s);
}

// rows of rvip_wgrad3x3_desc.dot_rows for this shape: [kd * 9 * ceil(cout / 128)][cin]
extern "C" int rvip_conv3x3_wgrad_dot_rows(const rvip_wgrad3x3_desc* d) {
    if (!d || d->cout <= 0 || d->cout % 4 || d->c0 <= 0) return 0;
    const int kd = d->kd > 0 ? d->kd : 1;
    return kd * 9 * dot_geometry(9, d->c0 + d->c1, d->cout).nchunk;
}

// which kernel form rvip_conv3x3_wgrad launches for this shape: 0 nine-tap LDS-DMA kernel, 1 sub-pixel form (four phase workgroups),
// 2 sub-pixel form with both column phases per workgroup (PB = 1), 3 register-staged fallback, -1 invalid descriptor
extern "C" int rvip_conv3x3_wgrad_form(const rvip_wgrad3x3_desc* d) {
    if (!d || !RVIP_DT_OK(d->dtype) || d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c0 <= 0 || d->cout <= 0) return -1;
    const int up = d->up0 ? 1 : 0;
    const long long esz = RVIP_ESZ(d->dtype);
    const long long x0b = (long long)d->n * (d->h >> up) * (d->w >> up) * d->c0 * esz, x1b = (long long)d->n * d->h * d->w * d->c1 * esz;
    const long long dyb = (long long)d->n * d->h * d->w * d->cout * esz;
    Wg2Geom g2;
    const bool sp = wgrad_subpixel_geometry(d, g2);
    if (!sp) g2 = wgrad2_geometry(d->n, d->h, d->w, d->c0, d->c1, d->cout, d->dtype, wg_cus(d));
    if (g2.ok && x0b < (1LL << 31) && x1b < (1LL << 31) && dyb < (1LL << 31)) return sp ? (g2.pair ? 2 : 1) : 0;
    return 3;
}

// number of split-K slabs rvip_conv3x3_wgrad writes for this shape (rows of the deferred fold; 0 = invalid descriptor)
extern "C" int rvip_conv3x3_wgrad_splits(const rvip_wgrad3x3_desc* d) {
    if (!d || !RVIP_DT_OK(d->dtype) || d->n <= 0 || d->h <= 0 || d->w <= 0) return 0;
    const int up = d->up0 ? 1 : 0;
    const long long esz = RVIP_ESZ(d->dtype);
    const long long x0b = (long long)d->n * (d->h >> up) * (d->w >> up) * d->c0 * esz, x1b = (long long)d->n * d->h * d->w * d->c1 * esz;
    const long long dyb = (long long)d->n * d->h * d->w * d->cout * esz;
    Wg2Geom g2;
    if (!wgrad_subpixel_geometry(d, g2)) g2 = wgrad2_geometry(d->n, d->h, d->w, d->c0, d->c1, d->cout, d->dtype, wg_cus(d));
    if (g2.ok && x0b < (1LL << 31) && x1b < (1LL << 31) && dyb < (1LL << 31)) return g2.nsplit;
    int tw, tx, ty, nt, ns;
    wgrad_geometry(d->n, d->h, d->w, d->c0 + d->c1, d->cout, tw, tx, ty, nt, ns);
    return ns;
}
#endif  /* RVIP_KERNELS_ONLY */
