// HBM-bound kernels of the training step: BatchNormalization (train/infer/backward), Dropout, MaxPooling,
// UpSampling, the 1x1 sigmoid head with MSE / BCE-Dice loss and metrics, Keras-Adam, landmark argmax.
// All NHWC with 16-byte channel vectors (8 bf16 / 4 f32 per lane), fp32 arithmetic.
//
// Per-channel reductions are two-stage and deterministic:
//   stage 1: <= 1024 workgroups, each folds a contiguous range of pixel rows; thread = (row slot, channel
//            vector); the workgroup's [K][C] partial goes to the workspace row blockIdx.x;
//   stage 2: fold_finalize<K, Post>: 32 channels per workgroup, 8 row groups per channel, double
//            accumulation in a fixed order, then the op-specific epilogue (Post).
#include "rvip_common.h"

namespace rvip {

// q -> (x, y, image) for a row-major [image][h][w] index.  64-bit division is emulated on the GPU (~100 instructions per
// quotient): three of them per 2x2 window made the pooled bn_apply VALU-bound (3.1 TB/s); the index fits 32 bits in practice.
__device__ __forceinline__ void split_xy(long long q, int w, int h, int& x, int& y, long long& img) {
    if (q < (1LL << 31)) {
        const unsigned u = (unsigned)q, t = u / (unsigned)w, i = t / (unsigned)h;
        x = (int)(u - t * (unsigned)w); y = (int)(t - i * (unsigned)h); img = i;
    } else {
        x = (int)(q % w); y = (int)((q / w) % h); img = q / ((long long)w * h);
    }
}

// ------------------------------------------------------------------------------------------------
// reduction geometry (host + device agree through these fields)
// ------------------------------------------------------------------------------------------------
struct RedGeom {
    int cg;            // channel vectors per row
    int rpi;           // rows per workgroup iteration = 256 / cg
    int nblk;          // workgroups
    long long chunk;   // rows per workgroup (multiple of rpi)
};

// cap: most workgroups the kernel keeps RESIDENT (256 CUs x waves/SIMD its register count allows): a grid one quarter larger than
// that runs a second, mostly empty round
static inline bool red_geom(long long rows, int c, int ve, RedGeom& g, int cap = 1024) {
    if (c <= 0 || c % ve || rows <= 0) return false;
    g.cg = c / ve;
    if (g.cg > 256) return false;
    g.rpi = 256 / g.cg;
    long long nb = cdiv(rows, (long long)g.rpi * 8);
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    long long ch = cdiv(rows, nb);
    ch = cdiv(ch, g.rpi) * g.rpi;
    g.chunk = ch;
    g.nblk = (int)cdiv(rows, ch);
    return true;
}

template <int K, int VE>
__device__ __forceinline__ void block_fold(const float (&part)[K][VE], bool active, int slot, int c, int rpi,
                                           float* lds, float* out_row) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int e = 0; e < VE; ++e) lds[slot * VE + e] = part[k][e];
        }
        __syncthreads();
        for (int col = tid; col < c; col += 256) {
            float s = 0.f;
            for (int r = 0; r < rpi; ++r) s += lds[r * c + col];
            out_row[k * c + col] = s;
        }
    }
}

// Buffer loads for the folds that request many rows at once: one 32-bit offset register per thread and a scalar offset per row
// instead of a 64-bit address pair per load in flight; offsets past the end (and NULL sources: 0 records) read as zero.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// stage 2: totals[k] for channel ch = sum over workspace rows, then Post::run(ch, totals).
// 512 threads = 32 channels x 16 row groups; every thread keeps 4 loads in flight (the loop is latency-bound).
// GROUPS row groups of 32 channels: 32 (1024 threads) halves the chain of load round trips of the 1024-row folds; a Post whose
// epilogue needs many registers (PostBnStats: double-precision moving-average arithmetic) stays at 16 -- under the 128-VGPR
// cap of a 1024-thread workgroup it spills, and a kernel with scratch pays for it at every (tiny) launch.
template <int K, typename Post, int GROUPS>
__global__ __launch_bounds__(32 * GROUPS) void fold_finalize(const float* __restrict__ ws, int nblk, int width, Post post) {
    __shared__ double sh[GROUPS][K][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ch = blockIdx.x * 32 + c;
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    if (ch < width) {
        int b = g;
        for (; b + 3 * GROUPS < nblk; b += 4 * GROUPS) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float v0 = ws[((size_t)b * K + k) * width + ch];
                const float v1 = ws[((size_t)(b + GROUPS) * K + k) * width + ch];
                const float v2 = ws[((size_t)(b + 2 * GROUPS) * K + k) * width + ch];
                const float v3 = ws[((size_t)(b + 3 * GROUPS) * K + k) * width + ch];
                s[k] += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
            }
        }
        for (; b < nblk; b += GROUPS) {
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] += (double)ws[((size_t)b * K + k) * width + ch];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) sh[g][k][c] = s[k];
    __syncthreads();
    if (g == 0 && ch < width) {
        double t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int gg = 0; gg < GROUPS; ++gg) acc += sh[gg][k][c];
            t[k] = acc;
        }
        post.run(ch, t);
    }
}

template <int K, typename Post, int GROUPS = 32>
static int launch_fold(const float* ws, int nblk, int width, Post post, hipStream_t s) {
    hipLaunchKernelGGL((fold_finalize<K, Post, GROUPS>), dim3((unsigned)cdiv(width, 32)), dim3(32 * GROUPS), 0, s, ws, nblk, width, post);
    return check_launch();
}

#define RVIP_FOLDK_GROUPS 32
#define RVIP_FOLDK_DEPTH 8
// the same fold for outputs that are independent per k (weight gradients with 8-9 rows per channel): blockIdx.y = k,
// so the K chains of load latencies run side by side instead of one after the other.  Post::run_k(ch, k, total).
template <typename Post>
__global__ __launch_bounds__(1024) void fold_finalize_k(const float* __restrict__ ws, int nblk, int width, int K, Post post) {
    __shared__ double sh[RVIP_FOLDK_GROUPS / 2][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5, k = blockIdx.y;
    const int ch = blockIdx.x * 32 + c;
    double s = 0.0;
    if (ch < width) {
        for (int b0 = g; b0 < nblk; b0 += RVIP_FOLDK_GROUPS * RVIP_FOLDK_DEPTH) {
            float v[RVIP_FOLDK_DEPTH];
#pragma unroll
            for (int u = 0; u < RVIP_FOLDK_DEPTH; ++u) {
                const int b = b0 + u * RVIP_FOLDK_GROUPS;
                v[u] = b < nblk ? ws[((size_t)b * K + k) * width + ch] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < RVIP_FOLDK_DEPTH; u += 4) s += ((double)v[u] + (double)v[u + 1]) + ((double)v[u + 2] + (double)v[u + 3]);
        }
    }
    const double o = __shfl_xor(s, 32);
    if (!(g & 1)) sh[g >> 1][c] = s + o;
    __syncthreads();
    if (g == 0 && ch < width) {
        double acc = 0.0;
#pragma unroll
        for (int gg = 0; gg < RVIP_FOLDK_GROUPS / 2; ++gg) acc += sh[gg][c];
        post.run_k(ch, k, acc);
    }
}

template <typename Post>
static int launch_fold_k(const float* ws, int nblk, int width, int K, Post post, hipStream_t s) {
    hipLaunchKernelGGL((fold_finalize_k<Post>), dim3((unsigned)cdiv(width, 32), (unsigned)K), dim3(32 * RVIP_FOLDK_GROUPS), 0, s, ws, nblk, width, K, post);
    return check_launch();
}

// Deferred folds, batched: dst[i] = sum over r of src[r * width + i] for a TABLE of reductions in one launch (blockIdx.y =
// entry).  The bias gradients (bn_bwd_apply) and weight gradients (wgrad slabs) are only read by the optimiser, so their
// ~40 single-purpose fold launches per step collapse into a few.  Fixed summation order per entry -> reproducible.
struct FoldEntry { const float* src; float* dst; int nrows; int stride; long long width; };      // stride: floats between rows (0 = width)

// narrow rows (C floats), many rows: 1024 threads = 32 columns x 32 row groups, double accumulation
__global__ __launch_bounds__(1024) void fold_batch_narrow(const FoldEntry* __restrict__ tab) {
    __shared__ double sh[16][32];
    const FoldEntry en = tab[blockIdx.y];
    if ((long long)blockIdx.x * 32 >= en.width) return;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long col = (long long)blockIdx.x * 32 + c;
    const size_t rstride = en.stride > 0 ? (size_t)en.stride : (size_t)en.width;
    double s = 0.0;
    if (col < en.width) {
        for (int b0 = g; b0 < en.nrows; b0 += 32 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * 32;
                v[u] = b < en.nrows ? en.src[(size_t)b * rstride + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4) s += ((double)v[u] + (double)v[u + 1]) + ((double)v[u + 2] + (double)v[u + 3]);
        }
    }
    const double o = __shfl_xor(s, 32);
    if (!(g & 1)) sh[g >> 1][c] = s + o;
    __syncthreads();
    if (g == 0 && col < en.width) {
        double acc = 0.0;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) acc += sh[gg][c];
        en.dst[col] = (float)acc;
    }
}

// wide rows (9*Cin*Cout floats, a multiple of 4), few to hundreds of rows: 256 threads = 32 float4 columns x 8 row groups
__global__ __launch_bounds__(256) void fold_batch_wide(const FoldEntry* __restrict__ tab) {
    __shared__ float4 sh[8][32];
    const FoldEntry en = tab[blockIdx.y];
    const long long count4 = en.width >> 2;
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float4* src = reinterpret_cast<const float4*>(en.src);
    float4* dst = reinterpret_cast<float4*>(en.dst);
    for (long long base = (long long)blockIdx.x * 32; base < count4; base += (long long)gridDim.x * 32) {
        const long long i = base + e;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (i < count4) {
            for (int k0 = g; k0 < en.nrows; k0 += 8 * 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * 8;
                    v[u] = k < en.nrows ? src[(size_t)k * count4 + i] : float4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        __syncthreads();
        sh[g][e] = acc;
        __syncthreads();
        if (g == 0 && i < count4) {
            float4 t = sh[0][e];
#pragma unroll
            for (int gg = 1; gg < 8; ++gg) { t.x += sh[gg][e].x; t.y += sh[gg][e].y; t.z += sh[gg][e].z; t.w += sh[gg][e].w; }
            dst[i] = t;
        }
    }
}

struct PostSum {                      // out[ch] = total (plain sum), K = 1
    float* out;
    __device__ void run(int ch, const double (&t)[1]) const { out[ch] = (float)t[0]; }
};

// ------------------------------------------------------------------------------------------------
// BatchNormalization forward statistics
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const unsigned char* __restrict__ z, long long rows, int c,
                                                       RedGeom g, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % g.cg, prow = tid / g.cg;
    const bool active = prow < g.rpi;
    const long long r0 = blockIdx.x * g.chunk, r1 = (r0 + g.chunk < rows) ? r0 + g.chunk : rows;
    float part[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) part[0][e] = part[1][e] = 0.f;
    if (active) {
        for (long long r = r0 + prow; r < r1; r += 4 * g.rpi) {
            float v[4][VE];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long rr = r + u * g.rpi;
                if (rr < r1) Vec<T>::load(z + ((size_t)rr * c + cgi * VE) * sizeof(T), v[u]);
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) v[u][e] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VE; ++e) { part[0][e] += v[u][e]; part[1][e] = fmaf(v[u][e], v[u][e], part[1][e]); }
        }
    }
    block_fold<2, VE>(part, active, prow * g.cg + cgi, c, g.rpi, lds, ws + (size_t)blockIdx.x * 2 * c);
}

struct PostBnStats {
    const float* gamma; const float* beta; float* mov_mean; float* mov_var;
    float* mean; float* invstd; float* scale; float* shift;
    double n; float momentum, eps; int unbiased;
    __device__ void run(int ch, const double (&t)[2]) const {
        const double m = t[0] / n;
        double var = t[1] / n - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
        mean[ch] = (float)m;
        invstd[ch] = is;
        const float sc = gm * is;
        scale[ch] = sc;
        shift[ch] = bt - (float)m * sc;
        if (mov_mean) {
            const double vu = (unbiased && n > 1.0) ? var * (n / (n - 1.0)) : var;
            mov_mean[ch] = mov_mean[ch] * momentum + (float)m * (1.f - momentum);
            mov_var[ch] = mov_var[ch] * momentum + (float)vu * (1.f - momentum);
        }
    }
};

__global__ void bn_infer_coeffs_kernel(const float* gamma, const float* beta, const float* mm, const float* mv, float eps,
                                       int c, float* scale, float* shift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c) return;
    const float sc = (gamma ? gamma[i] : 1.f) / sqrtf(mv[i] + eps);
    scale[i] = sc;
    shift[i] = (beta ? beta[i] : 0.f) - mm[i] * sc;
}

// ------------------------------------------------------------------------------------------------
// y = dropout(act(scale*z + shift)) [+ 2x2 max-pool]
// ------------------------------------------------------------------------------------------------
// The forward apply passes walk their tensor from the END (RVIP_REV bit 0, default on; bit 1 = the BN-backward apply pass, default off):
// the igemm in front of the pass has just written the end of z, and the igemm behind it starts at the beginning of y, which this pass
// then wrote last -- a line stays in the 256 MiB Infinity Cache while the bytes moved since its last use fit it (MI355X_MICROARCH.md),
// so at the 256^2 level (134 MB per tensor) two passes in the same direction find nothing of a hand-over and two in opposite directions
// about half.  Measured, same box, five alternating runs: bn_apply 0.450 -> 0.432 ms per step, step 4.149 -> 4.132 ms; the backward
// apply pass reversed: 0.610 -> 0.622 ms (not taken).  Results are the same bits either way (the partial rows keep their chunk's place).
static int rev_flags() {
    static const int f = [] { const char* e = getenv("RVIP_REV"); return e ? atoi(e) : 1; }();
    return f;
}
struct ApplyArgs {
    const unsigned char* z; unsigned char* y; unsigned char* pooled;
    uint16_t* argmax;          // pooled + column-split kernel only: [windows][C / VE], 2 bits per channel = 2 * row + col of the first maximum
    uint8_t* keep_bits;        // un-pooled kernel with dropout: bit planes [C/32][rows] of 32-bit words, bit (c & 31) = element kept (rvip_hip.h)
    const float* scale; const float* shift;
    int act; float inv_keep; uint32_t thr; const uint8_t* mask; const uint32_t* state; int layer_id; int drop;
    int n, h, w, c;
    int rev = 0;               // walk the tensor from its END (see rev_flags)
};

// ACT / DROP: compile-time activation (RVIP_ACT_*) and dropout mode (0 none, 1 counter stream, 2 mask array) of the hot
// configurations; -1 = read the descriptor.  With the descriptor's values the compiler keeps the `switch (act)` and the mode tests
// inside the element loop (a chain of scalar compares and branches per value -- these kernels sit at the VALU / issue limit of an
// HBM stream); the launchers pick a specialised instantiation for what the reference's default graph uses and the generic one otherwise.
// returns the keep bits of the VE elements (bit e = kept; all ones without dropout)
template <typename T, int VE, int ACT = -1, int DROP = -1>
__device__ __forceinline__ unsigned apply_xform(const ApplyArgs& a, size_t e0, const float (&sc)[VE], const float (&sh)[VE],
                                                uint32_t key, float (&v)[VE]) {
    const int act = ACT < 0 ? a.act : ACT;
    const bool drop = DROP < 0 ? a.drop != 0 : DROP > 0, has_mask = DROP < 0 ? a.mask != nullptr : DROP == 2;
    unsigned kb = (1u << VE) - 1u;
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = act_fwd(fmaf(v[e], sc[e], sh[e]), act);
    if (drop) {
        kb = 0;
        if (has_mask) {
#pragma unroll
            for (int e = 0; e < VE; ++e) { const bool k = a.mask[e0 + e] != 0; v[e] = k ? v[e] * a.inv_keep : 0.f; kb |= (k ? 1u : 0u) << e; }
        } else {
            bool keep[VE];
            dropout_keep<VE>(key, e0, a.thr, keep);
#pragma unroll
            for (int e = 0; e < VE; ++e) { v[e] = keep[e] ? v[e] * a.inv_keep : 0.f; kb |= (keep[e] ? 1u : 0u) << e; }
        }
    }
    return kb;
}

// thread = (row slot, channel vector): the channel vector is fixed per thread, so scale/shift live in registers;
// every thread keeps UNR independent 16-byte loads in flight (the kernel is a pure HBM stream).
template <typename T, bool POOL, int ACT = -1, int DROP = -1>
__global__ __launch_bounds__(256) void bn_apply_kernel(ApplyArgs a, int cg, int rpi, long long ngroups) {
    constexpr int VE = Vec<T>::VE;
#ifndef RVIP_APPLY_UNR
#define RVIP_APPLY_UNR 1      // rows per round and thread: 1 / 2 / 4 measured 0.513 / 0.518 / 0.531 ms per step over the 17 launches (the store round size again)
#endif
    constexpr int UNR = POOL ? 2 : RVIP_APPLY_UNR;
    const int tid = threadIdx.x, cv = tid % cg, prow = tid / cg;
    if (prow >= rpi) return;
    float sc[VE], sh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { sc[e] = a.scale ? a.scale[cv * VE + e] : 1.f; sh[e] = a.shift ? a.shift[cv * VE + e] : 0.f; }
    const uint32_t key = (a.drop && !a.mask) ? dropout_key(a.state[RVIP_STATE_SEED], a.state[RVIP_STATE_STEP], (uint32_t)a.layer_id) : 0u;
    const long long G = gridDim.x;
    if constexpr (!POOL) {
        const long long rows = (long long)a.n * a.h * a.w;
        for (long long g = blockIdx.x; g < ngroups; g += UNR * G) {
            float v[UNR][VE]; size_t e0[UNR]; bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long long gi = g + u * G, r = (a.rev ? ngroups - 1 - gi : gi) * rpi + prow;
                ok[u] = gi < ngroups && r < rows;
                e0[u] = (size_t)r * a.c + cv * VE;
                if (ok[u]) Vec<T>::load_nt(a.z + e0[u] * sizeof(T), v[u]);     // z is not read again before the backward pass
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                // (no early `continue`: the lane exchange below needs every lane of the wave)
                unsigned kb = 0;
                if (ok[u]) {
                    kb = apply_xform<T, VE, ACT, DROP>(a, e0[u], sc, sh, key, v[u]);
                    Vec<T>::store(a.y + e0[u] * sizeof(T), v[u]);
                }
                if (a.keep_bits) {                          // wave-uniform: the Dropout backward of the consumer's data gradient reads these
                    // word of a pixel and 32-channel block: channel c sits at bit 8 * ((c & 15) >> 2) + 4 * (c >> 4) + (c & 3)
                    // (rvip_hip.h: the order in which the MFMA epilogue's lanes hold a pixel's channels).  Every thread places its
                    // bits in a word; the threads of a block (adjacent lanes: cv = tid % cg) OR their words; the first one stores.
                    const long long gi = g + u * G, r = (a.rev ? ngroups - 1 - gi : gi) * rpi + prow;
                    constexpr int TPB = 32 / VE;            // threads per 32-channel block: 4 (16-bit types) or 8 (f32)
                    const int m = cv & (TPB - 1);           // this thread's channels are VE * m .. VE * m + VE - 1 of the block
                    unsigned word;                          // (two shifts instead of a loop over the bits: RVIP_BIT_OF_CHANNEL is nibble-wise)
                    if constexpr (VE == 8) word = ((kb & 15u) | ((kb >> 4) << 8)) << (16 * (m & 1) + 4 * (m >> 1));
                    else word = kb << (8 * (m & 3) + 4 * (m >> 2));
                    if (cg > 1) word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0xB1, 0xF, 0xF, true);      // lane ^ 1 (quad permute)
                    if (cg > 2) word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x4E, 0xF, 0xF, true);      // lane ^ 2
                    if constexpr (TPB == 8) { if (cg > 4) word |= (unsigned)__shfl_xor((int)word, 4); }               // (C = 8 / 16: a partial block)
                    if (ok[u] && !(cv & (TPB - 1))) reinterpret_cast<uint32_t*>(a.keep_bits)[(size_t)(cv / TPB) * rows + r] = word;
                }
            }
        }
    } else {
        const int oh = a.h >> 1, ow = a.w >> 1;
        const long long quads = (long long)a.n * oh * ow;
        for (long long g = blockIdx.x; g < ngroups; g += UNR * G) {
            float v[UNR][4][VE]; size_t e0[UNR][4]; bool ok[UNR]; long long qi[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long long q = (g + u * G) * rpi + prow;
                qi[u] = q;
                ok[u] = (g + u * G) < ngroups && q < quads;
                int ox, oy; long long img;
                split_xy(q, ow, oh, ox, oy, img);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e0[u][k] = (((size_t)img * a.h + 2 * oy + (k >> 1)) * a.w + 2 * ox + (k & 1)) * a.c + cv * VE;
                    if (ok[u]) Vec<T>::load_nt(a.z + e0[u][k] * sizeof(T), v[u][k]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (!ok[u]) continue;
                float best[VE];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    apply_xform<T, VE, ACT, DROP>(a, e0[u][k], sc, sh, key, v[u][k]);
                    Vec<T>::store(a.y + e0[u][k] * sizeof(T), v[u][k]);
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        const float vr = Vec<T>::round(v[u][k][e]);                 // pool what was stored
                        best[e] = (k == 0 || vr > best[e]) ? vr : best[e];
                    }
                }
                Vec<T>::store(a.pooled + ((size_t)qi[u] * a.c + cv * VE) * sizeof(T), best);
            }
        }
    }
}

// MaxPooling variant, column-split: thread = (window slot, pixel column, channel vector).  The two lanes `cg` apart hold the two
// columns of a 2x2 window, so every load / store instruction of a wave covers contiguous bytes, and a thread has only 2 x UNR
// loads and stores per round (the window-per-thread form issues 8 + 10: its stores, not its loads or its index arithmetic, hold
// it at 3.5 TB/s -- tools/probe_apply_pool.py).  The window maximum is completed with one lane exchange.  cg must be a power of
// two <= 32 (the two lanes share a wave).
template <typename T, int ACT = -1, int DROP = -1>
__global__ __launch_bounds__(256) void bn_apply_pool2_kernel(ApplyArgs a, int cg, int wpi, long long ngroups) {
    constexpr int VE = Vec<T>::VE;
#ifndef RVIP_POOL2_UNR
#define RVIP_POOL2_UNR 1
#endif
    constexpr int UNR = RVIP_POOL2_UNR;
    const int tid = threadIdx.x, cv = tid % cg, px = (tid / cg) & 1, wslot = tid / (2 * cg);
    float sc[VE], sh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { sc[e] = a.scale ? a.scale[cv * VE + e] : 1.f; sh[e] = a.shift ? a.shift[cv * VE + e] : 0.f; }
    const uint32_t key = (a.drop && !a.mask) ? dropout_key(a.state[RVIP_STATE_SEED], a.state[RVIP_STATE_STEP], (uint32_t)a.layer_id) : 0u;
    const long long G = gridDim.x;
    const int oh = a.h >> 1, ow = a.w >> 1;
    const long long quads = (long long)a.n * oh * ow;
    for (long long g = blockIdx.x; g < ngroups; g += UNR * G) {
        float v[UNR][2][VE]; size_t e0[UNR][2]; bool ok[UNR]; long long qi[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long gi = g + u * G, q = (a.rev ? ngroups - 1 - gi : gi) * wpi + wslot;
            qi[u] = q;
            ok[u] = gi < ngroups && q < quads;                     // the same for both lanes of a window
            int ox, oy; long long img;
            split_xy(q, ow, oh, ox, oy, img);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                e0[u][r] = (((size_t)img * a.h + 2 * oy + r) * a.w + 2 * ox + px) * a.c + cv * VE;
                if (ok[u]) Vec<T>::load_nt(a.z + e0[u][r] * sizeof(T), v[u][r]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (!ok[u]) continue;
            float best[VE];
            unsigned rbits = 0;                                                       // bit e: the lane's column maximum of channel e sits in row 1
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                apply_xform<T, VE, ACT, DROP>(a, e0[u][r], sc, sh, key, v[u][r]);
                Vec<T>::store(a.y + e0[u][r] * sizeof(T), v[u][r]);
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float vr = Vec<T>::round(v[u][r][e]);                     // pool what was stored
                    const bool take = r == 0 || vr > best[e];                       // first maximum wins
                    best[e] = take ? vr : best[e];
                    if (r == 1 && take) rbits |= 1u << e;
                }
            }
            // window position of the first maximum in row-major order (2 * row + col), as rvip_maxpool2x2_bwd finds it from y: this lane
            // holds column px; the other column's candidate wins a tie only if its position is smaller
            const unsigned orbits = (unsigned)__shfl_xor((int)rbits, cg);
            unsigned arg = 0;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float ob = __shfl_xor(best[e], cg);
                const int mine = 2 * (int)((rbits >> e) & 1u) + px, theirs = 2 * (int)((orbits >> e) & 1u) + (px ^ 1);
                const bool keep = best[e] > ob || (best[e] == ob && mine < theirs);
                arg |= (unsigned)(keep ? mine : theirs) << (2 * e);
                best[e] = fmaxf(best[e], ob);
            }
            if (px == 0) {
                Vec<T>::store(a.pooled + ((size_t)qi[u] * a.c + cv * VE) * sizeof(T), best);
                if (a.argmax) a.argmax[(size_t)qi[u] * cg + cv] = (uint16_t)arg;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward of conv -> [act] -> BN -> [act] -> dropout
// ------------------------------------------------------------------------------------------------
#ifndef RVIP_BWD_U
#define RVIP_BWD_U 2             // rows in flight per thread in the BN-backward reduce pass
#endif
#ifndef RVIP_BWD_U_APPLY
#define RVIP_BWD_U_APPLY 1       // ... and in the apply pass, which also stores (measured 1 / 2 / 4: 0.733 / 0.761 / 0.91 ms per step; reduce 0.551 / 0.536 / 0.63)
#endif
struct BnBwdArgs {
    const unsigned char* dy; const unsigned char* z; unsigned char* dz;
    // MaxPooling2D backward folded into both passes (dp != NULL): the gradient reaching the stage output is
    // round(route(dp -> argmax position of the 2x2 window) + dy), dy = the skip-connection gradient or NULL
    const unsigned char* dp; const uint16_t* argmax; int h, w;
    int lw, lh;                                  // log2(w), log2(h) when both are powers of two (shift / mask pixel coordinates), else -1
    const float* mean; const float* invstd; const float* scale; const float* shift; const float* coef;
    int act, act_after_bn, has_bn;
    float inv_keep; uint32_t thr; const uint8_t* mask; const uint32_t* state; int layer_id; int drop;
    long long rows; int c;
    int rev = 0;                                 // chunk of a workgroup counted from the END of the tensor (see rev_flags)
};

// g = dL/d(BN-side output before dropout) [times act'(y) when the activation follows BN]
template <typename T, int VE, int ACT = -1, int DROP = -1, int AFTER = -1>
__device__ __forceinline__ void xform_g(const BnBwdArgs& a, size_t e0, int cbase, uint32_t key, const float (&z)[VE], float (&g)[VE]) {
    const int act = ACT < 0 ? a.act : ACT;
    const bool drop = DROP < 0 ? a.drop != 0 : DROP > 0, has_mask = DROP < 0 ? a.mask != nullptr : DROP == 2;
    const bool after = AFTER < 0 ? a.act_after_bn != 0 : AFTER > 0;
    if (drop) {
        if (has_mask) {
#pragma unroll
            for (int e = 0; e < VE; ++e) g[e] = a.mask[e0 + e] ? g[e] * a.inv_keep : 0.f;
        } else {
            bool keep[VE];
            dropout_keep<VE>(key, e0, a.thr, keep);
#pragma unroll
            for (int e = 0; e < VE; ++e) g[e] = keep[e] ? g[e] * a.inv_keep : 0.f;
        }
    }
    if (after) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float sc = a.scale ? a.scale[cbase + e] : 1.f, sh = a.shift ? a.shift[cbase + e] : 0.f;
            g[e] *= act_bwd(act_fwd(fmaf(z[e], sc, sh), act), act);
        }
    }
}

// Incoming gradient of pixel row rr, channel vector cgi: dy, or -- pooled stage (PARG) -- what rvip_maxpool2x2_bwd would have stored.
// Two phases so that a thread's loads of all its rows are in flight together: gy_issue only loads (no branch: the window index
// is clamped, validity is a flag), gy_finish routes.  PARG: compile-time (0 / 1) in the specialised instantiations, -1 = a.dp.
template <int VE> struct GyRaw { float add[VE]; float d[VE]; unsigned arg, pos; bool sel; };
template <typename T, int VE, int PARG, bool NT>
__device__ __forceinline__ void gy_issue(const BnBwdArgs& a, long long rr, int cgi, int cg, size_t e0, GyRaw<VE>& q) {
    const bool parg = PARG < 0 ? a.dp != nullptr : PARG > 0;
    if (!parg) {
        if constexpr (NT) Vec<T>::load_nt(a.dy + e0 * sizeof(T), q.add);
        else Vec<T>::load(a.dy + e0 * sizeof(T), q.add);
        return;
    }
    int x, y; long long img;
    if (a.lw >= 0) { x = (int)(rr & (a.w - 1)); y = (int)((rr >> a.lw) & (a.h - 1)); img = rr >> (a.lw + a.lh); }
    else split_xy(rr, a.w, a.h, x, y, img);
    const int oh = a.h >> 1, ow = a.w >> 1;
    if (a.dy) Vec<T>::load_nt(a.dy + e0 * sizeof(T), q.add);
    else {
#pragma unroll
        for (int e = 0; e < VE; ++e) q.add[e] = 0.f;
    }
    const int wy = (y >> 1) < oh ? (y >> 1) : oh - 1, wx = (x >> 1) < ow ? (x >> 1) : ow - 1;      // odd extents: the last row / column has no window
    q.sel = (y >> 1) < oh && (x >> 1) < ow;
    const size_t w = ((size_t)img * oh + wy) * ow + wx;
    Vec<T>::load(a.dp + (w * a.c + cgi * VE) * sizeof(T), q.d);
    q.arg = a.argmax[w * cg + cgi];
    q.pos = 2u * (y & 1) + (x & 1);
}
template <typename T, int VE, int PARG>
__device__ __forceinline__ void gy_finish(const BnBwdArgs& a, const GyRaw<VE>& q, float (&g)[VE]) {
    const bool parg = PARG < 0 ? a.dp != nullptr : PARG > 0;
#pragma unroll
    for (int e = 0; e < VE; ++e)
        g[e] = parg ? Vec<T>::round(q.add[e] + ((q.sel && ((q.arg >> (2 * e)) & 3u) == q.pos) ? q.d[e] : 0.f)) : q.add[e];
}

template <typename T, int ACT = -1, int DROP = -1, int AFTER = -1, int PARG = -1>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a, RedGeom gm, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % gm.cg, prow = tid / gm.cg;
    const bool active = prow < gm.rpi;
    const long long r0 = blockIdx.x * gm.chunk, r1 = (r0 + gm.chunk < a.rows) ? r0 + gm.chunk : a.rows;
    const uint32_t key = (a.drop && !a.mask) ? dropout_key(a.state[RVIP_STATE_SEED], a.state[RVIP_STATE_STEP], (uint32_t)a.layer_id) : 0u;
    float part[2][VE], mu[VE], is[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[0][e] = part[1][e] = 0.f;
        mu[e] = active ? a.mean[cgi * VE + e] : 0.f;
        is[e] = active ? a.invstd[cgi * VE + e] : 0.f;
    }
    if (active) {
        for (long long r = r0 + prow; r < r1; r += RVIP_BWD_U * gm.rpi) {
            float z[RVIP_BWD_U][VE], g[RVIP_BWD_U][VE]; size_t e0[RVIP_BWD_U]; bool ok[RVIP_BWD_U];
            GyRaw<VE> raw[RVIP_BWD_U];
#pragma unroll
            for (int u = 0; u < RVIP_BWD_U; ++u) {
                const long long rr = r + u * gm.rpi;
                ok[u] = rr < r1;
                e0[u] = (size_t)rr * a.c + cgi * VE;
                if (ok[u]) { Vec<T>::load(a.z + e0[u] * sizeof(T), z[u]); gy_issue<T, VE, PARG, false>(a, rr, cgi, gm.cg, e0[u], raw[u]); }
            }
#pragma unroll
            for (int u = 0; u < RVIP_BWD_U; ++u) {
                if (!ok[u]) continue;
                gy_finish<T, VE, PARG>(a, raw[u], g[u]);
                xform_g<T, VE, ACT, DROP, AFTER>(a, e0[u], cgi * VE, key, z[u], g[u]);
#pragma unroll
                for (int e = 0; e < VE; ++e) { part[0][e] += g[u][e]; part[1][e] = fmaf(g[u][e], (z[u][e] - mu[e]) * is[e], part[1][e]); }
            }
        }
    }
    block_fold<2, VE>(part, active, prow * gm.cg + cgi, a.c, gm.rpi, lds, ws + (size_t)blockIdx.x * 2 * a.c);
}

struct PostBnBwd {
    const float* gamma; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; float* coef; double n; int c;
    __device__ void run(int ch, const double (&t)[2]) const {
        const float db = (float)t[0], dg = (float)t[1];
        dbeta[ch] = db;
        dgamma[ch] = dg;
        const float gm = gamma[ch], is = invstd[ch], mu = mean[ch];
        const float c1 = gm * is;
        const float c2 = -gm * is * is * (float)(t[1] / n);
        const float c3 = -gm * is * (float)(t[0] / n) - c2 * mu;
        coef[ch] = c1; coef[c + ch] = c2; coef[2 * c + ch] = c3;
    }
};

template <typename T, int ACT = -1, int DROP = -1, int AFTER = -1, int PARG = -1>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a, RedGeom gm, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % gm.cg, prow = tid / gm.cg;
    const bool active = prow < gm.rpi;
    const long long cb = a.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;      // this workgroup's chunk (its partial row keeps the chunk's place)
    const long long r0 = cb * gm.chunk, r1 = (r0 + gm.chunk < a.rows) ? r0 + gm.chunk : a.rows;
    const uint32_t key = (a.drop && !a.mask) ? dropout_key(a.state[RVIP_STATE_SEED], a.state[RVIP_STATE_STEP], (uint32_t)a.layer_id) : 0u;
    float part[1][VE], c1[VE], c2[VE], c3[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[0][e] = 0.f;
        const int ch = cgi * VE + e;
        c1[e] = (a.has_bn && active) ? a.coef[ch] : 1.f;
        c2[e] = (a.has_bn && active) ? a.coef[a.c + ch] : 0.f;
        c3[e] = (a.has_bn && active) ? a.coef[2 * a.c + ch] : 0.f;
    }
    if (active) {
        for (long long r = r0 + prow; r < r1; r += RVIP_BWD_U_APPLY * gm.rpi) {
            float z[RVIP_BWD_U_APPLY][VE], g[RVIP_BWD_U_APPLY][VE]; size_t e0[RVIP_BWD_U_APPLY]; bool ok[RVIP_BWD_U_APPLY];
            GyRaw<VE> raw[RVIP_BWD_U_APPLY];
#pragma unroll
            for (int u = 0; u < RVIP_BWD_U_APPLY; ++u) {
                const long long rr = r + u * gm.rpi;
                ok[u] = rr < r1;
                e0[u] = (size_t)rr * a.c + cgi * VE;
                // last reader of both tensors: non-temporal loads leave the cache to dz, which the weight / data gradient
                // kernels read next (measured: -0.05 ms per step, all of it in those kernels)
                if (ok[u]) { Vec<T>::load_nt(a.z + e0[u] * sizeof(T), z[u]); gy_issue<T, VE, PARG, true>(a, rr, cgi, gm.cg, e0[u], raw[u]); }
            }
#pragma unroll
            for (int u = 0; u < RVIP_BWD_U_APPLY; ++u) {
                if (!ok[u]) continue;
                gy_finish<T, VE, PARG>(a, raw[u], g[u]);
                xform_g<T, VE, ACT, DROP, AFTER>(a, e0[u], cgi * VE, key, z[u], g[u]);
                float d[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    float t = fmaf(c1[e], g[u][e], fmaf(c2[e], z[u][e], c3[e]));
                    if (!(AFTER < 0 ? a.act_after_bn != 0 : AFTER > 0)) t *= act_bwd(z[u][e], ACT < 0 ? a.act : ACT);   // z is the activation output here
                    d[e] = t;
                    part[0][e] += Vec<T>::round(t);                    // bias grad of what the wgrad kernels read
                }
                Vec<T>::store(a.dz + e0[u] * sizeof(T), d);
            }
        }
    }
    block_fold<1, VE>(part, active, prow * gm.cg + cgi, a.c, gm.rpi, lds, ws + (size_t)cb * a.c);
}

// ------------------------------------------------------------------------------------------------
// MaxPooling2D backward (+ skip gradient), UpSampling2D forward / backward
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ y, const unsigned char* __restrict__ dp,
                                                          const unsigned char* __restrict__ add, unsigned char* __restrict__ dx,
                                                          int n, int h, int w, int c) {
    constexpr int VE = Vec<T>::VE;
    const int cg = c / VE, oh = h >> 1, ow = w >> 1;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    const int cv = (int)(idx % cg);
    const long long q = idx / cg;
    if (q >= (long long)n * oh * ow) return;
    int ox, oy; long long img;
    split_xy(q, ow, oh, ox, oy, img);
    float v[4][VE], g[VE];
    size_t e0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e0[k] = (((size_t)img * h + 2 * oy + (k >> 1)) * w + 2 * ox + (k & 1)) * c + cv * VE;
        Vec<T>::load(y + e0[k] * sizeof(T), v[k]);
    }
    Vec<T>::load(dp + ((size_t)q * c + cv * VE) * sizeof(T), g);
    int arg[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        int best = 0; float bv = v[0][e];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k][e] > bv) { bv = v[k][e]; best = k; }
        arg[e] = best;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o[VE];
        if (add) Vec<T>::load(add + e0[k] * sizeof(T), o);
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] = (add ? o[e] : 0.f) + (arg[e] == k ? g[e] : 0.f);
        Vec<T>::store(dx + e0[k] * sizeof(T), o);
    }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void upsample_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                       int n, int h, int w, int c) {   // h, w = LOW resolution
    constexpr int VE = Vec<T>::VE;
    const int cg = c / VE;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    const int cv = (int)(idx % cg);
    const long long q = idx / cg;
    if (q >= (long long)n * h * w) return;
    int x, yy; long long img;
    split_xy(q, w, h, x, yy, img);
    const size_t lo = ((size_t)q * c + cv * VE) * sizeof(T);
    float v[VE];
    if constexpr (!BWD) {
        Vec<T>::load(src + lo, v);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            Vec<T>::store(dst + ((((size_t)img * 2 * h + 2 * yy + (k >> 1)) * 2 * w + 2 * x + (k & 1)) * c + cv * VE) * sizeof(T), v);
    } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t[VE];
            Vec<T>::load(src + ((((size_t)img * 2 * h + 2 * yy + (k >> 1)) * 2 * w + 2 * x + (k & 1)) * c + cv * VE) * sizeof(T), t);
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] += t[e];
        }
        Vec<T>::store(dst + lo, v);
    }
}

// dst[n][i][j] = src[n][2i+1][2j+1]: data gradient of the zero-stuffed read (Conv2DTranspose backward)
template <typename T>
__global__ __launch_bounds__(256) void subsample_odd_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                            int n, int h, int w, int c) {   // h, w = LOW resolution
    constexpr int VE = Vec<T>::VE;
    const int cg = c / VE;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    const int cv = (int)(idx % cg);
    const long long q = idx / cg;
    if (q >= (long long)n * h * w) return;
    int x, yy; long long img;
    split_xy(q, w, h, x, yy, img);
    float v[VE];
    Vec<T>::load(src + ((((size_t)img * 2 * h + 2 * yy + 1) * 2 * w + 2 * x + 1) * c + cv * VE) * sizeof(T), v);
    Vec<T>::store(dst + ((size_t)q * c + cv * VE) * sizeof(T), v);
}

// ------------------------------------------------------------------------------------------------
// head: 1x1 conv + sigmoid, loss sums, loss gradient, backward
// ------------------------------------------------------------------------------------------------
#define RVIP_MAXK 4
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const unsigned char* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ pred,
                                                       const float* __restrict__ yt, long long rows, int cin, int k,
                                                       long long chunk, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float red[4][16];
    const int tid = threadIdx.x;
    const long long r0 = blockIdx.x * chunk, r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float s[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) s[i] = 0.f;
    for (long long r = r0 + tid; r < r1; r += 256) {
        float lg[RVIP_MAXK];
#pragma unroll
        for (int kk = 0; kk < RVIP_MAXK; ++kk) lg[kk] = (kk < k && b) ? b[kk] : 0.f;
        for (int c = 0; c < cin; c += VE) {
            float v[VE];
            Vec<T>::load(x + ((size_t)r * cin + c) * sizeof(T), v);
#pragma unroll
            for (int e = 0; e < VE; ++e)
#pragma unroll
                for (int kk = 0; kk < RVIP_MAXK; ++kk) if (kk < k) lg[kk] = fmaf(v[e], w[(c + e) * k + kk], lg[kk]);
        }
#pragma unroll
        for (int kk = 0; kk < RVIP_MAXK; ++kk) {
            if (kk >= k) continue;
            const float zl = lg[kk];
            // one exponential serves the sigmoid and the softplus term: e = exp(-|z|), sigmoid = (z >= 0 ? 1 : e) / (1 + e),
            // log1p(e) = log(1 + e) with absolute error < 1e-7 (e <= 1).  Hardware exp2 / log2 / rcp: the libm forms made
            // this kernel VALU-bound (84 us for 134 MB).
            const float e = __expf(-fabsf(zl));
            const float inv = __frcp_rn(1.f + e);
            const float p = (zl >= 0.f ? 1.f : e) * inv;
            pred[(size_t)r * k + kk] = p;
            if (yt) {
                const float t = yt[(size_t)r * k + kk];
                const float d = p - t;
                s[0] = fmaf(d, d, s[0]);
                if (kk >= k - 3) {           // [..., -3:]: BCE-Dice and dice_coef_labels drop the background channel of a 4-class head
                    s[1] += fmaxf(zl, 0.f) - zl * t + __logf(1.f + e);
                    s[2] = fmaf(t, p, s[2]); s[3] += t; s[4] += p;
                }
                if (kk == k - 2) { s[5] = fmaf(t, p, s[5]); s[6] += t; s[7] += p; }
                if (kk == k - 1) { s[8] = fmaf(t, p, s[8]); s[9] += t; s[10] += p; }
            }
        }
    }
    if (!ws) return;
#pragma unroll
    for (int i = 0; i < 11; ++i) s[i] = wave_sum(s[i]);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 11; ++i) red[wv][i] = s[i];
    }
    __syncthreads();
    if (tid < 16) ws[(size_t)blockIdx.x * 16 + tid] = tid < 11 ? red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] : 0.f;
}

__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ pred, const float* __restrict__ yt,
                                                        const float* __restrict__ sums, float* __restrict__ dlogit,
                                                        float* __restrict__ loss_out, long long count, int k, int loss_kind,
                                                        float inv_count, float lg, float w_bce, float w_dice) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    const float inter = sums[2], st = sums[3], sp = sums[4];
    const float den = st + sp + 1.f;
    if (i == 0 && loss_out) {
        if (loss_kind == RVIP_LOSS_MSE) loss_out[0] = sums[0] * inv_count;
        else loss_out[0] = w_bce * sums[1] * inv_count - w_dice * lg * (2.f * inter + 1.f) / den;
    }
    if (i >= count) return;
    const float p = pred[i], t = yt[i];
    float d;
    if (loss_kind == RVIP_LOSS_MSE) {
        d = 2.f * (p - t) * inv_count * p * (1.f - p);
    } else {
        const float ddice = (2.f * t * den - (2.f * inter + 1.f)) / (den * den);
        d = w_bce * (p - t) * inv_count - w_dice * lg * ddice * p * (1.f - p);
        if ((int)(i % k) < k - 3) d = 0.f;       // Loss_and_metrics.py:222-224 / :240-242: the background channel is sliced off
    }
    dlogit[i] = d;
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const unsigned char* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ dl, unsigned char* __restrict__ dx,
                                                       long long rows, int cin, int k, RedGeom gm, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % gm.cg, prow = tid / gm.cg;
    const bool active = prow < gm.rpi;
    const long long r0 = blockIdx.x * gm.chunk, r1 = (r0 + gm.chunk < rows) ? r0 + gm.chunk : rows;
    float part[2 * RVIP_MAXK][VE];
#pragma unroll
    for (int q = 0; q < 2 * RVIP_MAXK; ++q)
#pragma unroll
        for (int e = 0; e < VE; ++e) part[q][e] = 0.f;
    float wr[VE][RVIP_MAXK];
#pragma unroll
    for (int e = 0; e < VE; ++e)
#pragma unroll
        for (int kk = 0; kk < RVIP_MAXK; ++kk) wr[e][kk] = (active && kk < k) ? w[(cgi * VE + e) * k + kk] : 0.f;
    if (active) {
        for (long long r = r0 + prow; r < r1; r += gm.rpi) {
            float d[RVIP_MAXK];
#pragma unroll
            for (int kk = 0; kk < RVIP_MAXK; ++kk) d[kk] = kk < k ? dl[(size_t)r * k + kk] : 0.f;
            float v[VE], o[VE];
            Vec<T>::load(x + ((size_t)r * cin + cgi * VE) * sizeof(T), v);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int kk = 0; kk < RVIP_MAXK; ++kk) { acc = fmaf(d[kk], wr[e][kk], acc); part[kk][e] = fmaf(v[e], d[kk], part[kk][e]); }
                o[e] = acc;
            }
            if (cgi == 0) {
#pragma unroll
                for (int kk = 0; kk < RVIP_MAXK; ++kk) part[RVIP_MAXK + kk][0] += d[kk];
            }
            if (dx) Vec<T>::store(dx + ((size_t)r * cin + cgi * VE) * sizeof(T), o);
        }
    }
    block_fold<2 * RVIP_MAXK, VE>(part, active, prow * gm.cg + cgi, cin, gm.rpi, lds, ws + (size_t)blockIdx.x * 2 * RVIP_MAXK * cin);
}

struct PostHeadBwd {                   // ws columns: [2*kmax][cin]: rows 0..kmax-1 = dw per class, kmax.. = db (channel 0 only)
    float* dw; float* db; int cin, k, kmax;
    __device__ void run_k(int ch, int kk, double t) const {
        if (kk < kmax) { if (kk < k) dw[ch * k + kk] = (float)t; }
        else if (ch == 0 && kk - kmax < k) db[kk - kmax] = (float)t;
    }
};

// ------------------------------------------------------------------------------------------------
// Last stage fused with the 1x1 sigmoid head (Unets.py:128): the stage's BN output y is consumed by nothing else, so
//   forward : pred = sigmoid(W_h^T y + b_h) with y = act(scale*z + shift) built in registers   (no y tensor)
//   backward: the gradient reaching y is g[p][c] = sum_k W_h[c][k] * dlogit[p][k], rebuilt in both BN-backward stages
//             (no gy tensor); dW_h[c][k] = sum_p y[p][c] * dlogit[p][k] and db_h ride along in the reduce stage.
// Thread = (channel vector cgi, pixel slot prow); the cg = C / VE lanes of a pixel are adjacent (cg a power of two).
// ------------------------------------------------------------------------------------------------
// dlogit == NULL: the logit gradient is not materialised (BCE-Dice form of the fused last stage): it is rebuilt per pixel from the
// stored heat-map and target,  d = ca * (p - t) + (cb * t + cc) * p (1 - p),  with the three global coefficients rvip_head_mse_coef
// wrote to dcoef (head_lazy_begin loads them)
struct HeadFuse { const float* w; const float* b; const float* dlogit; int k;
                  const float* pred = nullptr; const float* yt = nullptr; const float* dcoef = nullptr; float ca = 0.f, cb = 0.f, cc = 0.f; };
__device__ __forceinline__ HeadFuse head_lazy_begin(HeadFuse hd) {
    if (!hd.dlogit && hd.dcoef) { hd.ca = hd.dcoef[0]; hd.cb = hd.dcoef[1]; hd.cc = hd.dcoef[2]; }
    return hd;
}
__device__ __forceinline__ float head_lazy_d(const HeadFuse& hd, float p, float t) {
    return fmaf(hd.ca, p - t, fmaf(hd.cb, t, hd.cc) * (p * (1.f - p)));
}

// KK = class capacity of the instantiation (2 for the reference's two heat-maps): half the shuffles and logit registers of the
// MAXK-sized form; the target values of the round are loaded with its z rows, not after the arithmetic that depends on them.
// MSE = 1 (include/rvip_hip.h: rvip_bn_apply_head_mse): the MSE logit gradient d = 2 (p - t) inv_count p (1 - p) [* dscale] is
// written here, and S_kk[c] = sum (y[c] - beta[c]) * d[kk], Q_kk = sum d[kk] are accumulated while y is in registers ->
// mse.rows[block][KK + 1][C].  y - beta = gamma * xhat: the rows are sums of g * xhat up to gamma, so the stage's dgamma needs no
// `sum g*y - beta * sum g` difference (which loses what the float partial sums cannot hold when the gradient is mostly common-mode);
// the head's weight gradient adds beta * Q back, where nothing cancels.
struct HeadMse { float* dlogit; float* rows; const float* beta; float inv_count, dscale; };
// FAST: the default graph's geometry at compile time -- four 16-byte channel vectors per pixel (C = 32 in the 16-bit types) and exactly
// KK classes: the generic form's run-time loops over lanes and classes were 70 branches and 74 ds_bpermute in a VALU-bound kernel.
template <typename T, int KK, int SPEC = 0, int MSE = 0, int FAST = 0>
__global__ __launch_bounds__(256) void bn_apply_head_kernel(ApplyArgs a, HeadFuse hd, float* __restrict__ pred, const float* __restrict__ yt,
                                                            long long rows, long long chunk, int reduce, float* __restrict__ ws, HeadMse mse) {
    constexpr int VE = Vec<T>::VE, U = 2;
    __shared__ float red[4][16];
    const int tid = threadIdx.x, k = FAST ? KK : hd.k;
    const int cg = FAST ? 4 : a.c / VE, rpi = 256 / cg, cgi = tid % cg, prow = tid / cg;
    const long long r0 = blockIdx.x * chunk, r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float sc[VE], sh[VE], wr[VE][KK], bias[KK];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        sc[e] = a.scale ? a.scale[cgi * VE + e] : 1.f; sh[e] = a.shift ? a.shift[cgi * VE + e] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wr[e][kk] = kk < k ? hd.w[(cgi * VE + e) * k + kk] : 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) bias[kk] = (kk < k && hd.b) ? hd.b[kk] : 0.f;
    float s[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) s[i] = 0.f;
    // MSE == 1: the logit gradient is written here and ONE row set S, Q is kept (see HeadMse).  MSE == 2 (BCE-Dice): the gradient is
    // d = ca (p - t) + cb t p(1-p) + cc p(1-p) with coefficients that need the batch's Dice sums first, so THREE row sets are kept, one
    // per pixel term  t0 = p - t,  t1 = t p (1 - p),  t2 = p (1 - p):  rows j * KK + kk = S_{j,kk}[c], row 3 KK = Q_{j,kk} in column
    // j * KK + kk; nothing is written per pixel beyond the heat-map (rvip_head_mse_coef combines the sets, rvip_bn_bwd_apply_head_lazy
    // rebuilds d from the heat-map and the target).
    constexpr int NT = MSE == 2 ? 3 : 1, NR = MSE ? NT * KK + 1 : 1;
    static_assert(!MSE || NT * KK <= VE, "the Q row holds one column per (term, class)");
    float hp[NR][VE], bt[MSE ? VE : 1];
    if constexpr (MSE) {
#pragma unroll
        for (int e = 0; e < VE; ++e) bt[e] = mse.beta ? mse.beta[cgi * VE + e] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int e = 0; e < VE; ++e) hp[q][e] = 0.f;
    for (long long rb = r0 + prow; rb < r1; rb += (long long)U * rpi) {
        float v[U][VE], tv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = rb + (long long)u * rpi;
            tv[u] = (yt && r < r1 && cgi < k) ? yt[(size_t)r * k + cgi] : 0.f;      // lane cgi finishes class cgi (+ cg, ... below)
            if (r < r1) Vec<T>::load_nt(a.z + ((size_t)r * a.c + cgi * VE) * sizeof(T), v[u]);      // z is not read again before the backward pass
            else {
#pragma unroll
                for (int e = 0; e < VE; ++e) v[u][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = rb + (long long)u * rpi;
            float lg[KK], dl[MSE ? NT * KK : 1];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) lg[kk] = 0.f;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float y = Vec<T>::round(act_fwd(fmaf(v[u][e], sc[e], sh[e]), SPEC ? RVIP_ACT_NONE : a.act));   // what rvip_bn_apply would have stored
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) lg[kk] = fmaf(y, wr[e][kk], lg[kk]);
                if constexpr (MSE) v[u][e] = y;              // y takes z's register: the rows below read it again (the kernel is VALU-bound)
            }
            if constexpr (MSE) {
#pragma unroll
                for (int kk = 0; kk < NT * KK; ++kk) dl[kk] = 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (cg == 4) {                                       // 32 channels: the four lanes of a pixel are one DPP quad
                    lg[kk] += lane_xor2_dpp(lg[kk]);
                    lg[kk] += lane_xor1_dpp(lg[kk]);
                } else {
                    for (int o = cg >> 1; o > 0; o >>= 1) lg[kk] += __shfl_xor(lg[kk], o);
                }
                lg[kk] += bias[kk];
            }
            if (!MSE && r >= r1) continue;
            for (int kk = cgi; kk < k && r < r1; kk += cg) {
                float zl = lg[0];
#pragma unroll
                for (int q = 1; q < KK; ++q) zl = (kk == q) ? lg[q] : zl;
                const float e_ = __expf(-fabsf(zl));
                const float inv = __frcp_rn(1.f + e_);
                const float pv = (zl >= 0.f ? 1.f : e_) * inv;
                pred[(size_t)r * k + kk] = pv;
                if (yt) {
                    const float t = kk == cgi ? tv[u] : yt[(size_t)r * k + kk];
                    const float d = pv - t;
                    if constexpr (MSE == 1) {
                        float dd = 2.f * (pv - t) * mse.inv_count * pv * (1.f - pv);          // rvip_head_grad's expression, then rvip_scale_f32's
                        if (mse.dscale != 1.f) dd *= mse.dscale;
                        mse.dlogit[(size_t)r * k + kk] = dd;
#pragma unroll
                        for (int q = 0; q < KK; ++q) dl[q] = (kk == q) ? dd : dl[q];
                    }
                    if constexpr (MSE == 2) {
                        const float pq = pv * (1.f - pv);
#pragma unroll
                        for (int q = 0; q < KK; ++q) {
                            dl[q] = (kk == q) ? d : dl[q];
                            dl[KK + q] = (kk == q) ? t * pq : dl[KK + q];
                            dl[2 * KK + q] = (kk == q) ? pq : dl[2 * KK + q];
                        }
                    }
                    s[0] = fmaf(d, d, s[0]);
                    if (kk >= k - 3) {
                        s[1] += fmaxf(zl, 0.f) - zl * t + __logf(1.f + e_);
                        s[2] = fmaf(t, pv, s[2]); s[3] += t; s[4] += pv;
                    }
                    if (kk == k - 2) { s[5] = fmaf(t, pv, s[5]); s[6] += t; s[7] += pv; }
                    if (kk == k - 1) { s[8] = fmaf(t, pv, s[8]); s[9] += t; s[10] += pv; }
                }
            }
            if constexpr (MSE) {
                // every class was finished by exactly one lane of the pixel: a sum over its cg lanes hands d[.] to all of them
#pragma unroll
                for (int kk = 0; kk < NT * KK; ++kk) {
                    if (cg == 4) {
                        dl[kk] += lane_xor2_dpp(dl[kk]);
                        dl[kk] += lane_xor1_dpp(dl[kk]);
                    } else {
                        for (int o = cg >> 1; o > 0; o >>= 1) dl[kk] += __shfl_xor(dl[kk], o);
                    }
                    if (cgi == 0) hp[NT * KK][kk] += dl[kk];
                }
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float yb = v[u][e] - bt[e];
#pragma unroll
                    for (int kk = 0; kk < NT * KK; ++kk) hp[kk][e] = fmaf(yb, dl[kk], hp[kk][e]);
                }
            }
        }
    }
    if constexpr (MSE) {
        __shared__ float fold_lds[256 * VE];
        block_fold<NR, VE>(hp, true, tid, a.c, rpi, fold_lds, mse.rows + (size_t)blockIdx.x * NR * a.c);
    }
    if (!reduce) return;
#pragma unroll
    for (int i = 0; i < 11; ++i) s[i] = wave_sum(s[i]);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 11; ++i) red[wv][i] = s[i];
    }
    __syncthreads();
    if (tid < 16) ws[(size_t)blockIdx.x * 16 + tid] = tid < 11 ? red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] : 0.f;
}

// incoming gradient of pixel r for this thread's channels, from the head's logit gradient
template <typename T, int VE, int KK = RVIP_MAXK>
__device__ __forceinline__ void head_grad_vec(const HeadFuse& hd, long long r, const float (&wr)[VE][KK], float (&d)[KK], float (&g)[VE]) {
    if (hd.dlogit) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) d[kk] = kk < hd.k ? hd.dlogit[(size_t)r * hd.k + kk] : 0.f;
    } else if (KK == 2 && hd.k == 2) {                            // both classes of the pixel in one 8-byte load per tensor
        const float2 p2 = *reinterpret_cast<const float2*>(hd.pred + (size_t)r * 2), t2 = *reinterpret_cast<const float2*>(hd.yt + (size_t)r * 2);
        d[0] = head_lazy_d(hd, p2.x, t2.x);
        d[KK - 1] = head_lazy_d(hd, p2.y, t2.y);
    } else {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) d[kk] = kk < hd.k ? head_lazy_d(hd, hd.pred[(size_t)r * hd.k + kk], hd.yt[(size_t)r * hd.k + kk]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) acc = fmaf(d[kk], wr[e][kk], acc);
        g[e] = Vec<T>::round(acc);                                // the gy tensor the unfused path stores in the activation dtype
    }
}

// KK = class capacity of this instantiation (2 for the reference's two heat-maps, RVIP_MAXK otherwise): the per-class partials are
// KK x VE registers each way, and with the MAXK-sized arrays the kernel had one 16-byte load in flight per thread at ~170 VGPRs --
// latency-bound at 1.7 TB/s.  U rows in flight per thread.
template <typename T, int KK, int SPEC = 0>
__global__ __launch_bounds__(256) void bn_bwd_reduce_head_kernel(BnBwdArgs a, HeadFuse hd, RedGeom gm, float* __restrict__ ws_bn, float* __restrict__ ws_hd) {
    constexpr int VE = Vec<T>::VE, U = 2;
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % gm.cg, prow = tid / gm.cg;
    const bool active = prow < gm.rpi;
    const long long r0 = blockIdx.x * gm.chunk, r1 = (r0 + gm.chunk < a.rows) ? r0 + gm.chunk : a.rows;
    float part[2][VE], hpart[2 * KK][VE], mu[VE], is[VE], sc[VE], sh[VE], wr[VE][KK];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[0][e] = part[1][e] = 0.f;
        const int ch = cgi * VE + e;
        mu[e] = active ? a.mean[ch] : 0.f; is[e] = active ? a.invstd[ch] : 0.f;
        sc[e] = (active && a.scale) ? a.scale[ch] : 1.f; sh[e] = (active && a.shift) ? a.shift[ch] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wr[e][kk] = (active && kk < hd.k) ? hd.w[ch * hd.k + kk] : 0.f;
#pragma unroll
        for (int q = 0; q < 2 * KK; ++q) hpart[q][e] = 0.f;
    }
    if (active) {
        for (long long rb = r0 + prow; rb < r1; rb += (long long)U * gm.rpi) {
            float z[U][VE], d[U][KK];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                          // all loads of the round first
                const long long r = rb + (long long)u * gm.rpi;
                ok[u] = r < r1;
                if (ok[u]) {
                    Vec<T>::load(a.z + ((size_t)r * a.c + cgi * VE) * sizeof(T), z[u]);
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) d[u][kk] = kk < hd.k ? hd.dlogit[(size_t)r * hd.k + kk] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const long long r = rb + (long long)u * gm.rpi;
                const size_t e0 = (size_t)r * a.c + cgi * VE;
                float g[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    float acc = 0.f;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) acc = fmaf(d[u][kk], wr[e][kk], acc);
                    g[e] = Vec<T>::round(acc);                    // the gy tensor the unfused path stores in the activation dtype
                    const float y = Vec<T>::round(act_fwd(fmaf(z[u][e], sc[e], sh[e]), (!SPEC && a.act_after_bn) ? a.act : RVIP_ACT_NONE));
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) hpart[kk][e] = fmaf(y, d[u][kk], hpart[kk][e]);     // dW_h
                }
                if (cgi == 0) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) hpart[KK + kk][0] += d[u][kk];                       // db_h
                }
                if constexpr (!SPEC) xform_g<T, VE>(a, e0, cgi * VE, 0u, z[u], g);      // SPEC: no dropout, no activation after BN
#pragma unroll
                for (int e = 0; e < VE; ++e) { part[0][e] += g[e]; part[1][e] = fmaf(g[e], (z[u][e] - mu[e]) * is[e], part[1][e]); }
            }
        }
    }
    block_fold<2, VE>(part, active, prow * gm.cg + cgi, a.c, gm.rpi, lds, ws_bn + (size_t)blockIdx.x * 2 * a.c);
    block_fold<2 * KK, VE>(hpart, active, prow * gm.cg + cgi, a.c, gm.rpi, lds, ws_hd + (size_t)blockIdx.x * 2 * KK * a.c);
}

template <typename T, int KK, int SPEC = 0>
__global__ __launch_bounds__(256) void bn_bwd_apply_head_kernel(BnBwdArgs a, HeadFuse hd0, RedGeom gm, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    const HeadFuse hd = head_lazy_begin(hd0);
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % gm.cg, prow = tid / gm.cg;
    const bool active = prow < gm.rpi;
    const long long r0 = blockIdx.x * gm.chunk, r1 = (r0 + gm.chunk < a.rows) ? r0 + gm.chunk : a.rows;
    float part[1][VE], c1[VE], c2[VE], c3[VE], wr[VE][KK];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[0][e] = 0.f;
        const int ch = cgi * VE + e;
        c1[e] = (a.has_bn && active) ? a.coef[ch] : 1.f;
        c2[e] = (a.has_bn && active) ? a.coef[a.c + ch] : 0.f;
        c3[e] = (a.has_bn && active) ? a.coef[2 * a.c + ch] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wr[e][kk] = (active && kk < hd.k) ? hd.w[ch * hd.k + kk] : 0.f;
    }
    if (active) {
        for (long long r = r0 + prow; r < r1; r += 2 * gm.rpi) {
            float z[2][VE], g[2][VE], d[KK]; size_t e0[2]; bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long rr = r + u * gm.rpi;
                ok[u] = rr < r1;
                e0[u] = (size_t)rr * a.c + cgi * VE;
                if (ok[u]) { Vec<T>::load_nt(a.z + e0[u] * sizeof(T), z[u]); head_grad_vec<T, VE, KK>(hd, rr, wr, d, g[u]); }      // last use of z
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!ok[u]) continue;
                if constexpr (!SPEC) xform_g<T, VE>(a, e0[u], cgi * VE, 0u, z[u], g[u]);
                float dd[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    float t = fmaf(c1[e], g[u][e], fmaf(c2[e], z[u][e], c3[e]));
                    if (SPEC || !a.act_after_bn) t *= act_bwd(z[u][e], SPEC ? RVIP_ACT_RELU : a.act);
                    dd[e] = t;
                    part[0][e] += Vec<T>::round(t);
                }
                Vec<T>::store(a.dz + e0[u] * sizeof(T), dd);
            }
        }
    }
    block_fold<1, VE>(part, active, prow * gm.cg + cgi, a.c, gm.rpi, lds, ws + (size_t)blockIdx.x * a.c);
}

// first layer (Cin = 1) weight gradient: dw[t][co] = sum_p x[p + off(t)] * dy[p][co]
template <typename T>
__global__ __launch_bounds__(256) void c1_wgrad_kernel(const T* __restrict__ x, const unsigned char* __restrict__ dy,
                                                       int n, int h, int w, int cout, RedGeom gm, float* __restrict__ ws) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cgi = tid % gm.cg, prow = tid / gm.cg;
    const bool active = prow < gm.rpi;
    const long long rows = (long long)n * h * w;
    const long long r0 = blockIdx.x * gm.chunk, r1 = (r0 + gm.chunk < rows) ? r0 + gm.chunk : rows;
    float part[9][VE];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VE; ++e) part[t][e] = 0.f;
    if (active) {
        for (long long r = r0 + prow; r < r1; r += gm.rpi) {
            int px, py; long long img;
            split_xy(r, w, h, px, py, img);
            float g[VE];
            Vec<T>::load(dy + ((size_t)r * cout + cgi * VE) * sizeof(T), g);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                float xv = 0.f;
                if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
                    if constexpr (sizeof(T) == 4) xv = x[(img * h + yy) * w + xx];
                    else xv = Vec<T>::dec(x[(img * h + yy) * w + xx].bits);
                }
#pragma unroll
                for (int e = 0; e < VE; ++e) part[t][e] = fmaf(xv, g[e], part[t][e]);
            }
        }
    }
    block_fold<9, VE>(part, active, prow * gm.cg + cgi, cout, gm.rpi, lds, ws + (size_t)blockIdx.x * 9 * cout);
}

// tiled form of the same reduction: halo of the single input channel in LDS, thread = (pixel slot, channel vector)
template <typename T>
__global__ __launch_bounds__(256) void c1_wgrad_tiled(const T* __restrict__ x, const unsigned char* __restrict__ dy,
                                                      int n, int h, int w, int cout, int tiles_x, int tiles_y, float* __restrict__ ws,
                                                      int depth, int dshift, int xstride, int xoff) {   // x element = pixel * xstride + xoff
    constexpr int VE = Vec<T>::VE;
    __shared__ float xs[10 * 34];
    __shared__ float lds[256 * VE];
    const int tid = threadIdx.x, cg = cout / VE, cv = tid % cg, ps = tid / cg, pps = 256 / cg;
    float part[9][VE];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VE; ++e) part[t][e] = 0.f;
    const int ntiles = n * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bx = tile;
        const int tx0 = (bx % tiles_x) * 32; bx /= tiles_x;
        const int ty0 = (bx % tiles_y) * 8;
        const long long img = bx / tiles_y;
        const long long ximg = img + dshift;                   // Conv3D depth tap: the slice dshift away, zeros outside the volume
        const bool dok = (unsigned)((int)(img % depth) + dshift) < (unsigned)depth;
        __syncthreads();
        for (int i = tid; i < 340; i += 256) {
            const int gy = ty0 - 1 + i / 34, gx = tx0 - 1 + i % 34;
            float xv = 0.f;
            if (dok && (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w) {
                if constexpr (sizeof(T) == 4) xv = x[((ximg * h + gy) * w + gx) * xstride + xoff];
                else xv = Vec<T>::dec(x[((ximg * h + gy) * w + gx) * xstride + xoff].bits);
            }
            xs[i] = xv;
        }
        __syncthreads();
        for (int p = ps; p < 256; p += pps) {
            const int py = p >> 5, px = p & 31;
            const int gy = ty0 + py, gx = tx0 + px;
            if (gy >= h || gx >= w) continue;
            float g[VE];
            Vec<T>::load(dy + ((((size_t)img * h + gy) * w + gx) * cout + cv * VE) * sizeof(T), g);
            // two channels per instruction (v_pk_fma_f32; the pass is VALU-bound); every sum keeps its order
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float xv = xs[(py + t / 3) * 34 + px + t % 3];
                const rvip_f32x2 x2 = {xv, xv};
#pragma unroll
                for (int e2 = 0; e2 < VE / 2; ++e2) {
                    const rvip_f32x2 r2 = __builtin_elementwise_fma(x2, rvip_f32x2{g[2 * e2], g[2 * e2 + 1]}, rvip_f32x2{part[t][2 * e2], part[t][2 * e2 + 1]});
                    part[t][2 * e2] = r2.x; part[t][2 * e2 + 1] = r2.y;
                }
            }
        }
    }
    block_fold<9, VE>(part, true, ps * cg + cv, cout, pps, lds, ws + (size_t)blockIdx.x * 9 * cout);
}

// The same reduction on the matrix cores (16-bit types, Cout = 32, 2-D): dw[tap][co] = sum_p x[p + tap] * dy[p][co] is the weight
// gradient of a 1 x 1 convolution whose input is im2col(x) -- the nine shifted copies of the single channel as a 32-"channel" pixel
// row (channels 9..31 zero).  Per 8 x 32 tile the workgroup builds that [256][64 B] tile in LDS beside the [256][64 B] dy tile and
// contracts them over the pixels exactly as wgrad3x3_kernel does for one tap (ds_read_tr16_b64 transposes the pixels into the K
// run, v_mfma_f32_32x32x16): the products of two 16-bit values are exact in fp32, so against c1_wgrad_tiled only the order of the
// fp32 additions differs.  c1_wgrad_tiled needed ~125 lane-operations per 16 bytes of dy (VALU-bound, 3.1 TB/s); this form moves
// the multiply-adds to four MFMAs per wave and tile.  Same rows as c1_wgrad_tiled: ws[blockIdx.x][9][32].
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_pw;
__device__ __forceinline__ s16x4 tr_read_pw(const unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_pw*)(p)); }

template <typename T>
__global__ __launch_bounds__(256) void c1_wgrad_mfma(const T* __restrict__ x, const unsigned char* __restrict__ dy,
                                                     int n, int h, int w, int tiles_x, int tiles_y, float* __restrict__ ws) {
    static_assert(sizeof(T) == 2, "16-bit storage types");
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * 256 * 64];       // im2col(x) tile, dy tile: [256 pixels][64 B]
    __shared__ unsigned short xs[10 * 34 + 4];
    unsigned char* lx = sm;
    unsigned char* lg = sm + 256 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, hf = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // taps 16..31 of every im2col row stay zero for good; taps 9..15 are rewritten as zeros with every tile's row
    *reinterpret_cast<uint4*>(lx + tid * 64 + 32) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(lx + tid * 64 + 48) = make_uint4(0, 0, 0, 0);
    const int py = tid >> 5, px = tid & 31;                                          // this thread's pixel of the 8 x 32 tile
    const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, grp = lane >> 4;
    const int kk = 8 * (grp >> 1) + q;
    const int cb = (16 * (grp & 1) + 4 * p4) * 2;
    const int ntiles = n * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bx = tile;
        const int tx0 = (bx % tiles_x) * 32; bx /= tiles_x;
        const int ty0 = (bx % tiles_y) * 8;
        const long long img = bx / tiles_y;
        __syncthreads();                                                             // the previous tile has been contracted
        for (int i = tid; i < 340; i += 256) {
            const int gy = ty0 - 1 + i / 34, gx = tx0 - 1 + i % 34;
            xs[i] = ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w) ? x[(img * h + gy) * w + gx].bits : (unsigned short)0;
        }
        {
            const int gy = ty0 + py, gx = tx0 + px;
            uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
            if (gy < h && gx < w) {
                const uint4* src = reinterpret_cast<const uint4*>(dy + ((size_t)(img * h + gy) * w + gx) * 64);
                r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
            }
            uint4* dst = reinterpret_cast<uint4*>(lg + tid * 64);
            dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
        }
        __syncthreads();
        {
            unsigned v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = xs[(py + t / 3) * 34 + px + t % 3];
            uint4* dst = reinterpret_cast<uint4*>(lx + tid * 64);
            dst[0] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
            dst[1] = make_uint4(v[8], 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int P0 = wv * 64 + s4 * 16 + kk;
            const s16x4 g0 = tr_read_pw(lg + P0 * 64 + cb), g1 = tr_read_pw(lg + (P0 + 4) * 64 + cb);
            const s16x4 x0 = tr_read_pw(lx + P0 * 64 + cb), x1 = tr_read_pw(lx + (P0 + 4) * 64 + cb);
            const uint4 fb = __builtin_bit_cast(uint4, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
            const uint4 fa = __builtin_bit_cast(uint4, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
            acc = mfma16<T>(fa, fb, acc);
        }
    }
    // the four waves (quarters of every tile's pixels) in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(sm);                                      // [4][9][32]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tap = (r & 3) + 8 * (r >> 2) + 4 * hf;
        if (tap < 9) red[(wv * 9 + tap) * 32 + j] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < 9 * 32; e += 256)
        ws[(size_t)blockIdx.x * 9 * 32 + e] = ((red[e] + red[9 * 32 + e]) + red[2 * 9 * 32 + e]) + red[3 * 9 * 32 + e];
}

struct PostC1Wgrad {
    float* dw; int cout; int tstride;                          // tap stride of dw: cout for one input channel, cin * cout for HWIO with cin > 1
    __device__ void run_k(int ch, int k, double t) const { dw[k * tstride + ch] = (float)t; }
};

// landmark = flat argmax over H*W per (slice, class), first maximum wins; optional > thr mask
__global__ __launch_bounds__(256) void landmarks_kernel(const float* __restrict__ pred, long long* __restrict__ idx_out,
                                                        uint8_t* __restrict__ mask_out, int hw, int k, float thr) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int n = blockIdx.x / k, kk = blockIdx.x % k, tid = threadIdx.x;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < hw; i += 256) {
        const float v = pred[((size_t)n * hw + i) * k + kk];
        if (mask_out) mask_out[((size_t)n * hw + i) * k + kk] = v > thr ? 1 : 0;
        if (v > bv) { bv = v; bi = i; }                        // strided ascending scan: first max per thread
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v2 = sv[tid + o]; const int i2 = si[tid + o];
            if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) { sv[tid] = v2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) idx_out[blockIdx.x] = si[0] == 0x7fffffff ? 0 : si[0];
}

// ------------------------------------------------------------------------------------------------
// Keras Adam, state tick, dtype conversion
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ th, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long count, float b1, float b2, float eps,
                                                   float gscale, const uint32_t* __restrict__ state) {
    const float t = (float)(state[RVIP_STATE_STEP] + 1u);
    const float lr = __builtin_bit_cast(float, state[RVIP_STATE_LR]);
    const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        th[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long count, float scale) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < count; i += (long long)gridDim.x * 256) x[i] *= scale;
}
__global__ void state_tick_kernel(uint32_t* state) { if (threadIdx.x == 0 && blockIdx.x == 0) state[RVIP_STATE_STEP] += 1u; }

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ s, D* __restrict__ d, long long count) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        float v;
        if constexpr (sizeof(S) == 4) v = s[i]; else v = Vec<S>::dec(s[i].bits);
        if constexpr (sizeof(D) == 4) d[i] = v; else d[i].bits = Vec<D>::enc(v);
    }
}

}  // namespace rvip

using namespace rvip;


extern "C" size_t rvip_reduce_workspace(long long rows, int width) {
    (void)rows;
    return (size_t)1024 * (size_t)(width > 16 ? width : 16) * sizeof(float);
}

extern "C" int rvip_bn_train_stats(const void* z, long long rows, int c, int dtype, const float* gamma, const float* beta,
                                   float* moving_mean, float* moving_var, float momentum, float eps, int unbiased_moving,
                                   float* mean, float* invstd, float* scale, float* shift,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!z || !mean || !invstd || !scale || !shift || !workspace || !RVIP_DT_OK(dtype)) return RVIP_EINVAL;
    RedGeom g;
    if (!red_geom(rows, c, RVIP_VE(dtype), g)) return RVIP_EINVAL;
    if (workspace_bytes < (size_t)g.nblk * 2 * c * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(g.nblk), dim3(256), 0, s, (const unsigned char*)z, rows, c, g, ws);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(bn_stats_kernel<f16_t>, dim3(g.nblk), dim3(256), 0, s, (const unsigned char*)z, rows, c, g, ws);
    else hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(g.nblk), dim3(256), 0, s, (const unsigned char*)z, rows, c, g, ws);
    int rc = check_launch();
    if (rc) return rc;
    PostBnStats p{gamma, beta, moving_mean, moving_var, mean, invstd, scale, shift, (double)rows, momentum, eps, unbiased_moving};
    return launch_fold<2, PostBnStats, 16>(ws, g.nblk, c, p, s);
}

// Stage 2 alone: fold `rows` partial rows [rows][2][c] (sum, sum of squares; e.g. written by rvip_conv3x3_fwd_stats)
// into mean / invstd / scale / shift and the moving statistics.  `count` = N*H*W elements per channel.
extern "C" int rvip_bn_stats_finalize(const float* partial, int rows, long long count, int c, const float* gamma, const float* beta,
                                      float* moving_mean, float* moving_var, float momentum, float eps, int unbiased_moving,
                                      float* mean, float* invstd, float* scale, float* shift, void* stream) {
    (void)hipGetLastError();
    if (!partial || rows <= 0 || count <= 0 || c <= 0 || !mean || !invstd || !scale || !shift) return RVIP_EINVAL;
    PostBnStats p{gamma, beta, moving_mean, moving_var, mean, invstd, scale, shift, (double)count, momentum, eps, unbiased_moving};
    return launch_fold<2, PostBnStats, 16>(partial, rows, c, p, (hipStream_t)stream);
}

extern "C" int rvip_bn_infer_coeffs(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, int c,
                                    float* scale, float* shift, void* stream) {
    (void)hipGetLastError();
    if (!mm || !mv || !scale || !shift || c <= 0) return RVIP_EINVAL;
    hipLaunchKernelGGL(bn_infer_coeffs_kernel, dim3((unsigned)cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mm, mv, eps, c, scale, shift);
    return check_launch();
}

extern "C" int rvip_bn_apply(const rvip_apply_desc* d, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->z || !d->y || !RVIP_DT_OK(d->dtype)) return RVIP_EINVAL;
    const int ve = RVIP_VE(d->dtype);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->c % ve) return RVIP_EINVAL;
    if (d->pooled && ((d->h | d->w) & 1)) return RVIP_EINVAL;
    if (d->drop_rate < 0.f || d->drop_rate >= 1.f) return RVIP_EINVAL;
    const int drop = d->drop_rate > 0.f;
    if (drop && !d->mask && !d->state) return RVIP_EINVAL;
    ApplyArgs a;
    a.z = (const unsigned char*)d->z; a.y = (unsigned char*)d->y; a.pooled = (unsigned char*)d->pooled;
    a.scale = d->scale; a.shift = d->shift; a.act = d->act;
    a.inv_keep = drop ? 1.f / (1.f - d->drop_rate) : 1.f; a.thr = dropout_thr(d->drop_rate);
    a.mask = d->mask; a.state = d->state; a.layer_id = d->layer_id; a.drop = drop;
    a.n = d->n; a.h = d->h; a.w = d->w; a.c = d->c;
    a.argmax = nullptr;
    a.keep_bits = nullptr;
    a.rev = rev_flags() & 1;
    if (d->keep_bits) {
        if (d->pooled || !drop || d->c % 8 || ((d->c % 32) && (32 % d->c)) || ((uintptr_t)d->keep_bits & 3)) return RVIP_EINVAL;      // whole 32-channel blocks, or one partial block of 8 / 16
        a.keep_bits = d->keep_bits;
    }
    const int cg = d->c / ve;
    if (cg > 256) return RVIP_EINVAL;
    const int rpi = 256 / cg;
    hipStream_t s = (hipStream_t)stream;
    const long long units = d->pooled ? (long long)d->n * (d->h / 2) * (d->w / 2) : (long long)d->n * d->h * d->w;
    const long long ngroups = cdiv(units, rpi);
    const int unr = d->pooled ? 2 : RVIP_APPLY_UNR;
    long long nb = cdiv(ngroups, unr);
    static const long long nb_cap = [] { const char* e = getenv("RVIP_APPLY_BLOCKS"); return e ? atoll(e) : 1024LL; }();     // one resident round of workgroups: 7-10 % faster than 4096 (tools/probe_apply.py)
    if (nb > nb_cap) nb = nb_cap;
    dim3 grid((unsigned)nb);
    static const bool pool2 = [] { const char* e = getenv("RVIP_POOL_SPLIT"); return !(e && e[0] == '0'); }();
    // specialised instantiations of the reference's default graph (conv + ReLU -> BN: no activation in this pass): 1 = no dropout, 2 = stream
    static const bool spec = [] { const char* e = getenv("RVIP_SPECIALISE"); return !(e && e[0] == '0'); }();
    const int fast = (spec && d->act == RVIP_ACT_NONE) ? (!drop ? 1 : (!d->mask ? 2 : 0)) : 0;
    if (d->argmax && !(d->pooled && pool2 && cg <= 32 && (cg & (cg - 1)) == 0)) return RVIP_EUNSUPPORTED;   // rvip_bn_apply_argmax_ok
    if (d->pooled && pool2 && cg <= 32 && (cg & (cg - 1)) == 0) {        // column-split form: both lanes of a window in one wave
        a.argmax = d->argmax;
        const int wpi = 128 / cg;
        const long long ng2 = cdiv(units, wpi);
        long long nb2 = cdiv(ng2, RVIP_POOL2_UNR);
        if (nb2 > nb_cap) nb2 = nb_cap;
        by_dtype(d->dtype, [&](auto t) {
            using T = decltype(t);
            if (fast == 1) hipLaunchKernelGGL((bn_apply_pool2_kernel<T, RVIP_ACT_NONE, 0>), dim3((unsigned)nb2), dim3(256), 0, s, a, cg, wpi, ng2);
            else if (fast == 2) hipLaunchKernelGGL((bn_apply_pool2_kernel<T, RVIP_ACT_NONE, 1>), dim3((unsigned)nb2), dim3(256), 0, s, a, cg, wpi, ng2);
            else hipLaunchKernelGGL((bn_apply_pool2_kernel<T>), dim3((unsigned)nb2), dim3(256), 0, s, a, cg, wpi, ng2);
            return 0;
        });
    } else if (d->pooled) {
        by_dtype(d->dtype, [&](auto t) {
            hipLaunchKernelGGL((bn_apply_kernel<decltype(t), true>), grid, dim3(256), 0, s, a, cg, rpi, ngroups);
            return 0;
        });
    } else {
        by_dtype(d->dtype, [&](auto t) {
            using T = decltype(t);
            if (fast == 1) hipLaunchKernelGGL((bn_apply_kernel<T, false, RVIP_ACT_NONE, 0>), grid, dim3(256), 0, s, a, cg, rpi, ngroups);
            else if (fast == 2) hipLaunchKernelGGL((bn_apply_kernel<T, false, RVIP_ACT_NONE, 1>), grid, dim3(256), 0, s, a, cg, rpi, ngroups);
            else hipLaunchKernelGGL((bn_apply_kernel<T, false>), grid, dim3(256), 0, s, a, cg, rpi, ngroups);
            return 0;
        });
    }
    return check_launch();
}

static int fill_bnbwd(const rvip_bnbwd_desc* d, BnBwdArgs& a, RedGeom& g) {
    if (!d || (!d->dy && !d->dpooled) || !d->z || !RVIP_DT_OK(d->dtype)) return RVIP_EINVAL;
    if (d->dpooled && (!d->argmax || d->h <= 0 || d->w <= 0 || ((d->h | d->w) & 1) || d->rows % ((long long)d->h * d->w))) return RVIP_EINVAL;
    a.dp = (const unsigned char*)d->dpooled; a.argmax = d->argmax; a.h = d->h; a.w = d->w;
    a.lw = a.lh = -1;
    if (d->dpooled && !(d->w & (d->w - 1)) && !(d->h & (d->h - 1))) { a.lw = __builtin_ctz(d->w); a.lh = __builtin_ctz(d->h); }
    if (!red_geom(d->rows, d->c, RVIP_VE(d->dtype), g)) return RVIP_EINVAL;
    if (d->drop_rate < 0.f || d->drop_rate >= 1.f) return RVIP_EINVAL;
    const int drop = d->drop_rate > 0.f;
    if (drop && !d->mask && !d->state) return RVIP_EINVAL;
    if (!d->workspace) return RVIP_EINVAL;
    a.dy = (const unsigned char*)d->dy; a.z = (const unsigned char*)d->z; a.dz = (unsigned char*)d->dz;
    a.mean = d->mean; a.invstd = d->invstd; a.scale = d->scale; a.shift = d->shift; a.coef = d->coef;
    a.act = d->act; a.act_after_bn = d->act_after_bn; a.has_bn = d->gamma != nullptr;
    a.inv_keep = drop ? 1.f / (1.f - d->drop_rate) : 1.f; a.thr = dropout_thr(d->drop_rate);
    a.mask = d->mask; a.state = d->state; a.layer_id = d->layer_id; a.drop = drop;
    a.rows = d->rows; a.c = d->c;
    return RVIP_OK;
}

// specialised instantiations of the BN-backward passes for the reference's default graph (conv + ReLU -> BN [-> dropout]):
// 1 = no dropout, 2 = counter-stream dropout, 0 = the generic kernel
static int bnbwd_fast(const BnBwdArgs& a) {
    static const bool spec = [] { const char* e = getenv("RVIP_SPECIALISE"); return !(e && e[0] == '0'); }();
    if (!spec || a.act != RVIP_ACT_RELU || a.act_after_bn) return 0;
    return !a.drop ? 1 : (!a.mask ? 2 : 0);
}

// 1 if rvip_bn_apply writes the window argmax for this pooled shape (the column-split kernel runs): the caller may then drop
// rvip_maxpool2x2_bwd and hand dpooled / argmax to the two BN-backward passes
extern "C" int rvip_bn_apply_argmax_ok(int c, int dtype) {
    if (!RVIP_DT_OK(dtype) || c <= 0 || c % RVIP_VE(dtype)) return 0;
    static const bool pool2 = [] { const char* e = getenv("RVIP_POOL_SPLIT"); return !(e && e[0] == '0'); }();
    const int cg = c / RVIP_VE(dtype);
    return (pool2 && cg <= 32 && (cg & (cg - 1)) == 0) ? 1 : 0;
}

extern "C" int rvip_bn_bwd_reduce(const rvip_bnbwd_desc* d, void* stream) {
    (void)hipGetLastError();
    BnBwdArgs a; RedGeom g;
    int rc = fill_bnbwd(d, a, g);
    if (rc) return rc;
    if (!d->gamma || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->coef) return RVIP_EINVAL;
    if (d->workspace_bytes < (size_t)g.nblk * 2 * d->c * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)d->workspace;
    const int fast = bnbwd_fast(a);
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if (fast == 1 && a.dp) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, RVIP_ACT_RELU, 0, 0, 1>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        else if (fast == 1) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, RVIP_ACT_RELU, 0, 0, 0>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        else if (fast == 2 && !a.dp) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, RVIP_ACT_RELU, 1, 0, 0>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        return 0;
    });
    rc = check_launch();
    if (rc) return rc;
    PostBnBwd p{d->gamma, d->mean, d->invstd, d->dgamma, d->dbeta, d->coef, (double)d->rows, d->c};
    return launch_fold<2, PostBnBwd>(ws, g.nblk, d->c, p, s);
}

// ---- stage 1 of the BatchNormalization backward from the consumers' by-products (include/rvip_hip.h: rvip_bn_bwd_coef) ----
namespace rvip {
struct CoefSrc { const void* rows; int nrows, stride, offset; };      // t1: float rows, t2: double rows
struct CoefArgs {
    CoefSrc t1[2], t2[2];
    const float* gamma; const float* beta; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; float* coef; int* flags;
    double n; int c; float min_gamma, max_beta_ratio;
    BnBwdArgs fb;                                  // the stage's classic descriptor: the exact route of an ill-conditioned block
};
template <typename A>
__device__ __forceinline__ void coef_store(const A& a, int ch, double t1, double tx) {
    a.dbeta[ch] = (float)t1;
    a.dgamma[ch] = (float)tx;
    const float gm = a.gamma[ch], is = a.invstd[ch], mu = a.mean[ch];
    const float c1 = gm * is;
    const float c2 = -gm * is * is * (float)(tx / a.n);
    const float c3 = -gm * is * (float)(t1 / a.n) - c2 * mu;
    a.coef[ch] = c1; a.coef[a.c + ch] = c2; a.coef[2 * a.c + ch] = c3;
}
// 1024 threads = 32 channels x 32 row groups; double accumulation in a fixed order (sources in order, rows strided by group).
// A workgroup whose 32 channels hold an ill-conditioned one (|gamma| tiny, or |beta| >> |gamma|: the division below would amplify
// the rounding of the sums) recomputes ITS channels the classic way -- one workgroup streaming (g, z) of 32 channels, slow and
// exact; the condition does not arise in training from the Keras initialisation (gamma = 1, beta = 0), so the launch list needs
// no second, guarded kernel.
template <typename T>
__global__ __launch_bounds__(1024) void bn_bwd_coef_kernel(CoefArgs a) {
    constexpr int VE = Vec<T>::VE, NV = 32 / VE, NS = 1024 / NV;
    __shared__ double sh[2][32][32];
    __shared__ float part_lds[NS][32];
    __shared__ int sbad;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ch = blockIdx.x * 32 + c;
    double s[2] = {0.0, 0.0};
    if (ch < a.c) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const CoefSrc src = k == 0 ? a.t1[q] : a.t2[q];
                if (!src.rows) continue;
                if (k == 0) {
                    const float* base = reinterpret_cast<const float*>(src.rows) + src.offset + ch;
                    int b = g;
                    for (; b + 96 < src.nrows; b += 128) {
                        const float v0 = base[(size_t)b * src.stride], v1 = base[(size_t)(b + 32) * src.stride];
                        const float v2 = base[(size_t)(b + 64) * src.stride], v3 = base[(size_t)(b + 96) * src.stride];
                        s[k] += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
                    }
                    for (; b < src.nrows; b += 32) s[k] += (double)base[(size_t)b * src.stride];
                } else {
                    const double* base = reinterpret_cast<const double*>(src.rows) + src.offset + ch;
                    for (int b = g; b < src.nrows; b += 32) s[k] += base[(size_t)b * src.stride];
                }
            }
        }
    }
    sh[0][g][c] = s[0]; sh[1][g][c] = s[1];
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
    float gm = 1.f, bt = 0.f;
    if (threadIdx.x < 64) {                                  // wave 0: its first 32 lanes own the block's channels (g == 0)
        int isbad = 0;
        if (g == 0 && ch < a.c) {
#pragma unroll
            for (int gg = 0; gg < 32; ++gg) { t1 += sh[0][gg][c]; t2 += sh[1][gg][c]; }
            gm = a.gamma[ch]; bt = a.beta ? a.beta[ch] : 0.f;
            isbad = !(fabsf(gm) >= a.min_gamma && fabsf(bt) <= a.max_beta_ratio * fabsf(gm));
        }
        const int any = __ballot(isbad) != 0ull;               // one wave vote instead of 32 words through LDS, a serial OR and a barrier
        if (threadIdx.x == 0) { a.flags[blockIdx.x] = any; sbad = any; }
    }
    __syncthreads();
    if (!sbad) {
        // y = gamma * xhat + beta  =>  sum g*xhat = (sum g*y - beta * sum g) / gamma
        if (g == 0 && ch < a.c) coef_store(a, ch, t1, (t2 - (double)bt * t1) / (double)gm);
        return;
    }
    // ---------------- exact route for this block: sum g and sum g*xhat over every row, as rvip_bn_bwd_reduce computes them ----------------
    const BnBwdArgs& f = a.fb;
    const int cv = threadIdx.x % NV, slot = threadIdx.x / NV, cgi = blockIdx.x * NV + cv, cg = a.c / VE;
    const bool active = cgi < cg;
    const uint32_t key = (f.drop && !f.mask) ? dropout_key(f.state[RVIP_STATE_SEED], f.state[RVIP_STATE_STEP], (uint32_t)f.layer_id) : 0u;
    float part[2][VE], mu[VE], is[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[0][e] = part[1][e] = 0.f;
        mu[e] = active ? a.mean[cgi * VE + e] : 0.f;
        is[e] = active ? a.invstd[cgi * VE + e] : 0.f;
    }
    if (active) {
        for (long long rr = slot; rr < f.rows; rr += NS) {
            float z[VE], gv[VE];
            GyRaw<VE> raw;
            const size_t e0 = (size_t)rr * a.c + cgi * VE;
            Vec<T>::load(f.z + e0 * sizeof(T), z);
            gy_issue<T, VE, -1, false>(f, rr, cgi, cg, e0, raw);
            gy_finish<T, VE, -1>(f, raw, gv);
            xform_g<T, VE>(f, e0, cgi * VE, key, z, gv);
#pragma unroll
            for (int e = 0; e < VE; ++e) { part[0][e] += gv[e]; part[1][e] = fmaf(gv[e], (z[e] - mu[e]) * is[e], part[1][e]); }
        }
    }
    double tot[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VE; ++e) part_lds[slot][cv * VE + e] = part[k][e];
        __syncthreads();
        double acc = 0.0;
        for (int r = g; r < NS; r += 32) acc += (double)part_lds[r][c];
        sh[k][g][c] = acc;
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int gg = 0; gg < 32; ++gg) tot[k] += sh[k][gg][c];
        }
    }
    if (g == 0 && ch < a.c) coef_store(a, ch, tot[0], tot[1]);
}

// ---- the same for the last stage under the MSE head (include/rvip_hip.h: rvip_head_mse_coef): the rows come from the forward pass ----
struct HeadCoefArgs {
    const float* rows; int nrows;                  // [nrows][3][C]: S_0, S_1, (Q_0, Q_1, ...)
    const float* w; const float* dlogit; int k;
    float* head_dw; float* head_db;
    const float* sums; float* loss_out; float inv_count;
    const unsigned char* z; long long px;          // the exact route streams these
    const float* gamma; const float* beta; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; float* coef; int* flags;
    double n; int c; float min_gamma, max_beta_ratio;
    // MODE 2 (BCE-Dice): rows are [nrows][3 KK + 1][C] (three term sets, see bn_apply_head_kernel); the coefficients come from the folded sums
    float w_bce, w_dice, lg, dscale; const float* pred; const float* yt; float* dcoef;
};
// the three global coefficients of the BCE-Dice logit gradient  d = ca (p - t) + (cb t + cc) p (1 - p)  (rvip_head_grad's expression
// times dscale):  ca = w_bce / count,  cb = -w_dice lg 2 / den,  cc = w_dice lg (2 I + 1) / den^2,  den = sum t + sum p + 1
__device__ __forceinline__ void bcedice_coefs(const HeadCoefArgs& a, float& ca, float& cb, float& cc) {
    const float inter = a.sums[2], den = a.sums[3] + a.sums[4] + 1.f;
    ca = a.w_bce * a.inv_count * a.dscale;
    cb = -a.w_dice * a.lg * (2.f / den) * a.dscale;
    cc = a.w_dice * a.lg * ((2.f * inter + 1.f) / (den * den)) * a.dscale;
}
template <typename T, int MODE = 1>
__global__ __launch_bounds__(1024) void head_mse_coef_kernel(HeadCoefArgs a) {
    constexpr int VE = Vec<T>::VE, NV = 32 / VE, NS = 1024 / NV, KK = 2, NT = MODE == 2 ? 3 : 1, NC = NT * KK;
    __shared__ double sh[KK + 1][32][32];
    __shared__ float part_lds[NS][32];
    __shared__ int bad[32];
    __shared__ int sbad;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ch = blockIdx.x * 32 + c;
    const size_t rs = (size_t)(NC + 1) * a.c;
    float ca = 1.f, cb = 0.f, cc = 0.f;
    if constexpr (MODE == 2) bcedice_coefs(a, ca, cb, cc);
    double s[NC + 1];
#pragma unroll
    for (int q = 0; q <= NC; ++q) s[q] = 0.0;
    {
        // one latency chain on one CU: D rows (S of every (term, class), and Q for the threads that fold it) are requested together
        const bool mine = ch < a.c, qmine = c < NC;           // every workgroup folds the Q row itself (T1 of each channel needs all of it)
        const __amdgpu_buffer_rsrc_t rr = rows_rsrc(a.rows, (size_t)a.nrows * rs * 4);
        constexpr unsigned NONE = 0x80000000u;
        const unsigned vs = mine ? (unsigned)((size_t)g * rs + ch) * 4u : NONE, vq = qmine ? (unsigned)((size_t)g * rs + (size_t)NC * a.c + c) * 4u : NONE;
        constexpr int D = MODE == 2 ? 8 : 16;
        for (int b = 0; b < a.nrows; b += 32 * D) {
            float v[D][NC], qv[D];
#pragma unroll
            for (int u = 0; u < D; ++u) {
                // (a scalar offset past the end would make the range check's `records - soffset` wrap: such rows get offset 0 + NONE)
                const bool any = b + 32 * u < a.nrows;                             // wave-uniform
                const unsigned so = any ? (unsigned)((size_t)(b + 32 * u) * rs) * 4u : 0u;
#pragma unroll
                for (int kk = 0; kk < NC; ++kk) v[u][kk] = buf_f32(rr, any ? vs : NONE, any ? so + (unsigned)(kk * a.c) * 4u : 0u);
                qv[u] = buf_f32(rr, any ? vq : NONE, so);
            }
#pragma unroll
            for (int u = 0; u < D; u += 4) {
#pragma unroll
                for (int kk = 0; kk < NC; ++kk) s[kk] += ((double)v[u][kk] + (double)v[u + 1][kk]) + ((double)v[u + 2][kk] + (double)v[u + 3][kk]);
                s[NC] += ((double)qv[u] + (double)qv[u + 1]) + ((double)qv[u + 2] + (double)qv[u + 3]);
            }
        }
    }
    if constexpr (MODE == 2) {
        // the gradient is linear in the three terms: combine this thread's partial sums with the global coefficients, then fold as before
        // (column c of the Q row belongs to term c / KK)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) s[kk] = (double)ca * s[kk] + (double)cb * s[KK + kk] + (double)cc * s[2 * KK + kk];
        s[NC] *= (double)(c < KK ? ca : (c < 2 * KK ? cb : cc));
    }
#pragma unroll
    for (int q = 0; q < KK; ++q) sh[q][g][c] = s[q];
    sh[KK][g][c] = s[NC];
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
    float gm = 1.f, bt = 0.f;
    if (g == 0) {
        double S[KK], Q[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            S[kk] = Q[kk] = 0.0;
#pragma unroll
            for (int gg = 0; gg < 32; ++gg) {
                S[kk] += sh[kk][gg][c];
#pragma unroll
                for (int j = 0; j < NT; ++j) Q[kk] += sh[KK][gg][j * KK + kk];
            }
        }
        int isbad = 0;
        if (ch < a.c) {
            gm = a.gamma[ch]; bt = a.beta ? a.beta[ch] : 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk < a.k) {                                          // S = sum (y - beta) d:  t2 = gamma * sum g*xhat
                    const double w = (double)a.w[ch * a.k + kk];
                    t1 += w * Q[kk]; t2 += w * S[kk];
                    a.head_dw[ch * a.k + kk] = (float)(S[kk] + (double)bt * Q[kk]);
                }
            }
            isbad = !(fabsf(gm) >= a.min_gamma && fabsf(bt) <= a.max_beta_ratio * fabsf(gm));
        }
        if (blockIdx.x == 0 && c == 0) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                if (kk < a.k) a.head_db[kk] = (float)Q[kk];
            if constexpr (MODE == 2) {
                a.dcoef[0] = ca; a.dcoef[1] = cb; a.dcoef[2] = cc;          // what rvip_bn_bwd_apply_head_lazy rebuilds the gradient with
                if (a.loss_out) {
                    const float den = a.sums[3] + a.sums[4] + 1.f;
                    a.loss_out[0] = a.w_bce * a.sums[1] * a.inv_count - a.w_dice * a.lg * (2.f * a.sums[2] + 1.f) / den;      // rvip_head_grad's value
                }
            } else if (a.loss_out) a.loss_out[0] = a.sums[0] * a.inv_count;
        }
        bad[c] = isbad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int any = 0;
        for (int i = 0; i < 32; ++i) any |= bad[i];
        a.flags[blockIdx.x] = any;
        sbad = any;
    }
    __syncthreads();
    if (!sbad) {
        if (g == 0 && ch < a.c) coef_store(a, ch, t1, t2 / (double)gm);
        return;
    }
    // ---------------- exact route for this block: sum g and sum g*xhat over every pixel, as rvip_bn_bwd_reduce_head computes them ----------------
    const int cv = threadIdx.x % NV, slot = threadIdx.x / NV, cgi = blockIdx.x * NV + cv, cg = a.c / VE;
    const bool active = cgi < cg;
    HeadFuse hd{a.w, nullptr, a.dlogit, a.k};
    if constexpr (MODE == 2) { hd.dlogit = nullptr; hd.pred = a.pred; hd.yt = a.yt; hd.ca = ca; hd.cb = cb; hd.cc = cc; }
    float part[2][VE], mu[VE], is[VE], wr[VE][KK];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        part[0][e] = part[1][e] = 0.f;
        mu[e] = active ? a.mean[cgi * VE + e] : 0.f;
        is[e] = active ? a.invstd[cgi * VE + e] : 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wr[e][kk] = (active && kk < a.k) ? a.w[(cgi * VE + e) * a.k + kk] : 0.f;
    }
    if (active) {
        for (long long rr = slot; rr < a.px; rr += NS) {
            float z[VE], gv[VE], d[KK];
            Vec<T>::load(a.z + ((size_t)rr * a.c + cgi * VE) * sizeof(T), z);
            head_grad_vec<T, VE, KK>(hd, rr, wr, d, gv);
#pragma unroll
            for (int e = 0; e < VE; ++e) { part[0][e] += gv[e]; part[1][e] = fmaf(gv[e], (z[e] - mu[e]) * is[e], part[1][e]); }
        }
    }
    double tot[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VE; ++e) part_lds[slot][cv * VE + e] = part[k][e];
        __syncthreads();
        double acc = 0.0;
        for (int r = g; r < NS; r += 32) acc += (double)part_lds[r][c];
        sh[k][g][c] = acc;
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int gg = 0; gg < 32; ++gg) tot[k] += sh[k][gg][c];
        }
    }
    if (g == 0 && ch < a.c) coef_store(a, ch, tot[0], tot[1]);
}
}  // namespace rvip

extern "C" int rvip_bn_bwd_coef(const rvip_bncoef_desc* d, void* stream) {
    (void)hipGetLastError();
    if (!d || d->c <= 0 || d->count <= 0 || !d->gamma || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->coef || !d->flags) return RVIP_EINVAL;
    if (!d->t1[0].rows || !d->t2[0].rows || !(d->min_gamma > 0.f) || !(d->max_beta_ratio > 0.f) || !d->fallback) return RVIP_EINVAL;
    CoefArgs a;
    for (int q = 0; q < 2; ++q) {
        const rvip_bncoef_src* in[2] = {&d->t1[q], &d->t2[q]};
        CoefSrc* out[2] = {&a.t1[q], &a.t2[q]};
        for (int k = 0; k < 2; ++k) {
            if (in[k]->rows && (in[k]->nrows <= 0 || in[k]->offset < 0 || in[k]->stride < in[k]->offset + d->c)) return RVIP_EINVAL;
            *out[k] = CoefSrc{in[k]->rows, in[k]->nrows, in[k]->stride, in[k]->offset};
        }
    }
    RedGeom g;
    const int rc = fill_bnbwd(d->fallback, a.fb, g);
    if (rc) return rc;
    if (d->fallback->c != d->c || d->fallback->rows != d->count || d->fallback->act_after_bn || !d->fallback->gamma) return RVIP_EINVAL;
    a.gamma = d->gamma; a.beta = d->beta; a.mean = d->mean; a.invstd = d->invstd;
    a.dgamma = d->dgamma; a.dbeta = d->dbeta; a.coef = d->coef; a.flags = d->flags;
    a.n = (double)d->count; a.c = d->c; a.min_gamma = d->min_gamma; a.max_beta_ratio = d->max_beta_ratio;
    by_dtype(d->fallback->dtype, [&](auto t) {
        hipLaunchKernelGGL(bn_bwd_coef_kernel<decltype(t)>, dim3((unsigned)cdiv(d->c, 32)), dim3(1024), 0, (hipStream_t)stream, a);
        return 0;
    });
    return check_launch();
}

extern "C" int rvip_bn_bwd_apply(const rvip_bnbwd_desc* d, void* stream) {
    (void)hipGetLastError();
    BnBwdArgs a; RedGeom g;
    int rc = fill_bnbwd(d, a, g);
    if (rc) return rc;
    if (!d->dz || (!d->dbias && !d->bias_rows)) return RVIP_EINVAL;
    if (d->gamma && !d->coef) return RVIP_EINVAL;
    const bool defer = d->bias_rows != nullptr;      // leave the [nblk][C] partial rows to a later rvip_fold_rows_batch
    if ((defer ? d->bias_rows_bytes : d->workspace_bytes) < (size_t)g.nblk * d->c * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = defer ? d->bias_rows : (float*)d->workspace;
    const int fast = bnbwd_fast(a);
    a.rev = (rev_flags() >> 1) & 1;
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if (fast == 1 && a.dp) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, RVIP_ACT_RELU, 0, 0, 1>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        else if (fast == 1) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, RVIP_ACT_RELU, 0, 0, 0>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        else if (fast == 2 && !a.dp) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, RVIP_ACT_RELU, 1, 0, 0>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g.nblk), dim3(256), 0, s, a, g, ws);
        return 0;
    });
    rc = check_launch();
    if (rc || defer) return rc;
    PostSum p{d->dbias};
    return launch_fold<1, PostSum>(ws, g.nblk, d->c, p, s);
}

// bn_bwd_apply_head_kernel: 127 VGPRs for <= 2 classes (4 waves per SIMD: 1024 resident workgroups), 143 beyond (3 waves: 768)
#define RVIP_APPLY_HEAD_CAP(k) ((k) <= 2 ? 1024 : 768)
extern "C" int rvip_bn_bwd_apply_head_rows(long long rows, int c, int dtype, int k) {
    RedGeom g;
    if (!RVIP_DT_OK(dtype) || k <= 0 || !red_geom(rows, c, RVIP_VE(dtype), g, RVIP_APPLY_HEAD_CAP(k))) return 0;
    return g.nblk;
}
extern "C" int rvip_bn_bwd_rows(long long rows, int c, int dtype) {      // pass rows / 4 for a pool-fused descriptor
    RedGeom g;
    if (!RVIP_DT_OK(dtype) || !red_geom(rows, c, RVIP_VE(dtype), g)) return 0;
    return g.nblk;
}

extern "C" int rvip_fold_rows_batch(const void* table, int entries, long long max_width, int wide, void* stream) {
    (void)hipGetLastError();
    if (!table || entries <= 0 || max_width <= 0) return RVIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (wide) {
        if (max_width % 4) return RVIP_EINVAL;
        long long nb = cdiv(max_width / 4, 32);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(fold_batch_wide, dim3((unsigned)nb, (unsigned)entries), dim3(256), 0, s, (const FoldEntry*)table);
    } else {
        hipLaunchKernelGGL(fold_batch_narrow, dim3((unsigned)cdiv(max_width, 32), (unsigned)entries), dim3(1024), 0, s, (const FoldEntry*)table);
    }
    return check_launch();
}

extern "C" int rvip_maxpool2x2_bwd(const void* y, const void* dpooled, const void* add, void* dx, int n, int h, int w, int c,
                                   int dtype, void* stream) {
    (void)hipGetLastError();
    if (!y || !dpooled || !dx || !RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w <= 0 || ((h | w) & 1) || c <= 0 || c % RVIP_VE(dtype)) return RVIP_EINVAL;
    const long long total = (long long)n * (h / 2) * (w / 2) * (c / RVIP_VE(dtype));
    dim3 grid((unsigned)cdiv(total, 256));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const unsigned char*)y, (const unsigned char*)dpooled, (const unsigned char*)add, (unsigned char*)dx, n, h, w, c);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<f16_t>, grid, dim3(256), 0, s, (const unsigned char*)y, (const unsigned char*)dpooled, (const unsigned char*)add, (unsigned char*)dx, n, h, w, c);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, s, (const unsigned char*)y, (const unsigned char*)dpooled, (const unsigned char*)add, (unsigned char*)dx, n, h, w, c);
    return check_launch();
}

template <bool BWD>
static int launch_upsample(const void* src, void* dst, int n, int h, int w, int c, int dtype, void* stream) {
    if (!src || !dst || !RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % RVIP_VE(dtype)) return RVIP_EINVAL;
    const long long total = (long long)n * h * w * (c / RVIP_VE(dtype));
    dim3 grid((unsigned)cdiv(total, 256));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL((upsample_kernel<bf16_t, BWD>), grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, n, h, w, c);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL((upsample_kernel<f16_t, BWD>), grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, n, h, w, c);
    else hipLaunchKernelGGL((upsample_kernel<float, BWD>), grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, n, h, w, c);
    return check_launch();
}
// h, w are the LOW-resolution extents for both directions
extern "C" int rvip_upsample2x_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream) {
    (void)hipGetLastError();
    return launch_upsample<false>(x, y, n, h, w, c, dtype, stream);
}
extern "C" int rvip_upsample2x_bwd(const void* dy, void* dx, int n, int h, int w, int c, int dtype, void* stream) {
    (void)hipGetLastError();
    return launch_upsample<true>(dy, dx, n, h, w, c, dtype, stream);
}

extern "C" int rvip_subsample_odd(const void* src, void* dst, int n, int h, int w, int c, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!src || !dst || !RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % RVIP_VE(dtype)) return RVIP_EINVAL;
    const long long total = (long long)n * h * w * (c / RVIP_VE(dtype));
    dim3 grid((unsigned)cdiv(total, 256));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(subsample_odd_kernel<bf16_t>, grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, n, h, w, c);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(subsample_odd_kernel<f16_t>, grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, n, h, w, c);
    else hipLaunchKernelGGL(subsample_odd_kernel<float>, grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, n, h, w, c);
    return check_launch();
}

struct PostHeadSums {
    float* sums;
    __device__ void run(int ch, const double (&t)[1]) const { sums[ch] = (float)t[0]; }
};

extern "C" int rvip_head_fwd(const void* x, const float* w, const float* b, float* pred, const float* y_true, float* sums,
                             long long rows, int cin, int k, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !pred || !RVIP_DT_OK(dtype) || rows <= 0 || cin <= 0 || cin % RVIP_VE(dtype) || k <= 0 || k > RVIP_MAXK) return RVIP_EINVAL;
    if (y_true && (!sums || !workspace)) return RVIP_EINVAL;
    long long nb = cdiv(rows, 256 * 4);
    if (nb > 1024) nb = 1024;
    const long long chunk = cdiv(rows, nb);
    nb = cdiv(rows, chunk);
    if (y_true && workspace_bytes < (size_t)nb * 16 * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = y_true ? (float*)workspace : nullptr;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const unsigned char*)x, w, b, pred, y_true, rows, cin, k, chunk, ws);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(head_fwd_kernel<f16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const unsigned char*)x, w, b, pred, y_true, rows, cin, k, chunk, ws);
    else hipLaunchKernelGGL(head_fwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const unsigned char*)x, w, b, pred, y_true, rows, cin, k, chunk, ws);
    int rc = check_launch();
    if (rc || !y_true) return rc;
    PostHeadSums p{sums};
    return launch_fold<1, PostHeadSums>(ws, (int)nb, 16, p, s);
}

// Last stage + head, forward (see bn_apply_head_kernel).  d->y / d->pooled are ignored; no dropout on this stage.
extern "C" int rvip_bn_apply_head(const rvip_apply_desc* d, const float* head_w, const float* head_b, int k, float* pred,
                                  const float* y_true, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->z || !head_w || !pred || !RVIP_DT_OK(d->dtype) || k <= 0 || k > RVIP_MAXK) return RVIP_EINVAL;
    const int ve = RVIP_VE(d->dtype);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->c % ve || d->drop_rate > 0.f) return RVIP_EINVAL;
    const int cg = d->c / ve;
    if (cg > 64 || (cg & (cg - 1))) return RVIP_EUNSUPPORTED;
    if (y_true && (!sums || !workspace)) return RVIP_EINVAL;
    ApplyArgs a;
    a.z = (const unsigned char*)d->z; a.y = nullptr; a.pooled = nullptr; a.argmax = nullptr;
    a.scale = d->scale; a.shift = d->shift; a.act = d->act;
    a.inv_keep = 1.f; a.thr = 0; a.mask = nullptr; a.state = nullptr; a.layer_id = 0; a.drop = 0;
    a.n = d->n; a.h = d->h; a.w = d->w; a.c = d->c;
    const long long rows = (long long)d->n * d->h * d->w;
    long long nb = cdiv(rows, 256 * 4);
    if (nb > 1024) nb = 1024;
    const long long chunk = cdiv(rows, nb);
    nb = cdiv(rows, chunk);
    if (y_true && workspace_bytes < (size_t)nb * 16 * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = y_true ? (float*)workspace : nullptr;
    HeadFuse hd{head_w, head_b, nullptr, k};
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if (k == 2 && a.act == RVIP_ACT_NONE && a.c / Vec<T>::VE == 4) hipLaunchKernelGGL((bn_apply_head_kernel<T, 2, 1, 0, 1>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, y_true ? 1 : 0, ws, HeadMse{});
        else if (k <= 2 && a.act == RVIP_ACT_NONE) hipLaunchKernelGGL((bn_apply_head_kernel<T, 2, 1>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, y_true ? 1 : 0, ws, HeadMse{});
        else if (k <= 2) hipLaunchKernelGGL((bn_apply_head_kernel<T, 2>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, y_true ? 1 : 0, ws, HeadMse{});
        else hipLaunchKernelGGL((bn_apply_head_kernel<T, RVIP_MAXK>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, y_true ? 1 : 0, ws, HeadMse{});
        return 0;
    });
    int rc = check_launch();
    if (rc || !y_true) return rc;
    PostHeadSums p{sums};
    return launch_fold<1, PostHeadSums>(ws, (int)nb, 16, p, s);
}

// Last stage + head, backward stage 1 and 2 (d->dy is ignored: the incoming gradient is rebuilt from dlogit).
// Stage 1 also writes the head's weight / bias gradients.  workspace: rvip_reduce_workspace(rows, 16 * c) bytes.
extern "C" int rvip_bn_bwd_reduce_head(const rvip_bnbwd_desc* d, const float* head_w, const float* dlogit, int k,
                                       float* head_dw, float* head_db, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->z || !head_w || !dlogit || !head_dw || !head_db || !RVIP_DT_OK(d->dtype) || k <= 0 || k > RVIP_MAXK) return RVIP_EINVAL;
    if (!d->gamma || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->coef || !d->workspace || d->drop_rate > 0.f) return RVIP_EINVAL;
    RedGeom g;
    const int kcap = k <= 2 ? 2 : RVIP_MAXK;                        // class capacity of the instantiation (register budget)
    if (!red_geom(d->rows, d->c, RVIP_VE(d->dtype), g, kcap == 2 ? 768 : 512)) return RVIP_EINVAL;      // 3 / 2 waves per SIMD
    const size_t need = (size_t)g.nblk * (2 + 2 * RVIP_MAXK) * d->c * sizeof(float);
    if (d->workspace_bytes < need) return RVIP_EWORKSPACE;
    BnBwdArgs a;
    a.dy = nullptr; a.z = (const unsigned char*)d->z; a.dz = (unsigned char*)d->dz;
    a.dp = nullptr; a.argmax = nullptr; a.h = a.w = 0; a.lw = a.lh = -1;
    a.mean = d->mean; a.invstd = d->invstd; a.scale = d->scale; a.shift = d->shift; a.coef = d->coef;
    a.act = d->act; a.act_after_bn = d->act_after_bn; a.has_bn = 1;
    a.inv_keep = 1.f; a.thr = 0; a.mask = nullptr; a.state = nullptr; a.layer_id = 0; a.drop = 0;
    a.rows = d->rows; a.c = d->c;
    hipStream_t s = (hipStream_t)stream;
    float* ws_bn = (float*)d->workspace;
    float* ws_hd = ws_bn + (size_t)g.nblk * 2 * d->c;
    HeadFuse hd{head_w, nullptr, dlogit, k};
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if (kcap == 2 && a.act == RVIP_ACT_RELU && !a.act_after_bn) hipLaunchKernelGGL((bn_bwd_reduce_head_kernel<T, 2, 1>), dim3(g.nblk), dim3(256), 0, s, a, hd, g, ws_bn, ws_hd);
        else if (kcap == 2) hipLaunchKernelGGL((bn_bwd_reduce_head_kernel<T, 2>), dim3(g.nblk), dim3(256), 0, s, a, hd, g, ws_bn, ws_hd);
        else hipLaunchKernelGGL((bn_bwd_reduce_head_kernel<T, RVIP_MAXK>), dim3(g.nblk), dim3(256), 0, s, a, hd, g, ws_bn, ws_hd);
        return 0;
    });
    int rc = check_launch();
    if (rc) return rc;
    PostBnBwd p{d->gamma, d->mean, d->invstd, d->dgamma, d->dbeta, d->coef, (double)d->rows, d->c};
    rc = launch_fold<2, PostBnBwd>(ws_bn, g.nblk, d->c, p, s);
    if (rc) return rc;
    PostHeadBwd ph{head_dw, head_db, d->c, k, kcap};
    return launch_fold_k<PostHeadBwd>(ws_hd, g.nblk, d->c, 2 * kcap, ph, s);
}

static int bn_bwd_apply_head_impl(const rvip_bnbwd_desc* d, HeadFuse hd, void* stream) {
    const int k = hd.k;
    if (!d || !d->z || !d->dz || !hd.w || !RVIP_DT_OK(d->dtype) || k <= 0 || k > RVIP_MAXK) return RVIP_EINVAL;
    if ((!d->dbias && !d->bias_rows) || (d->gamma && !d->coef) || d->drop_rate > 0.f) return RVIP_EINVAL;
    RedGeom g;
    if (!red_geom(d->rows, d->c, RVIP_VE(d->dtype), g, RVIP_APPLY_HEAD_CAP(k))) return RVIP_EINVAL;
    const bool defer = d->bias_rows != nullptr;
    if ((defer ? d->bias_rows_bytes : d->workspace_bytes) < (size_t)g.nblk * d->c * sizeof(float) || (!defer && !d->workspace)) return RVIP_EWORKSPACE;
    BnBwdArgs a;
    a.dy = nullptr; a.z = (const unsigned char*)d->z; a.dz = (unsigned char*)d->dz;
    a.dp = nullptr; a.argmax = nullptr; a.h = a.w = 0; a.lw = a.lh = -1;
    a.mean = d->mean; a.invstd = d->invstd; a.scale = d->scale; a.shift = d->shift; a.coef = d->coef;
    a.act = d->act; a.act_after_bn = d->act_after_bn; a.has_bn = d->gamma != nullptr;
    a.inv_keep = 1.f; a.thr = 0; a.mask = nullptr; a.state = nullptr; a.layer_id = 0; a.drop = 0;
    a.rows = d->rows; a.c = d->c;
    hipStream_t s = (hipStream_t)stream;
    float* ws = defer ? d->bias_rows : (float*)d->workspace;
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if (k <= 2 && a.act == RVIP_ACT_RELU && !a.act_after_bn) hipLaunchKernelGGL((bn_bwd_apply_head_kernel<T, 2, 1>), dim3(g.nblk), dim3(256), 0, s, a, hd, g, ws);
        else if (k <= 2) hipLaunchKernelGGL((bn_bwd_apply_head_kernel<T, 2>), dim3(g.nblk), dim3(256), 0, s, a, hd, g, ws);
        else hipLaunchKernelGGL((bn_bwd_apply_head_kernel<T, RVIP_MAXK>), dim3(g.nblk), dim3(256), 0, s, a, hd, g, ws);
        return 0;
    });
    int rc = check_launch();
    if (rc || defer) return rc;
    PostSum p{d->dbias};
    return launch_fold<1, PostSum>(ws, g.nblk, d->c, p, s);
}

extern "C" int rvip_bn_bwd_apply_head(const rvip_bnbwd_desc* d, const float* head_w, const float* dlogit, int k, void* stream) {
    (void)hipGetLastError();
    if (!dlogit) return RVIP_EINVAL;
    return bn_bwd_apply_head_impl(d, HeadFuse{head_w, nullptr, dlogit, k}, stream);
}

// the same pass with the logit gradient rebuilt from the heat-map, the target and the three coefficients rvip_head_mse_coef (BCE-Dice
// form) left in dcoef: no dlogit tensor exists in that form of the step
extern "C" int rvip_bn_bwd_apply_head_lazy(const rvip_bnbwd_desc* d, const float* head_w, const float* pred, const float* y_true,
                                           const float* dcoef, int k, void* stream) {
    (void)hipGetLastError();
    if (!pred || !y_true || !dcoef) return RVIP_EINVAL;
    if (k > 2) return RVIP_EUNSUPPORTED;
    HeadFuse hd{head_w, nullptr, nullptr, k};
    hd.pred = pred; hd.yt = y_true; hd.dcoef = dcoef;
    return bn_bwd_apply_head_impl(d, hd, stream);
}

// The MSE form of the fused last stage (include/rvip_hip.h, ABI 6): same grid as rvip_bn_apply_head.
static long long head_fwd_blocks(long long rows, long long* chunk_out) {
    long long nb = cdiv(rows, 256 * 4);
    if (nb > 1024) nb = 1024;
    const long long chunk = cdiv(rows, nb);
    if (chunk_out) *chunk_out = chunk;
    return cdiv(rows, chunk);
}

extern "C" int rvip_bn_apply_head_mse_rows(long long rows, int c, int dtype, int k) {
    if (!RVIP_DT_OK(dtype) || rows <= 0 || c <= 0 || c % RVIP_VE(dtype) || k <= 0 || k > 2) return 0;
    const int cg = c / RVIP_VE(dtype);
    if (cg > 64 || (cg & (cg - 1))) return 0;
    return (int)head_fwd_blocks(rows, nullptr);
}

extern "C" int rvip_bn_apply_head_mse(const rvip_apply_desc* d, const float* head_w, const float* head_b, const float* beta, int k, float* pred,
                                      const float* y_true, float* sums, float* dlogit, float inv_count, float dscale,
                                      float* mse_rows, size_t mse_rows_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->z || !head_w || !pred || !y_true || !sums || !workspace || !dlogit || !mse_rows || !RVIP_DT_OK(d->dtype) || k <= 0) return RVIP_EINVAL;
    const int ve = RVIP_VE(d->dtype);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->c % ve || d->drop_rate > 0.f || !(inv_count > 0.f) || !(dscale > 0.f)) return RVIP_EINVAL;
    const int cg = d->c / ve;
    if (k > 2 || d->act != RVIP_ACT_NONE || cg > 64 || (cg & (cg - 1))) return RVIP_EUNSUPPORTED;
    ApplyArgs a;
    a.z = (const unsigned char*)d->z; a.y = nullptr; a.pooled = nullptr; a.argmax = nullptr;
    a.scale = d->scale; a.shift = d->shift; a.act = d->act;
    a.inv_keep = 1.f; a.thr = 0; a.mask = nullptr; a.state = nullptr; a.layer_id = 0; a.drop = 0;
    a.n = d->n; a.h = d->h; a.w = d->w; a.c = d->c;
    const long long rows = (long long)d->n * d->h * d->w;
    long long chunk;
    const long long nb = head_fwd_blocks(rows, &chunk);
    if (workspace_bytes < (size_t)nb * 16 * sizeof(float) || mse_rows_bytes < (size_t)nb * 3 * d->c * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    HeadFuse hd{head_w, head_b, nullptr, k};
    HeadMse mse{dlogit, mse_rows, beta, inv_count, dscale};
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if (cg == 4 && k == 2) hipLaunchKernelGGL((bn_apply_head_kernel<T, 2, 1, 1, 1>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, 1, ws, mse);
        else hipLaunchKernelGGL((bn_apply_head_kernel<T, 2, 1, 1>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, 1, ws, mse);
        return 0;
    });
    int rc = check_launch();
    if (rc) return rc;
    PostHeadSums p{sums};
    return launch_fold<1, PostHeadSums>(ws, (int)nb, 16, p, s);
}

// BCE-Dice form of the fused last stage (Loss_and_metrics.py:229-245, the Train notebook's loss): three row sets, nothing per pixel
// beyond the heat-map; rows [rvip_bn_apply_head_mse_rows()][3 k_cap + 1][C]
extern "C" int rvip_bn_apply_head_bcedice(const rvip_apply_desc* d, const float* head_w, const float* head_b, const float* beta, int k, float* pred,
                                          const float* y_true, float* sums, float* rows_out, size_t rows_bytes,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->z || !head_w || !pred || !y_true || !sums || !workspace || !rows_out || !RVIP_DT_OK(d->dtype) || k <= 0) return RVIP_EINVAL;
    const int ve = RVIP_VE(d->dtype);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->c % ve || d->drop_rate > 0.f) return RVIP_EINVAL;
    const int cg = d->c / ve;
    if (k > 2 || d->act != RVIP_ACT_NONE || cg > 64 || (cg & (cg - 1))) return RVIP_EUNSUPPORTED;
    if (d->dtype == RVIP_F32) return RVIP_EUNSUPPORTED;       // the Q row carries six columns per lane: 8-element channel vectors only
    ApplyArgs a;
    a.z = (const unsigned char*)d->z; a.y = nullptr; a.pooled = nullptr; a.argmax = nullptr;
    a.scale = d->scale; a.shift = d->shift; a.act = d->act;
    a.inv_keep = 1.f; a.thr = 0; a.mask = nullptr; a.state = nullptr; a.layer_id = 0; a.drop = 0;
    a.n = d->n; a.h = d->h; a.w = d->w; a.c = d->c;
    const long long rows = (long long)d->n * d->h * d->w;
    long long chunk;
    const long long nb = head_fwd_blocks(rows, &chunk);
    if (workspace_bytes < (size_t)nb * 16 * sizeof(float) || rows_bytes < (size_t)nb * 7 * d->c * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    HeadFuse hd{head_w, head_b, nullptr, k};
    HeadMse mse{nullptr, rows_out, beta, 0.f, 1.f};
    by_dtype(d->dtype, [&](auto t) {
        using T = decltype(t);
        if constexpr (sizeof(T) == 2) {
            if (cg == 4 && k == 2) hipLaunchKernelGGL((bn_apply_head_kernel<T, 2, 1, 2, 1>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, 1, ws, mse);
            else hipLaunchKernelGGL((bn_apply_head_kernel<T, 2, 1, 2>), dim3((unsigned)nb), dim3(256), 0, s, a, hd, pred, y_true, rows, chunk, 1, ws, mse);
        }
        return 0;
    });
    int rc = check_launch();
    if (rc) return rc;
    PostHeadSums p{sums};
    return launch_fold<1, PostHeadSums>(ws, (int)nb, 16, p, s);
}

extern "C" int rvip_head_mse_coef(const rvip_headcoef_desc* d, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->bn || !d->head_w || !d->mse_rows || !d->head_dw || !d->head_db || !d->flags || d->nrows <= 0) return RVIP_EINVAL;
    const bool bcedice = d->loss_kind == RVIP_LOSS_BCE_DICE;
    if (d->loss_kind != RVIP_LOSS_MSE && !bcedice) return RVIP_EINVAL;
    if (bcedice ? (!d->pred || !d->y_true || !d->dcoef || !d->sums || !(d->dscale > 0.f)) : !d->dlogit) return RVIP_EINVAL;
    const rvip_bnbwd_desc* b = d->bn;
    if (!b->z || !RVIP_DT_OK(b->dtype) || b->c <= 0 || b->c % RVIP_VE(b->dtype) || b->rows <= 0) return RVIP_EINVAL;
    if (!b->gamma || !b->mean || !b->invstd || !b->dgamma || !b->dbeta || !b->coef || b->drop_rate > 0.f) return RVIP_EINVAL;
    if (!(d->min_gamma > 0.f) || !(d->max_beta_ratio > 0.f) || (d->loss_out && !d->sums)) return RVIP_EINVAL;
    if (d->k <= 0 || d->k > 2 || b->act != RVIP_ACT_RELU || b->act_after_bn) return RVIP_EUNSUPPORTED;
    HeadCoefArgs a;
    a.rows = d->mse_rows; a.nrows = d->nrows; a.w = d->head_w; a.dlogit = d->dlogit; a.k = d->k;
    a.head_dw = d->head_dw; a.head_db = d->head_db; a.sums = d->sums; a.loss_out = d->loss_out; a.inv_count = d->inv_count;
    a.z = (const unsigned char*)b->z; a.px = b->rows;
    a.gamma = b->gamma; a.beta = d->beta; a.mean = b->mean; a.invstd = b->invstd;
    a.dgamma = b->dgamma; a.dbeta = b->dbeta; a.coef = b->coef; a.flags = d->flags;
    a.n = (double)b->rows; a.c = b->c; a.min_gamma = d->min_gamma; a.max_beta_ratio = d->max_beta_ratio;
    a.w_bce = d->w_bce; a.w_dice = d->w_dice; a.lg = d->local_over_global; a.dscale = d->dscale;
    a.pred = d->pred; a.yt = d->y_true; a.dcoef = d->dcoef;
    by_dtype(b->dtype, [&](auto t) {
        using T = decltype(t);
        if (bcedice) hipLaunchKernelGGL((head_mse_coef_kernel<T, 2>), dim3((unsigned)cdiv(b->c, 32)), dim3(1024), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((head_mse_coef_kernel<T, 1>), dim3((unsigned)cdiv(b->c, 32)), dim3(1024), 0, (hipStream_t)stream, a);
        return 0;
    });
    return check_launch();
}

extern "C" int rvip_head_grad(const float* pred, const float* y_true, const float* sums, float* dlogit, float* loss_out,
                              long long rows, int k, int loss_kind, float inv_count, float local_over_global,
                              float w_bce, float w_dice, void* stream) {
    (void)hipGetLastError();
    if (!pred || !y_true || !sums || !dlogit || rows <= 0 || k <= 0 || k > RVIP_MAXK) return RVIP_EINVAL;
    if (loss_kind != RVIP_LOSS_MSE && loss_kind != RVIP_LOSS_BCE_DICE) return RVIP_EINVAL;
    const long long count = rows * k;
    hipLaunchKernelGGL(head_grad_kernel, dim3((unsigned)cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, pred, y_true, sums, dlogit, loss_out, count, k, loss_kind, inv_count, local_over_global, w_bce, w_dice);
    return check_launch();
}

extern "C" int rvip_head_bwd(const void* x, const float* w, const float* dlogit, void* dx, float* dw, float* db,
                             long long rows, int cin, int k, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !dlogit || !dw || !db || !workspace || !RVIP_DT_OK(dtype) || k <= 0 || k > RVIP_MAXK) return RVIP_EINVAL;
    RedGeom g;
    if (!red_geom(rows, cin, RVIP_VE(dtype), g)) return RVIP_EINVAL;
    if (workspace_bytes < (size_t)g.nblk * 2 * RVIP_MAXK * cin * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(head_bwd_kernel<bf16_t>, dim3(g.nblk), dim3(256), 0, s, (const unsigned char*)x, w, dlogit, (unsigned char*)dx, rows, cin, k, g, ws);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(head_bwd_kernel<f16_t>, dim3(g.nblk), dim3(256), 0, s, (const unsigned char*)x, w, dlogit, (unsigned char*)dx, rows, cin, k, g, ws);
    else hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(g.nblk), dim3(256), 0, s, (const unsigned char*)x, w, dlogit, (unsigned char*)dx, rows, cin, k, g, ws);
    int rc = check_launch();
    if (rc) return rc;
    PostHeadBwd p{dw, db, cin, k, RVIP_MAXK};
    return launch_fold_k<PostHeadBwd>(ws, g.nblk, cin, 2 * RVIP_MAXK, p, s);
}

// rows the first-layer weight gradient leaves in its workspace: [rows][9][cout] (the fold rvip_conv3x3_c1_wgrad runs behind its kernel,
// or -- dw == NULL -- leaves to the caller: one entry {nrows = rows, width = 9 * cout} of rvip_fold_rows_batch(..., wide = 0))
extern "C" int rvip_conv3x3_c1_wgrad_rows(int n, int h, int w_, int cout, int dtype) {
    RedGeom g;
    if (!RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w_ <= 0 || !red_geom((long long)n * h * w_, cout, RVIP_VE(dtype), g)) return 0;
    if (256 % g.cg == 0) {
        const long long nt = (long long)n * cdiv(w_, 32) * cdiv(h, 8);
        return (int)(nt < 1024 ? nt : 1024);
    }
    return g.nblk;
}

extern "C" int rvip_conv3x3_c1_wgrad(const void* x, const void* dy, float* dw, int n, int h, int w_, int cout, int dtype,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !dy || !workspace || !RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w_ <= 0) return RVIP_EINVAL;
    RedGeom g;
    if (!red_geom((long long)n * h * w_, cout, RVIP_VE(dtype), g)) return RVIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (256 % g.cg == 0) {
        const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
        long long nt = (long long)n * tx * ty;
        const int nb = (int)(nt < 1024 ? nt : 1024);
        if (workspace_bytes < (size_t)nb * 9 * cout * sizeof(float)) return RVIP_EWORKSPACE;
        static const bool mfma = [] { const char* e = getenv("RVIP_C1_WGRAD_MFMA"); return !(e && e[0] == '0'); }();      // 0: the VALU form
        if (mfma && cout == 32 && dtype == RVIP_BF16) hipLaunchKernelGGL(c1_wgrad_mfma<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, (const unsigned char*)dy, n, h, w_, tx, ty, ws);
        else if (mfma && cout == 32 && dtype == RVIP_F16) hipLaunchKernelGGL(c1_wgrad_mfma<f16_t>, dim3(nb), dim3(256), 0, s, (const f16_t*)x, (const unsigned char*)dy, n, h, w_, tx, ty, ws);
        else if (dtype == RVIP_BF16) hipLaunchKernelGGL(c1_wgrad_tiled<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, 1, 0, 1, 0);
        else if (dtype == RVIP_F16) hipLaunchKernelGGL(c1_wgrad_tiled<f16_t>, dim3(nb), dim3(256), 0, s, (const f16_t*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, 1, 0, 1, 0);
        else hipLaunchKernelGGL(c1_wgrad_tiled<float>, dim3(nb), dim3(256), 0, s, (const float*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, 1, 0, 1, 0);
        int rc2 = check_launch();
        if (rc2 || !dw) return rc2;                            // dw == NULL: the rows stay for the caller's batched fold
        PostC1Wgrad p2{dw, cout, cout};
        return launch_fold_k<PostC1Wgrad>(ws, nb, cout, 9, p2, s);
    }
    if (workspace_bytes < (size_t)g.nblk * 9 * cout * sizeof(float)) return RVIP_EWORKSPACE;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(c1_wgrad_kernel<bf16_t>, dim3(g.nblk), dim3(256), 0, s, (const bf16_t*)x, (const unsigned char*)dy, n, h, w_, cout, g, ws);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(c1_wgrad_kernel<f16_t>, dim3(g.nblk), dim3(256), 0, s, (const f16_t*)x, (const unsigned char*)dy, n, h, w_, cout, g, ws);
    else hipLaunchKernelGGL(c1_wgrad_kernel<float>, dim3(g.nblk), dim3(256), 0, s, (const float*)x, (const unsigned char*)dy, n, h, w_, cout, g, ws);
    int rc = check_launch();
    if (rc || !dw) return rc;
    PostC1Wgrad p{dw, cout, cout};
    return launch_fold_k<PostC1Wgrad>(ws, g.nblk, cout, 9, p, s);
}

// First layer with 2..4 input channels (NHWC x): dw[9][Cin][Cout], one 9-tap pass per input channel
extern "C" int rvip_conv3x3_cn_wgrad(const void* x, const void* dy, float* dw, int n, int h, int w_, int cin, int cout, int dtype,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !dy || !dw || !workspace || !RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w_ <= 0 || cin < 1 || cin > 4) return RVIP_EINVAL;
    const int ve = RVIP_VE(dtype);
    if (cout <= 0 || cout % ve || 256 % (cout / ve)) return RVIP_EUNSUPPORTED;
    const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
    long long nt = (long long)n * tx * ty;
    const int nb = (int)(nt < 1024 ? nt : 1024);
    if (workspace_bytes < (size_t)nb * 9 * cout * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    for (int ci = 0; ci < cin; ++ci) {
        by_dtype(dtype, [&](auto t) {
            using T = decltype(t);
            hipLaunchKernelGGL(c1_wgrad_tiled<T>, dim3(nb), dim3(256), 0, s, (const T*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, 1, 0, cin, ci);
            return 0;
        });
        int rc = check_launch();
        if (rc) return rc;
        PostC1Wgrad p{dw + (size_t)ci * cout, cout, cin * cout};
        rc = launch_fold_k<PostC1Wgrad>(ws, nb, cout, 9, p, s);
        if (rc) return rc;
    }
    return RVIP_OK;
}

// Conv3D first layer (Cin = 1): dw[27][Cout], one 9-tap pass per depth tap with x shifted inside the volume
extern "C" int rvip_conv3d_c1_wgrad(const void* x, const void* dy, float* dw, int n, int depth, int h, int w_, int cout, int dtype,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !dy || !dw || !workspace || !RVIP_DT_OK(dtype) || n <= 0 || depth <= 0 || n % depth || h <= 0 || w_ <= 0) return RVIP_EINVAL;
    const int ve = RVIP_VE(dtype);
    if (cout <= 0 || cout % ve || 256 % (cout / ve)) return RVIP_EINVAL;
    const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
    long long nt = (long long)n * tx * ty;
    const int nb = (int)(nt < 1024 ? nt : 1024);
    if (workspace_bytes < (size_t)nb * 9 * cout * sizeof(float)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    for (int kdi = 0; kdi < 3; ++kdi) {
        if (dtype == RVIP_BF16) hipLaunchKernelGGL(c1_wgrad_tiled<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, depth, kdi - 1, 1, 0);
        else if (dtype == RVIP_F16) hipLaunchKernelGGL(c1_wgrad_tiled<f16_t>, dim3(nb), dim3(256), 0, s, (const f16_t*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, depth, kdi - 1, 1, 0);
        else hipLaunchKernelGGL(c1_wgrad_tiled<float>, dim3(nb), dim3(256), 0, s, (const float*)x, (const unsigned char*)dy, n, h, w_, cout, tx, ty, ws, depth, kdi - 1, 1, 0);
        int rc = check_launch();
        if (rc) return rc;
        PostC1Wgrad p{dw + (size_t)kdi * 9 * cout, cout, cout};
        rc = launch_fold_k<PostC1Wgrad>(ws, nb, cout, 9, p, s);
        if (rc) return rc;
    }
    return RVIP_OK;
}

extern "C" int rvip_landmarks(const float* pred, long long* idx_out, uint8_t* mask_out, int n, int hw, int k, float thr, void* stream) {
    (void)hipGetLastError();
    if (!pred || !idx_out || n <= 0 || hw <= 0 || k <= 0) return RVIP_EINVAL;
    hipLaunchKernelGGL(landmarks_kernel, dim3((unsigned)(n * k)), dim3(256), 0, (hipStream_t)stream, pred, idx_out, mask_out, hw, k, thr);
    return check_launch();
}

extern "C" int rvip_adam_step(float* theta, const float* grad, float* m, float* v, long long count, float beta1, float beta2,
                              float eps, float grad_scale, const uint32_t* state, void* stream) {
    (void)hipGetLastError();
    if (!theta || !grad || !m || !v || !state || count <= 0) return RVIP_EINVAL;
    long long nb = cdiv(count, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, theta, grad, m, v, count, beta1, beta2, eps, grad_scale, state);
    return check_launch();
}

extern "C" int rvip_scale_f32(float* x, long long count, float scale, void* stream) {
    (void)hipGetLastError();
    if (!x || count <= 0) return RVIP_EINVAL;
    long long nb = cdiv(count, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, count, scale);
    return check_launch();
}

extern "C" int rvip_state_tick(uint32_t* state, void* stream) {
    (void)hipGetLastError();
    if (!state) return RVIP_EINVAL;
    hipLaunchKernelGGL(state_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
    return check_launch();
}

extern "C" int rvip_convert(const void* src, int sdt, void* dst, int ddt, long long count, void* stream) {
    (void)hipGetLastError();
    if (!src || !dst || count <= 0 || !RVIP_DT_OK(sdt) || !RVIP_DT_OK(ddt)) return RVIP_EINVAL;
    long long nb = cdiv(count, 256);
    if (nb > 4096) nb = 4096;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)nb), blk(256);
    by_dtype(sdt, [&](auto ts) {
        return by_dtype(ddt, [&](auto td) {
            using S = decltype(ts); using D = decltype(td);
            hipLaunchKernelGGL((convert_kernel<S, D>), grid, blk, 0, s, (const S*)src, (D*)dst, count);
            return 0;
        });
    });
    return check_launch();
}
