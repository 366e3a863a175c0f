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

namespace rvip {

struct WgArgs {
    const unsigned char* x0; const unsigned char* x1; const unsigned char* dy;
    float* slab;
    int c0, c1, up0;
    int n, h, w, cin, cout;
    int tiles_x, tiles_y, ntiles, nsplit;
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
                        r = *reinterpret_cast<const uint4*>(a.x0 + (pix * a.c0 + c) * sizeof(T));
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
                const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int toff = ((t / 3) * HWD + (t % 3)) * ROWB;
                    const s16x4 x0 = tr_read(lx + xaddr[0] + toff);
                    const s16x4 x1 = tr_read(lx + xaddr[1] + toff);
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
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

// dw[i] = sum_k slab[k][i] in split order.  256 threads = 32 elements x 8 split groups so that small kernels
// (9*32*32 elements, hundreds of splits) still fill the chip; fixed summation order -> reproducible.
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* __restrict__ slab, int nsplit, long long count, float* __restrict__ dw) {
    __shared__ float sh[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long i = blockIdx.x * 32LL + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < count) {
        int k = g;
        for (; k + 24 < nsplit; k += 32) {
            s0 += slab[(size_t)k * count + i];
            s1 += slab[(size_t)(k + 8) * count + i];
            s2 += slab[(size_t)(k + 16) * count + i];
            s3 += slab[(size_t)(k + 24) * count + i];
        }
        for (; k < nsplit; k += 8) s0 += slab[(size_t)k * count + i];
    }
    sh[g][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) t += sh[gg][e];
        dw[i] = t;
    }
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

template <typename T, int TW>
static int launch_wgrad(const WgArgs& a, hipStream_t s) {
    constexpr int TH = 256 / TW;
    constexpr int rowb = 32 * (int)sizeof(T);
    constexpr int lds_tiles = (TW + 2) * (TH + 2) * rowb + 256 * rowb;
    constexpr int lds = lds_tiles > 9 * 32 * 32 * 4 ? lds_tiles : 9 * 32 * 32 * 4;
    static bool attr_done = false;
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

using namespace rvip;

extern "C" size_t rvip_conv3x3_wgrad_workspace(int n, int h, int w, int cin, int cout) {
    int tw, tx, ty, nt, ns;
    wgrad_geometry(n, h, w, cin, cout, tw, tx, ty, nt, ns);
    return (size_t)ns * 9 * cin * cout * sizeof(float);
}

extern "C" int rvip_conv3x3_wgrad(const rvip_wgrad3x3_desc* d, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->x0 || !d->dy || !d->dw || !d->workspace) return RVIP_EINVAL;
    if (d->dtype != RVIP_BF16 && d->dtype != RVIP_F32) return RVIP_EINVAL;
    const int ve = d->dtype == RVIP_BF16 ? 8 : 4;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->c0 <= 0) return RVIP_EINVAL;
    if (d->c0 % ve || d->c1 % ve || d->cout % ve) return RVIP_EINVAL;
    if ((d->c1 > 0) != (d->x1 != nullptr)) return RVIP_EINVAL;
    if (d->up0 != 0 && d->up0 != 1) return RVIP_EINVAL;
    if (d->up0 && ((d->h | d->w) & 1)) return RVIP_EINVAL;
    WgArgs a;
    a.x0 = (const unsigned char*)d->x0; a.x1 = (const unsigned char*)d->x1; a.dy = (const unsigned char*)d->dy;
    a.slab = (float*)d->workspace;
    a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->c0 + d->c1; a.cout = d->cout;
    int tw;
    wgrad_geometry(a.n, a.h, a.w, a.cin, a.cout, tw, a.tiles_x, a.tiles_y, a.ntiles, a.nsplit);
    const size_t need = (size_t)a.nsplit * 9 * a.cin * a.cout * sizeof(float);
    if (d->workspace_bytes < need) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (d->dtype == RVIP_BF16) rc = tw == 32 ? launch_wgrad<bf16_t, 32>(a, s) : launch_wgrad<bf16_t, 16>(a, s);
    else rc = tw == 32 ? launch_wgrad<float, 32>(a, s) : launch_wgrad<float, 16>(a, s);
    if (rc) return rc;
    const long long count = 9LL * a.cin * a.cout;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)cdiv(count, 32)), dim3(256), 0, s, a.slab, a.nsplit, count, d->dw);
    return check_launch();
}
