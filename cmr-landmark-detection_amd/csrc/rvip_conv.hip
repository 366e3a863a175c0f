// 3x3 "same" convolution for gfx950 as implicit GEMM on MFMA (forward and data-gradient), the
// Cin = 1 first layer, and the weight re-layout.  Replaces Conv2D (+UpSampling2D/+Concatenate in
// front of it) of src/models/KerasLayers.py:683,689,756-767 of the reference.
//
// Tiling (one workgroup = 256 threads = 4 waves):
//   output tile  = TH x TW pixels (256 pixels; TW = 32 or 16) x BN output channels (32 or 64)
//   K loop       = input channels in chunks of 64 BYTES per pixel (32 bf16 / 16 f32)
//   LDS          = halo patch [(TH+2)*(TW+2)][64 B + 16 B pad]  +  weights [9][BN][64 B + 16 B pad]
//                  (80-byte rows make every ds_read_b128 of 32 consecutive rows conflict-free)
//   staging      = global -> registers -> LDS, with the NEXT chunk's global loads in flight while
//                  the current chunk is multiplied (issue-early / write-late)
//   MFMA         = D[co][pixel] += W[co][k] * X[k][pixel]   (weights are the A operand so that a lane
//                  ends up with 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores)
//                  bf16: v_mfma_f32_32x32x16_bf16; f32: 4 x v_mfma_f32_32x32x2_f32 per 16-byte fragment
//                  (exact fp32, k-ordered fma chain; the k <-> channel map only has to agree between
//                  the two operands, and both read the same 16-byte channel group).
#include "rvip_common.h"

namespace rvip {

int g_last_hip_error = 0;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
    }
};

struct ConvArgs {
    const unsigned char* x0; const unsigned char* x1;
    const unsigned char* wp; const float* bias;
    unsigned char* y; unsigned char* y1;
    int c0, c1, up0, csplit;
    int n, h, w, cin, cout, act;
    int tiles_x, tiles_y;
};

template <typename T, int TW, int NCT>
__global__ __launch_bounds__(256, 2) void conv3x3_igemm(ConvArgs a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HHT = TH + 2, NHALO = HWD * HHT;
    constexpr int BN = NCT * 32, PSTR = 80, VE = Vec<T>::VE, KCE = 4 * VE;
    constexpr int NIP = (NHALO * 4 + 255) / 256;
    constexpr int NWP = (9 * BN * 4 + 255) / 256;
    constexpr int LDS_IN = NHALO * PSTR;
    static_assert(LDS_IN % 16 == 0, "lds carve");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lin = smem;
    unsigned char* lw = smem + LDS_IN;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, hf = lane >> 5;
    int bx = blockIdx.x;
    const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty_i = bx % a.tiles_y;
    const int n = bx / a.tiles_y;
    const int ty0 = ty_i * TH, tx0 = tx_i * TW, co0 = blockIdx.y * BN;
    const int nchunks = (a.cin + KCE - 1) / KCE;
    const int h0 = a.h >> a.up0, w0 = a.w >> a.up0;

    // per-thread staging geometry (independent of the chunk)
    int pix0[NIP], pix1[NIP];          // source pixel indices (-1 = outside the image -> zero)
#pragma unroll
    for (int i = 0; i < NIP; ++i) {
        const int id = tid + 256 * i;
        const int hp = id >> 2;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        const bool ok = (id < NHALO * 4) && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        pix0[i] = ok ? ((n * h0 + (gy >> a.up0)) * w0 + (gx >> a.up0)) : -1;
        pix1[i] = ok ? ((n * a.h + gy) * a.w + gx) : -1;
    }

    uint4 rin[NIP], rwt[NWP];
    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            const int id = tid + 256 * i;
            const int c = kc * KCE + (id & 3) * VE;
            uint4 r = make_uint4(0, 0, 0, 0);
            if (pix0[i] >= 0) {
                if (c < a.c0) r = *reinterpret_cast<const uint4*>(a.x0 + ((size_t)pix0[i] * a.c0 + c) * sizeof(T));
                else if (c < a.cin) r = *reinterpret_cast<const uint4*>(a.x1 + ((size_t)pix1[i] * a.c1 + (c - a.c0)) * sizeof(T));
            }
            rin[i] = r;
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int id = tid + 256 * i;
            const int row = id >> 2;
            const int tap = row / BN, co = co0 + (row & (BN - 1));
            const int c = kc * KCE + (id & 3) * VE;
            uint4 r = make_uint4(0, 0, 0, 0);
            if (id < 9 * BN * 4 && co < a.cout && c < a.cin)
                r = *reinterpret_cast<const uint4*>(a.wp + ((size_t)(tap * a.cout + co) * a.cin + c) * sizeof(T));
            rwt[i] = r;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            const int id = tid + 256 * i;
            if (id < NHALO * 4) *reinterpret_cast<uint4*>(lin + (id >> 2) * PSTR + (id & 3) * 16) = rin[i];
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int id = tid + 256 * i;
            if (id < 9 * BN * 4) *reinterpret_cast<uint4*>(lw + (id >> 2) * PSTR + (id & 3) * 16) = rwt[i];
        }
    };

    int in_off[2], w_off[NCT];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int P = wv * 64 + pt * 32 + j;
        in_off[pt] = ((P / TW) * HWD + (P % TW)) * PSTR + hf * 16;
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) w_off[ct] = (ct * 32 + j) * PSTR + hf * 16;

    f32x16 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    gload(0);
    for (int kc = 0; kc < nchunks; ++kc) {
        __syncthreads();                       // everyone finished reading the previous chunk
        lstore();
        __syncthreads();
        if (kc + 1 < nchunks) gload(kc + 1);   // in flight during the MFMAs below
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * HWD + (tap % 3)) * PSTR;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint4 fa[NCT], fb[2];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) fa[ct] = *reinterpret_cast<const uint4*>(lw + w_off[ct] + tap * BN * PSTR + g * 32);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) fb[pt] = *reinterpret_cast<const uint4*>(lin + in_off[pt] + toff + g * 32);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) Mma<T>::run(fa[ct], fb[pt], acc[ct][pt]);
            }
        }
    }

    // epilogue: lane (j, hf) holds, for pixel j of each pixel tile, channels 8q + 4hf + {0..3}
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int P = wv * 64 + pt * 32 + j;
        const int gy = ty0 + P / TW, gx = tx0 + P % TW;
        if (gy >= a.h || gx >= a.w) continue;
        const size_t pix = ((size_t)n * a.h + gy) * a.w + gx;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = co0 + ct * 32 + 8 * q + 4 * hf;
                if (co >= a.cout) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = acc[ct][pt][4 * q + i];
                    if (a.bias) t += a.bias[co + i];
                    v[i] = act_fwd(t, a.act);
                }
                unsigned char* dst;
                if (a.y1 && co >= a.csplit) dst = a.y1 + (pix * (a.cout - a.csplit) + (co - a.csplit)) * sizeof(T);
                else dst = a.y + (pix * (a.y1 ? a.csplit : a.cout) + co) * sizeof(T);
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o;
                    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2*>(dst) = o;
                }
            }
        }
    }
}

template <typename T, int TW, int NCT>
static int launch_igemm(const ConvArgs& a, hipStream_t s) {
    constexpr int TH = 256 / TW;
    constexpr int lds = (TW + 2) * (TH + 2) * 80 + 9 * NCT * 32 * 80;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_igemm<T, TW, NCT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
        attr_done = true;
    }
    ConvArgs b = a;
    b.tiles_x = (int)cdiv(a.w, TW);
    b.tiles_y = (int)cdiv(a.h, TH);
    dim3 grid((unsigned)((long long)a.n * b.tiles_x * b.tiles_y), (unsigned)cdiv(a.cout, NCT * 32));
    hipLaunchKernelGGL((conv3x3_igemm<T, TW, NCT>), grid, dim3(256), lds, s, b);
    return check_launch();
}

template <typename T>
static int dispatch_igemm(const ConvArgs& a, hipStream_t s) {
    const bool wide = a.w > 16;
    const bool two = a.cout > 32;
    if (wide) return two ? launch_igemm<T, 32, 2>(a, s) : launch_igemm<T, 32, 1>(a, s);
    return two ? launch_igemm<T, 16, 2>(a, s) : launch_igemm<T, 16, 1>(a, s);
}

// ---------------------------------------------------------------------------------------------
// weight re-layout: fp32 HWIO -> packed [9][Cout][Cin] (forward) and [9][Cin][Cout] rotated (dgrad)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_w_kernel(const float* __restrict__ w, int cin, int cout, T* wf, T* wd) {
    const long long total = 9LL * cin * cout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int o = (int)(i % cout);
        const int ci = (int)((i / cout) % cin);
        const int t = (int)(i / ((long long)cout * cin));
        const float v = w[i];
        if constexpr (sizeof(T) == 4) {
            if (wf) wf[((size_t)t * cout + o) * cin + ci] = v;
            if (wd) wd[((size_t)(8 - t) * cin + ci) * cout + o] = v;
        } else {
            const uint16_t b = f32_to_bf16(v);
            if (wf) wf[((size_t)t * cout + o) * cin + ci].bits = b;
            if (wd) wd[((size_t)(8 - t) * cin + ci) * cout + o].bits = b;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// first layer: Cin = 1.  One thread = one pixel x VE output channels; bandwidth-bound on the store.
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(p->bits); }

template <typename T>
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, unsigned char* y,
                                                         int n, int h, int wd, int cout, int act) {
    constexpr int VE = Vec<T>::VE;
    const int cg = cout / VE;
    const long long total = (long long)n * h * wd * cg;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    if (idx >= total) return;
    const int cv = (int)(idx % cg);
    const long long p = idx / cg;
    const int px = (int)(p % wd), py = (int)((p / wd) % h);
    const long long img = p / ((long long)wd * h);
    float xin[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        xin[t] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd) ? ld1<T>(x + (img * h + yy) * wd + xx) : 0.f;
    }
    float v[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int co = cv * VE + e;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) s = fmaf(xin[t], w[t * cout + co], s);
        if (bias) s += bias[co];
        v[e] = act_fwd(s, act);
    }
    Vec<T>::store(y + ((size_t)p * cout + cv * VE) * sizeof(T), v);
}

}  // namespace rvip

using namespace rvip;

extern "C" int rvip_abi_version(void) { return 1; }
extern "C" const char* rvip_build_info(void) { return "rvip_hip gfx950 wave64 mfma"; }
extern "C" int rvip_last_hip_error(void) { return g_last_hip_error; }

extern "C" int rvip_conv3x3_fwd(const rvip_conv3x3_desc* d, void* stream) {
    (void)hipGetLastError();
    if (!d || !d->x0 || !d->w_packed || !d->y) return RVIP_EINVAL;
    const int ve = d->dtype == RVIP_BF16 ? 8 : 4;
    if (d->dtype != RVIP_BF16 && d->dtype != RVIP_F32) return RVIP_EINVAL;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->c0 <= 0) return RVIP_EINVAL;
    if (d->c0 % ve || d->c1 % ve || d->cout % 4) return RVIP_EINVAL;
    if ((d->c1 > 0) != (d->x1 != nullptr)) return RVIP_EINVAL;
    if (d->up0 && ((d->h | d->w) & 1)) return RVIP_EINVAL;
    if (d->up0 != 0 && d->up0 != 1) return RVIP_EINVAL;
    if (d->y1 && (d->csplit <= 0 || d->csplit >= d->cout || d->csplit % 4)) return RVIP_EINVAL;
    if ((long long)d->n * d->h * d->w >= (1LL << 31)) return RVIP_EINVAL;
    ConvArgs a;
    a.x0 = (const unsigned char*)d->x0; a.x1 = (const unsigned char*)d->x1;
    a.wp = (const unsigned char*)d->w_packed; a.bias = d->bias;
    a.y = (unsigned char*)d->y; a.y1 = (unsigned char*)d->y1;
    a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0; a.csplit = d->csplit;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->c0 + d->c1; a.cout = d->cout; a.act = d->act;
    a.tiles_x = a.tiles_y = 0;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == RVIP_BF16 ? dispatch_igemm<bf16_t>(a, s) : dispatch_igemm<float>(a, s);
}

extern "C" int rvip_pack_conv3x3_weights(const float* w, int cin, int cout, int dtype, void* wf, void* wd, void* stream) {
    (void)hipGetLastError();
    if (!w || cin <= 0 || cout <= 0 || (!wf && !wd)) return RVIP_EINVAL;
    const long long total = 9LL * cin * cout;
    const int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(pack_w_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (bf16_t*)wf, (bf16_t*)wd);
    else if (dtype == RVIP_F32) hipLaunchKernelGGL(pack_w_kernel<float>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (float*)wf, (float*)wd);
    else return RVIP_EINVAL;
    return check_launch();
}

extern "C" int rvip_conv3x3_c1_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int cout,
                                   int act, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !y || n <= 0 || h <= 0 || w_ <= 0) return RVIP_EINVAL;
    const int ve = dtype == RVIP_BF16 ? 8 : 4;
    if (cout <= 0 || cout % ve) return RVIP_EINVAL;
    const long long total = (long long)n * h * w_ * (cout / ve);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(conv3x3_c1_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act);
    else if (dtype == RVIP_F32) hipLaunchKernelGGL(conv3x3_c1_kernel<float>, grid, dim3(256), 0, s, (const float*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act);
    else return RVIP_EINVAL;
    return check_launch();
}
