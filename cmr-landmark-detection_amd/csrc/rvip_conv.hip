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
#include <atomic>
#include <cstdlib>

namespace rvip {

#ifndef RVIP_KERNELS_ONLY
int g_last_hip_error = 0;
#endif

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) { acc = mfma16<bf16_t>(a, b, acc); }
};
template <> struct Mma<f16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) { acc = mfma16<f16_t>(a, b, acc); }
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
    int zs;                                   // zero-stuffed x2 read of source 0 (Conv2DTranspose): only odd (y, x) carry data
    int depth, kd;                            // Conv3D: images per volume and depth taps (3); a plain 2-D conv has 1, 1
    int down2;                                // store the 2x2 block sums of the result at half resolution (gradient of UpSampling2D)
    int subpix;                               // UpSampling2D -> conv as four 2x2-tap phase convolutions on the low-resolution input
    int nt_in;                                // non-temporal input reads (last reader of x0)
    // sums launches of a data gradient whose result is gated element-wise (Dropout backward: keep bits x 1/(1-rate); ReLU backward of a
    // BN-less stage: sign bits of its forward output): bit planes [C/32][N*H*W] of one 32-bit word per pixel, first mbits_c channels
    const uint32_t* mbits; int mbits_c; float mscale;
    uint32_t* sbits;                          // forward launches: write the sign bits (stored value > 0) of the result in that layout
    int sums_from;
    int cus;                                  // compute units the persistent grid is sized for (rvip_conv3x3_desc.cu_limit; 256 = all)
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
        const bool ok = (id < NHALO * 4) && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w && (!a.zs || ((gy & gx) & 1));
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
                    o.x = Vec<T>::pack2(v[0], v[1]);
                    o.y = Vec<T>::pack2(v[2], v[3]);
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
    static std::atomic<bool> attr_done{false};      // idempotent attribute call; atomic so concurrent host threads do not race on the flag
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
// igemm v2: LDS-DMA staging.  Same math and tile/fragment maps as conv3x3_igemm above, but
//   * global -> LDS goes through buffer_load ... lds (1 KiB per wave-instruction = 16 rows x 64 B, no VGPR
//     staging, no ds_write); an out-of-range lane (halo outside the image, channel tail, row padding) gets an
//     offset beyond num_records and the hardware writes ZEROS -> "same" zero padding costs nothing;
//   * rows are UNPADDED 64 B; bank conflicts are removed by an XOR swizzle applied on the SOURCE side: LDS row r
//     keeps logical 16-byte piece p in slot p ^ ((r >> 2) & 3), readers apply the same XOR (conflict-free for any
//     16 rows that are distinct mod 16);
//   * two input stages: the DMA of work item i+1 is in flight while item i is multiplied, one barrier per item;
//   * NW waves (8: 512-pixel tile, 4: 256-pixel tile), persistent over pixel tiles; when all K chunks of the
//     weights fit next to the two input stages they are loaded ONCE per workgroup (weight-stationary).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// raw buffer descriptor (base, stride 0, num_records = bytes, raw 32-bit format) built from wave-uniform values
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
// One LDS-DMA piece: 64 lanes x 16 B -> LDS bytes [lds_off, lds_off + 1024).  Issued as inline asm so that hipcc
// does not see an LDS store (it would fence every later ds_read with vmcnt(0) and serialise load and compute);
// completion is tracked by hand: s_waitcnt vmcnt(0) + s_barrier at the top of every work item.  M0 (the DMA's LDS
// base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}
// the same with the non-temporal hint: a streamed tensor that nothing reads again soon should not displace the others
__device__ __forceinline__ void dma16_nt(i32x4 rsrc, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(unsigned long long)(lds_void_t*)p; }

struct ConvArgs2 {
    const unsigned char* x0; const unsigned char* x1;
    const unsigned char* wp; const float* bias;
    unsigned char* y; unsigned char* y1;
    unsigned x0_bytes, x1_bytes, wp_bytes, y_bytes, y1_bytes;
    int c0, c1, up0, csplit;
    int n, h, w, cin, cout, act;
    int tiles_x, tiles_y, ntiles, wres;      // wres: weight stages resident (= nchunks) or 0 -> 2 rotating stages
    int lds_bias_off;
    int zs;
    int depth, kd;                           // Conv3D as a K loop over depth taps: chunk kc reads image n + kc / nch - kd / 2
    int down2;                               // epilogue_down2 instead of the plain epilogue
    int subpix;                              // TAPS == 4 kernels: blockIdx.z = output phase (a, b); y is [N, 2h, 2w, Cout]
    int nt_in;                               // input pieces with the non-temporal hint
    int dbg;                                 // ablation only (RVIP_DBG): 1 = no DMA after the first item, 2 = no MFMA section, 4 = DMAs fetch nothing; ws16: 4 = input DMAs fetch nothing, 8 = no weight DMA / 16 = no input DMA after the first item, 32 = no epilogue, 64 = weights requested first
    float* stats;                            // optional [gridDim.x][2][cout] partial (sum, sum of squares) of the STORED output
    const uint32_t* mbits; unsigned mbits_bytes; int mbits_c; float mscale; int lds_mb_off;      // see ConvArgs; STATS == 3 kernels
    uint32_t* sbits; unsigned sbits_bytes;   // STATS == 0 kernels
    int sums_from;                           // STATS >= 2: columns below this channel are not needed (left unwritten or partial)
    int nstg;                                // input stages in LDS: 2, or 3 (conv3x3_igemm_ws16 with resident weights where LDS allows: two items in flight)
};

// Gate four consecutive channels of one pixel by four bits of a mask word (STATS == 3): v = bit ? v * scale : 0.  The bit is
// sign-extended to a word (v_bfe_i32) and ANDed onto the value: two instructions per value, plus the multiply when scale != 1
// (Dropout backward; the ReLU gate has scale 1 -- wave-uniform).
__device__ __forceinline__ void gate4(float (&v)[4], unsigned nib, float scale) {
    const bool scaled = scale != 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = __builtin_amdgcn_sbfe((int)nib, r, 1);              // 0 or -1
        const float g = __builtin_bit_cast(float, __builtin_bit_cast(int, v[r]) & m);
        v[r] = scaled ? g * scale : g;
    }
}

// Epilogue of the data gradient of an UpSampling2D -> conv pair (KerasLayers.py:756-758): the gradient w.r.t. the
// low-resolution tensor is the sum over each 2x2 block of the full-resolution data gradient.  The block's two rows are
// two pixel tiles of the same lane (TW = 32) or lanes j and j ^ 16 (TW = 16), its two columns lanes j and j ^ 1; the
// even lane stores the sum at [N, H/2, W/2, Cout].  No bias, activation, channel split or statistics in this mode.
// Returns nothing; issues (NPT / ROWS) * NCT * (bf16 ? 2 : 4) buffer stores per wave, ROWS = TW == 32 ? 2 : 1.
template <int M> __device__ __forceinline__ float swz(float v);
__device__ __forceinline__ float lane_channel_sum(const float (&a)[16], int j);
template <typename T, int TW, int NCT, int NPT, int STATS = 0>
__device__ __forceinline__ void epilogue_down2(f32x16 (&acc)[NCT][NPT], int pbase, int j, int hf, int n, int ty0, int tx0, int co0,
                                               const ConvArgs2& a, __amdgpu_buffer_rsrc_t ry, float* st_sum = nullptr) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ROWS = TW == 32 ? 2 : 1;
    const int hl = a.h >> 1, wl = a.w >> 1;
#pragma unroll
    for (int pt = 0; pt < NPT; pt += ROWS) {
        const int P = pbase + pt * 32 + j;
        const int gy = ty0 + P / TW, gx = tx0 + P % TW;
        const bool keep = gy < a.h && gx < a.w && !(j & 1) && (TW == 32 || !(j & 16));
        const unsigned pix = (unsigned)((n * hl + (gy >> 1)) * wl + (gx >> 1));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int cbase = co0 + ct * 32;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[ct][pt][r];
                acc[ct][pt][r] = 0.f;
                if constexpr (ROWS == 2) { t += acc[ct][pt + 1][r]; acc[ct][pt + 1][r] = 0.f; }
                else t += __shfl_xor(t, 16);
                v[r] = t + __shfl_xor(t, 1);
            }
            if constexpr (STATS >= 2) {              // column sums of what is stored (the kept lanes' block sums)
                float qs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) qs[r] = keep ? Vec<T>::round(v[r]) : 0.f;
                st_sum[ct] += lane_channel_sum(qs, j);
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = cbase + 8 * q + 4 * hf;
                    const unsigned off = (keep && co < a.cout) ? (pix * a.cout + co) * 4u : OOB;
                    const u32x4v dta = {__builtin_bit_cast(unsigned, v[4 * q]), __builtin_bit_cast(unsigned, v[4 * q + 1]),
                                        __builtin_bit_cast(unsigned, v[4 * q + 2]), __builtin_bit_cast(unsigned, v[4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(dta, ry, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    unsigned ax = Vec<T>::pack2(v[8 * qq + 0], v[8 * qq + 1]);
                    unsigned ay = Vec<T>::pack2(v[8 * qq + 2], v[8 * qq + 3]);
                    unsigned bxx = Vec<T>::pack2(v[8 * qq + 4], v[8 * qq + 5]);
                    unsigned byy = Vec<T>::pack2(v[8 * qq + 6], v[8 * qq + 7]);
                    auto r0s = __builtin_amdgcn_permlane32_swap(ax, bxx, false, false);
                    auto r1s = __builtin_amdgcn_permlane32_swap(ay, byy, false, false);
                    const u32x4v dta = {r0s[0], r1s[0], r0s[1], r1s[1]};
                    const int co = cbase + 16 * qq + 8 * hf;
                    const unsigned off = (keep && co < a.cout) ? (pix * a.cout + co) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(dta, ry, off, 0, 0);
                }
            }
        }
    }
}

// TAPS = 4: sub-pixel form of UpSampling2D(2) -> conv3x3 (KerasLayers.py:756-758).  Output pixel (2i+a, 2j+b) only sees the
// low-resolution pixels (i+a-1 .. i+a) x (j+b-1 .. j+b), each through a SUM of the 3x3 taps that fall on it, so the layer
// is four 2x2-tap convolutions on the low-resolution image (16 instead of 36 multiply-adds per low-resolution pixel and
// channel pair).  blockIdx.z = phase 2a+b; weights [4 phases][4 taps (u,v)][Cout][Cin] from rvip_pack_subpixel_weights;
// the same 3x3 halo is staged and tap (u,v) reads halo position (py + a + u, px + b + v).
// NCW = number of compute waves: 4 (one per SIMD, 128 pixels each at NPIX = 512) or 8 (two per SIMD, 64 pixels each: the
// epilogue stores are spread over twice as many waves, which is what the store-bound layers with one or two K chunks per
// tile need); the four loader waves are the same in both.
// Sum over the 32 lanes of a half-wave of a[r], for 16 values r per lane, leaving in lane j the total of r = (j >> 1) & 15
// (lanes j and j ^ 1 hold the same one): a butterfly that halves the number of live values at every step - lane pairs
// 16, 8, 4, 2 apart each keep one half of the values and hand over the other - 16 ds_swizzle and 16 adds instead of the
// 80 + 80 of sixteen separate 32-lane reductions.
template <int M>
__device__ __forceinline__ float swz(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (M << 10) | 0x1f));
}
// (epilogue_down2 above uses lane_channel_sum through its forward declaration)
// value of lane ^ 1 through DPP (quad_perm [1,0,3,2]): __shfl_xor(v, 1) compiles to ds_bpermute_b32, a trip through the LDS crossbar
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_channel_sum(const float (&a)[16], int j) {
    float b[8], c[4], d[2];
    const bool b4 = j & 16, b3 = j & 8, b2 = j & 4, b1 = j & 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (b4 ? a[8 + i] : a[i]) + swz<16>(b4 ? a[i] : a[8 + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = (b3 ? b[4 + i] : b[i]) + swz<8>(b3 ? b[i] : b[4 + i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) d[i] = (b2 ? c[2 + i] : c[i]) + swz<4>(b2 ? c[i] : c[2 + i]);
    const float e = (b1 ? d[1] : d[0]) + swz<2>(b1 ? d[0] : d[1]);
    return e + swz<1>(e);
}

// STATS: 0 none; 1 BatchNormalization statistics of the stored output (sum, sum of squares: forward launches); 2 column sums only,
// for data-gradient launches (with the 2x2-sum / channel-split epilogues): rows [gridDim.x][cout]; 3 = 2 + the Dropout backward
template <typename T, int TW, int NCT, int NPIX, int STATS, int TAPS = 9, int NCW = 4>
__global__ __launch_bounds__((NCW + 4) * 64, 1) void conv3x3_igemm_ws(ConvArgs2 a) {
    static_assert(TAPS == 9 || TAPS == 4, "taps");
    constexpr int TH = NPIX / TW, HWD = TW + 2, HHT = TH + 2, NHALO = HWD * HHT;
    constexpr int NHROWS = (NHALO + 15) / 16 * 16;
    constexpr int BN = NCT * 32, WROWS = TAPS * BN;
    constexpr int IN_BYTES = NHROWS * 64, W_BYTES = WROWS * 64;
    constexpr int VE = Vec<T>::VE, KCE = 4 * VE;
    constexpr int NQI = NHROWS / 16, NQW = WROWS / 16;                 // 1 KiB DMA pieces per stage
    constexpr int QI = (NQI + 3) / 4, QW = (NQW + 3) / 4;              // per loader wave
    constexpr int NPT = NPIX / (32 * NCW);                              // 32-pixel tiles per compute wave
    static_assert(NPT >= 1 && NPT * 32 * NCW == NPIX, "pixel tiles");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lin = smem;                       // [2][IN_BYTES]
    unsigned char* lw = smem + 2 * IN_BYTES;         // [wres or 2][W_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hf = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * BN;
    const int ph = (TAPS == 4) ? (int)blockIdx.z : 0, pa = ph >> 1, pb = ph & 1;      // output phase of the sub-pixel form
    const int nch = (a.cin + KCE - 1) / KCE;         // chunks per depth tap
    const int nchunks = nch * a.kd;
    const bool resident = a.wres > 0;
    const int first_tile = blockIdx.x;
    if (first_tile >= a.ntiles) return;
    float* lbias = reinterpret_cast<float*>(smem + a.lds_bias_off);        // [BN] bias of this output-channel tile
    if (tid < BN) lbias[tid] = (a.bias && co0 + tid < a.cout) ? a.bias[co0 + tid] : 0.f;

    if (wv >= NCW) {
        // ------------------------------------------------ loader waves ------------------------------------------------
        const int lwv = wv - NCW;
        const int h0 = a.h >> a.up0, w0 = a.w >> a.up0;
        const i32x4 rs0 = make_rsrc(a.x0, a.x0_bytes);
        const i32x4 rs1 = make_rsrc(a.x1 ? a.x1 : a.x0, a.x1 ? a.x1_bytes : 0u);
        const i32x4 rsw = make_rsrc(a.wp, a.wp_bytes);
        const unsigned lds_base = lds_offset_of(smem);
        const int drow = lane >> 2, dslot = lane & 3;
        int wrel[QW], wch[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const int row = (lwv + 4 * i) * 16 + drow;
            const int tap = row / BN, co = co0 + (row & (BN - 1));
            wch[i] = (co < a.cout && row < WROWS) ? (dslot ^ ((row >> 2) & 3)) * VE : 1 << 28;
            wrel[i] = (((tap * a.cout + co) * a.cin) + (dslot ^ ((row >> 2) & 3)) * VE) * (int)sizeof(T);
        }
        // The loaders' own instructions are the critical path of staging (profiles/r01_conv_ablation.txt: one K chunk in flight
        // per workgroup, and every instruction of these waves is issued in the shadow of two MFMA-bound waves), so whatever does
        // not depend on the K chunk is hoisted: per wave the tile-independent byte offsets of its pieces, per TILE the offsets
        // with the image-border test folded in (t0 / t1: offset inside the tile's window of source 0 / 1, or OOB), computed in
        // the idle window after the tile's predecessor has issued its last chunk.  A full chunk then costs one v_add per piece.
        unsigned wfast[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) wfast[i] = (wch[i] != (1 << 28)) ? (unsigned)wrel[i] : OOB;
        int ihy[QI], ihx[QI], irel0[QI], irel1[QI], ich[QI];
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int row = (lwv + 4 * i) * 16 + drow;
            const int hy = row / HWD, hx = row - hy * HWD;
            ich[i] = (dslot ^ ((hx >> 2) & 3)) * VE;
            ihy[i] = (row < NHALO) ? hy - 1 : -100000;
            ihx[i] = hx - 1;
            irel0[i] = ((((hy - 1) >> a.up0) * w0 + ((hx - 1) >> a.up0)) * a.c0 + ich[i]) * (int)sizeof(T);
            irel1[i] = (((hy - 1) * a.w + (hx - 1)) * a.c1 + ich[i]) * (int)sizeof(T);
        }
        unsigned t0[QI], t1[QI];                         // prepared tile: per-piece offsets (border test folded in)
        int p_nmod = 0, p_tb0 = 0, p_tb1 = 0;
        const int img0_bytes = h0 * w0 * a.c0 * (int)sizeof(T), img1_bytes = a.h * a.w * a.c1 * (int)sizeof(T);
        // STATS == 3: the mask words of the tile's OUTPUT pixels, one 32-channel plane per channel tile, [TH][TW] words per plane, are
        // one 1 KiB piece per loader wave (NCT * NPIX / 256 <= 4 pieces), double-buffered by tile parity; issued with chunk 0 of the tile
        constexpr int MB_PLANE = TH * TW * 4, NPM = MB_PLANE / 1024, MB_TILE = NCT * MB_PLANE, MB_LPR = TW / 4;
        const bool mb_mine = STATS == 3 && lwv < NCT * NPM && co0 + (lwv / NPM) * 32 < a.mbits_c;
        const int mb_row = (lwv % NPM) * (64 / MB_LPR) + lane / MB_LPR, mb_x = (lane % MB_LPR) * 4;
        const i32x4 rsm = make_rsrc(STATS == 3 ? (const void*)a.mbits : (const void*)a.x0, STATS == 3 ? a.mbits_bytes : 0u);
        unsigned p_mb = OOB;
        int mb_par = 0;
        auto prep_tile = [&](int tile) __attribute__((always_inline)) {
            unsigned bx = (unsigned)tile;
            const unsigned tx_i = bx % (unsigned)a.tiles_x; bx /= (unsigned)a.tiles_x;
            const unsigned ty_i = bx % (unsigned)a.tiles_y;
            const int n_out = (int)(bx / (unsigned)a.tiles_y);
            const int ty0 = (int)ty_i * TH, tx0 = (int)tx_i * TW;
            p_nmod = a.depth > 1 ? n_out % a.depth : 0;
            p_tb0 = ((n_out * h0 + (ty0 >> a.up0)) * w0 + (tx0 >> a.up0)) * a.c0 * (int)sizeof(T);
            p_tb1 = ((n_out * a.h + ty0) * a.w + tx0) * a.c1 * (int)sizeof(T);
            if constexpr (STATS == 3) {
                const bool ok = mb_mine && ty0 + mb_row < a.h && tx0 + mb_x < a.w;
                p_mb = ok ? (unsigned)(((co0 / 32 + lwv / NPM) * a.n * a.h + n_out * a.h + ty0 + mb_row) * a.w + tx0 + mb_x) * 4u : OOB;
            }
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int gy = ty0 + ihy[i], gx = tx0 + ihx[i];
                const bool ok = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
                const bool ok0 = a.zs ? (ok && ((gy & gx) & 1)) : ok;
                t0[i] = ok0 ? (unsigned)irel0[i] : OOB;
                t1[i] = ok ? (unsigned)irel1[i] : OOB;
            }
        };
        auto issue_weights = [&](int kc, int wstage) __attribute__((always_inline)) {
            const int kdi = a.kd > 1 ? kc / nch : 0;
            const int cbase = (kc - kdi * nch) * KCE;
            const int tapbase = ((kdi + ph) * TAPS * a.cout * a.cin + cbase) * (int)sizeof(T);
            const unsigned lw0 = lds_base + 2 * IN_BYTES + wstage * W_BYTES + lwv * 1024;      // + i * 4096 per piece
            if (a.cin - cbase >= KCE) {                  // full chunk: valid rows carry all their channels
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    if (i < QW - 1 || lwv + 4 * i < NQW) dma16(rsw, wfast[i] + (unsigned)tapbase, lw0 + i * 4096);   // OOB + tapbase stays >= 2^31
                }
            } else {
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    if (i < QW - 1 || lwv + 4 * i < NQW) {
                        const unsigned off = (wch[i] < a.cin - cbase) ? (unsigned)(wrel[i] + tapbase) : OOB;
                        dma16(rsw, off, lw0 + i * 4096);
                    }
                }
            }
        };
        // stages chunk kc of the PREPARED tile
        auto issue_input = [&](int kc, int stage) __attribute__((always_inline)) {
            const int kdi = a.kd > 1 ? kc / nch : 0;
            const int dsh = kdi - (a.kd >> 1);               // depth tap: the image dsh slices away, zeros outside the volume
            const bool dok = (unsigned)(p_nmod + dsh) < (unsigned)a.depth;
            const int cbase = (kc - kdi * nch) * KCE;
            const bool from0 = cbase < a.c0;                 // chunks never straddle the two sources (host checks)
            const int cb = from0 ? cbase : cbase - a.c0;
            const int crem = dok ? (from0 ? a.c0 : a.c1) - cb : 0;
            const unsigned li0 = lds_base + stage * IN_BYTES + lwv * 1024;                      // + i * 4096 per piece
            const unsigned base = (unsigned)((from0 ? p_tb0 + dsh * img0_bytes : p_tb1 + dsh * img1_bytes) + cb * (int)sizeof(T));
            if constexpr (STATS == 3) {
                if (kc == 0) {                               // first chunk of a tile: its mask words ride along (wave-uniform)
                    if (lwv < NCT * NPM) dma16(rsm, p_mb, lds_base + a.lds_mb_off + mb_par * MB_TILE + lwv * 1024);
                    mb_par ^= 1;
                }
            }
            if (crem >= KCE && !a.nt_in) {                   // full chunk inside the volume: offset = prepared + base
                if (from0) {
#pragma unroll
                    for (int i = 0; i < QI; ++i) {
                        if (i < QI - 1 || lwv + 4 * i < NQI) dma16(rs0, t0[i] + base, li0 + i * 4096);   // OOB + base stays >= 2^31
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < QI; ++i) {
                        if (i < QI - 1 || lwv + 4 * i < NQI) dma16(rs1, t1[i] + base, li0 + i * 4096);
                    }
                }
            } else {                                         // partial chunk / depth tap outside the volume / streaming hint
                const i32x4 rs = from0 ? rs0 : rs1;
#pragma unroll
                for (int i = 0; i < QI; ++i) {
                    if (i < QI - 1 || lwv + 4 * i < NQI) {
                        const unsigned t = from0 ? t0[i] : t1[i];
                        const unsigned off = (t != OOB && ich[i] < crem) ? t + base : OOB;
                        if (a.nt_in) dma16_nt(rs, off, li0 + i * 4096);
                        else dma16(rs, off, li0 + i * 4096);
                    }
                }
            }
        };
        if (resident) for (int kc = 0; kc < nchunks; ++kc) issue_weights(kc, kc);
        else issue_weights(0, 0);
        prep_tile(first_tile);
        issue_input(0, 0);
        if (nchunks == 1 && first_tile + (int)gridDim.x < a.ntiles) prep_tile(first_tile + gridDim.x);
        int it = 0;
        for (int tile = first_tile; tile < a.ntiles; tile += gridDim.x) {
            for (int kc = 0; kc < nchunks; ++kc, ++it) {
                // my pieces of item `it` have landed; after the barrier: everybody's have, and the compute waves are done
                // with item it-1, whose stage the next item may now overwrite
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                int ntile = tile, nkc = kc + 1;
                if (nkc == nchunks) { nkc = 0; ntile = tile + gridDim.x; }
                if (ntile < a.ntiles && !(a.dbg & 1)) {
                    issue_input(nkc, (it + 1) & 1);          // the prepared tile is ntile (see below)
                    if (!resident) issue_weights(nkc, (it + 1) & 1);
                    // ntile's last chunk is on its way: prepare its successor while the DMAs fly
                    if (nkc == nchunks - 1 && ntile + (int)gridDim.x < a.ntiles) prep_tile(ntile + gridDim.x);
                }
            }
        }
        if constexpr (STATS) {                           // the compute waves' reduction uses two more workgroup barriers
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        return;
    }

    // -------------------------------------------------- compute waves --------------------------------------------------
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y1 ? a.y1 : a.y), 0, a.y1 ? a.y1_bytes : 0, 0x00020000);
    // pixel tile pt of this wave sits 32/TW halo rows below tile pt-1: one base per (tx, g), the rest is an immediate
    constexpr int PT_STRIDE = (32 / TW) * HWD * 64;
    constexpr int TXN = TAPS == 9 ? 3 : 2;
    int in_base[TXN][2];
    {
        const int P = wv * (NPT * 32) + j;
        const int py = P / TW, px = P % TW;
#pragma unroll
        for (int tx = 0; tx < TXN; ++tx) {
            const int hx = px + tx + (TAPS == 4 ? pb : 0);
#pragma unroll
            for (int g = 0; g < 2; ++g) in_base[tx][g] = (py * HWD + hx) * 64 + (((2 * g + hf) ^ ((hx >> 2) & 3)) << 4);
        }
    }
    int w_addr[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int r = ct * 32 + j;
        w_addr[ct] = r * 64 + ((hf ^ ((r >> 2) & 3)) << 4);          // g = 0; g = 1 is this address ^ 32
    }
    f32x16 acc[NCT][NPT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
    // fused BatchNormalization statistics: two registers per channel tile - lane (j, hf) accumulates the sum and the sum
    // of squares of channel r = (j >> 1) & 15 of its accumulator quad set, over the 32 pixels of the half-wave and every
    // tile of the workgroup (lane_channel_sum below); works for NCT = 2 inside the 168 / 256 register budgets
    float st_sum[NCT], st_sq[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) st_sum[ct] = st_sq[ct] = 0.f;
    constexpr int MB_PLANE = TH * TW * 4, MB_TILE = NCT * MB_PLANE;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(STATS == 0 && a.sbits ? (void*)a.sbits : (void*)a.y), 0,
                                                                          STATS == 0 && a.sbits ? a.sbits_bytes : 0u, 0x00020000);
    int mb_par = 0;

    int it = 0;
    for (int tile = first_tile; tile < a.ntiles; tile += gridDim.x, mb_par ^= 1) {
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int n = bx / a.tiles_y;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        for (int kc = 0; kc < nchunks; ++kc, ++it) {
            asm volatile("s_barrier" ::: "memory");              // item `it` is in LDS (the loaders waited for their DMAs)
            if (a.dbg & 2) continue;
            const unsigned char* sin = lin + (it & 1) * IN_BYTES + (TAPS == 4 ? pa * HWD * 64 : 0);
            const unsigned char* sw = lw + (resident ? kc : (it & 1)) * W_BYTES;
            // 18 (tap, k-half) steps, two-deep software pipeline (see v2): reads of step i+1 above the MFMAs of step i
            uint4 fa[2][NCT], fb[2][NPT];
            auto load_step = [&](int st, int buf) __attribute__((always_inline)) {
                const int tap = st >> 1, g = st & 1;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) fa[buf][ct] = *reinterpret_cast<const uint4*>(sw + tap * BN * 64 + (w_addr[ct] ^ (g << 5)));
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) fb[buf][pt] = *reinterpret_cast<const uint4*>(sin + in_base[tap % TXN][g] + (tap / TXN) * HWD * 64 + pt * PT_STRIDE);
            };
            load_step(0, 0);
#pragma unroll
            for (int st = 0; st < 2 * TAPS; ++st) {
                const int cur = st & 1;
                if (st + 1 < 2 * TAPS) load_step(st + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) Mma<T>::run(fa[cur][ct], fb[cur][pt], acc[ct][pt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue of this tile; compute waves never wait on their stores (they issue no DMA)
        // store one (channel tile, pixel tile) of activated values: bf16 pairs are merged with v_permlane32_swap so that every
        // lane stores 16 contiguous bytes (8 channels of one pixel)
        auto store_tile = [&](int ct, const float (&v)[16], unsigned pix, bool pix_ok) __attribute__((always_inline)) {
            const int cbase = co0 + ct * 32;                                   // wave-uniform
            const bool second = a.y1 && cbase >= a.csplit;                     // csplit % 32 == 0 (host)
            const int cstride = a.y1 ? (second ? a.cout - a.csplit : a.csplit) : a.cout;
            const int cshift = second ? a.csplit : 0;
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = cbase + 8 * q + 4 * hf;
                    const unsigned off = (pix_ok && co < a.cout) ? (pix * cstride + (co - cshift)) * 4u : OOB;
                    const u32x4v dta = {__builtin_bit_cast(unsigned, v[4 * q]), __builtin_bit_cast(unsigned, v[4 * q + 1]),
                                        __builtin_bit_cast(unsigned, v[4 * q + 2]), __builtin_bit_cast(unsigned, v[4 * q + 3])};
                    if (second) __builtin_amdgcn_raw_buffer_store_b128(dta, ry1, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(dta, ry, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    unsigned ax = Vec<T>::pack2(v[8 * qq + 0], v[8 * qq + 1]);
                    unsigned ay = Vec<T>::pack2(v[8 * qq + 2], v[8 * qq + 3]);
                    unsigned bxx = Vec<T>::pack2(v[8 * qq + 4], v[8 * qq + 5]);
                    unsigned byy = Vec<T>::pack2(v[8 * qq + 6], v[8 * qq + 7]);
                    auto r0s = __builtin_amdgcn_permlane32_swap(ax, bxx, false, false);
                    auto r1s = __builtin_amdgcn_permlane32_swap(ay, byy, false, false);
                    const u32x4v dta = {r0s[0], r1s[0], r0s[1], r1s[1]};
                    const int co = cbase + 16 * qq + 8 * hf;
                    const unsigned off = (pix_ok && co < a.cout) ? (pix * cstride + (co - cshift)) * 2u : OOB;
                    if (second) __builtin_amdgcn_raw_buffer_store_b128(dta, ry1, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(dta, ry, off, 0, 0);
                }
            }
        };
        auto epilogue = [&](auto actf) {
            if constexpr (STATS) {                   // channel tile outermost: one set of 2 x 16 temporaries at a time
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    float qs[16], qq[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) qs[r] = qq[r] = 0.f;
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        const int P = wv * (NPT * 32) + pt * 32 + j;
                        const int gy = ty0 + P / TW, gx = tx0 + P % TW;
                        const bool pix_ok = gy < a.h && gx < a.w;
                        const unsigned pix = (unsigned)((n * a.h + gy) * a.w + gx);
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float t = acc[ct][pt][r] + lbias[ct * 32 + 8 * (r >> 2) + 4 * hf + (r & 3)];
                            acc[ct][pt][r] = 0.f;
                            v[r] = actf(t);
                        }
                        if constexpr (STATS == 3) {      // gate by the tile's mask words (Dropout / ReLU backward), wave-uniform per channel tile
                            if (co0 + ct * 32 < a.mbits_c) {
                                const unsigned word = *reinterpret_cast<const unsigned*>(smem + a.lds_mb_off + mb_par * MB_TILE + ct * MB_PLANE + P * 4);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    float u[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                                    gate4(u, (word >> (8 * (2 * (q & 1) + hf) + 4 * (q >> 1))) & 15u, a.mscale);      // channels 8q + 4hf + i: rvip_bit_of_channel
                                    v[4 * q] = u[0]; v[4 * q + 1] = u[1]; v[4 * q + 2] = u[2]; v[4 * q + 3] = u[3];
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float q = pix_ok ? Vec<T>::round(v[r]) : 0.f;     // statistics of what is stored
                            qs[r] += q;
                            if constexpr (STATS == 1) qq[r] = fmaf(q, q, qq[r]);
                        }
                        store_tile(ct, v, pix, pix_ok);
                    }
                    st_sum[ct] += lane_channel_sum(qs, j);
                    if constexpr (STATS == 1) st_sq[ct] += lane_channel_sum(qq, j);
                }
            } else {
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    const int P = wv * (NPT * 32) + pt * 32 + j;
                    const int gy = ty0 + P / TW, gx = tx0 + P % TW;
                    const bool pix_ok = gy < a.h && gx < a.w;
                    const unsigned pix = TAPS == 4 ? (unsigned)((n * 2 * a.h + 2 * gy + pa) * 2 * a.w + 2 * gx + pb)
                                                   : (unsigned)((n * a.h + gy) * a.w + gx);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float t = acc[ct][pt][r] + lbias[ct * 32 + 8 * (r >> 2) + 4 * hf + (r & 3)];
                            acc[ct][pt][r] = 0.f;
                            v[r] = actf(t);
                        }
                        store_tile(ct, v, pix, pix_ok);
                        if constexpr (STATS == 0) {
                            if (a.sbits) {                   // sign bits of what was stored: word = 32 channels of one pixel, plane = 32-channel block
                                unsigned word = 0;
#pragma unroll
                                for (int r = 0; r < 16; ++r) word |= (Vec<T>::round(v[r]) > 0.f ? 1u : 0u) << (8 * (2 * ((r >> 2) & 1) + hf) + 4 * (r >> 3) + (r & 3));
                                word |= (unsigned)__shfl_xor((int)word, 32);
                                const int cbase = co0 + ct * 32;
                                const unsigned off = (pix_ok && !hf && cbase < a.cout) ? ((unsigned)(cbase / 32) * (unsigned)((TAPS == 4 ? 4 : 1) * a.n * a.h * a.w) + pix) * 4u : OOB;
                                __builtin_amdgcn_raw_buffer_store_b32(word, rsb, off, 0, 0);
                            }
                        }
                    }
                }
            }
        };
        if (a.down2) epilogue_down2<T, TW, NCT, NPT, STATS>(acc, wv * (NPT * 32), j, hf, n, ty0, tx0, co0, a, ry, st_sum);
        else if constexpr (STATS >= 2) epilogue([](float t) { return t; });      // data gradients carry no activation (host-checked)
        else if (a.act == RVIP_ACT_RELU) epilogue([](float t) { return relu1(t); });
        else if (a.act == RVIP_ACT_NONE) epilogue([](float t) { return t; });
        else epilogue([&](float t) { return act_fwd(t, a.act); });
    }
    if constexpr (STATS) {
        asm volatile("s_barrier" ::: "memory");                            // every stage has been consumed
        constexpr int KS = STATS == 1 ? 2 : 1;
        float* lst = reinterpret_cast<float*>(smem);                       // [NCW compute waves][KS][BN]
        if (!(j & 1)) {                                                    // lanes j and j ^ 1 hold the same channel
            const int r = (j >> 1) & 15;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int c = ct * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                lst[(wv * KS + 0) * BN + c] = st_sum[ct];
                if constexpr (STATS == 1) lst[(wv * KS + 1) * BN + c] = st_sq[ct];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < KS * BN) {
            const int k = tid / BN, c = tid % BN;
            float t = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < NCW; ++w4) t += lst[(w4 * KS + k) * BN + c];
            if (co0 + c < a.cout) a.stats[((size_t)blockIdx.x * KS + k) * a.cout + co0 + c] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// igemm v5 (bf16 / f16): v4's producer / consumer structure on v_mfma_f32_16x16x32 instead of 32x32x16.  Same LDS stages,
// DMA pieces, barriers and FLOPs; MI355X_MICROARCH.md measures 1.12-1.15 x the FLOP/s for the 16x16x32 shape with LDS-fed
// operands (equal cycles, lower power, higher clock), and a timing mock-up of this kernel agreed (-9 % over the conv family).
// One tap of a 32-channel chunk is exactly one K = 32 step: lane (i16, kq) reads the 16-byte slot kq of row i16 of a
// 16-row block, so the 64-byte rows are read whole by four lanes.  Slot swizzle: physical slot = logical ^ 2 * bit2(row or
// halo x) -- conflict-free for ds_read_b128's four lane groups at every tap shift (searched exhaustively); the DMA pieces
// are written with the same mask.  Accumulators: f32x4 per (16-channel block, 16-pixel block), lane = pixel i16, registers =
// channels 4 kq + r.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_swz(int x) { return ((x >> 2) & 1) << 1; }
// sum over the 16 lanes of a row of a[r] (4 values per lane): lane i16 is left with the total of r = 2 bit3(i16) + bit2(i16)
__device__ __forceinline__ float lane16_channel_sum(const float (&a)[4], int i16) {
    const bool b3 = i16 & 8, b2 = i16 & 4;
    const float k0 = (b3 ? a[2] : a[0]) + swz<8>(b3 ? a[0] : a[2]);
    const float k1 = (b3 ? a[3] : a[1]) + swz<8>(b3 ? a[1] : a[3]);
    float e = (b2 ? k1 : k0) + swz<4>(b2 ? k0 : k1);
    e += swz<2>(e);
    return e + swz<1>(e);
}
// STATS: 0 none; 1 BatchNormalization statistics of the stored output (sum, sum of squares: forward launches); 2 column sums only,
// for data-gradient launches (with the 2x2-sum / channel-split epilogues): rows [gridDim.x][cout]; 3 = 2 + the Dropout backward
// (the body is a device function of the workgroup's coordinates so that the weight / data gradient pair kernel of rvip_pair.hip can
//  run it in a part of its grid: bx_ = first pixel tile, by_ = channel column, bz_ = output phase, gdx_ = the grid's tile stride)
template <typename T, int TW, int NCT, int NPIX, int STATS, int TAPS = 9, int NCW = 4>
__device__ __forceinline__ void igemm_ws16_body(const ConvArgs2& a, const int bx_, const int by_, const int bz_, const int gdx_) {
    static_assert(sizeof(T) == 2, "16-bit storage types");
    static_assert(TAPS == 9 || TAPS == 4, "taps");
    constexpr int TH = NPIX / TW, HWD = TW + 2, HHT = TH + 2, NHALO = HWD * HHT;
    constexpr int NHROWS = (NHALO + 15) / 16 * 16;
    constexpr int BN = NCT * 32, WROWS = TAPS * BN;
    constexpr int IN_BYTES = NHROWS * 64, W_BYTES = WROWS * 64;
    constexpr int VE = Vec<T>::VE, KCE = 4 * VE;
    constexpr int NQI = NHROWS / 16, NQW = WROWS / 16;                 // 1 KiB DMA pieces per stage
    constexpr int QI = (NQI + 3) / 4, QW = (NQW + 3) / 4;              // per loader wave
    constexpr int NPT = NPIX / (32 * NCW);                              // 32-pixel tiles per compute wave
    static_assert(NPT >= 1 && NPT * 32 * NCW == NPIX, "pixel tiles");
    static_assert(TW == 32 ? (NPT % 2 == 0) : true, "whole tile rows per wave");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lin = smem;                       // [nstg][IN_BYTES]
    unsigned char* lw = smem + a.nstg * IN_BYTES;    // [wres or 2][W_BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = by_ * BN;
    const int ph = (TAPS == 4) ? (int)bz_ : 0, pa = ph >> 1, pb = ph & 1;      // output phase of the sub-pixel form
    // subpix == 2 (TAPS = 4, gridDim.z = 1): the DATA GRADIENT of the sub-pixel form.  x0 is the full-resolution gradient [N, 2h, 2w, c0],
    // the result the low-resolution [N, h, w, cout]; the K loop runs over kd = 4 source phases (al, be) x the channel chunks: chunk
    // (al, be, c) reads x0[2y + al][2x + be] (an addressing mode of the loaders) through the 2x2 window at halo origin (1 - al, 1 - be)
    // with the summed taps of rvip_pack_subpixel_dgrad_weights -- 16 instead of 36 multiply-adds per low-resolution pixel, no 2x2-sum epilogue.
    const bool s2d = TAPS == 4 && a.subpix == 2;
    const int nch = (a.cin + KCE - 1) / KCE;         // chunks per depth tap
    const int nchunks = nch * a.kd;
    const bool resident = a.wres > 0;
    const int first_tile = bx_;
    if (first_tile >= a.ntiles) return;

    if (wv >= NCW) {
        // ------------------------------------------------ loader waves ------------------------------------------------
        __builtin_amdgcn_s_setprio(3);                       // their few instructions go ahead of the compute waves' streams
        const int lwv = wv - NCW;
        const int h0 = a.h >> a.up0, w0 = a.w >> a.up0;
        const i32x4 rs0 = make_rsrc(a.x0, a.x0_bytes);
        const i32x4 rs1 = make_rsrc(a.x1 ? a.x1 : a.x0, a.x1 ? a.x1_bytes : 0u);
        const i32x4 rsw = make_rsrc(a.wp, a.wp_bytes);
        const unsigned lds_base = lds_offset_of(smem);
        const int drow = lane >> 2, dslot = lane & 3;
        int wrel[QW], wch[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const int row = (lwv + 4 * i) * 16 + drow;
            const int tap = row / BN, co = co0 + (row & (BN - 1));
            wch[i] = (co < a.cout && row < WROWS) ? (dslot ^ slot_swz(row)) * VE : 1 << 28;
            wrel[i] = (((tap * a.cout + co) * a.cin) + (dslot ^ slot_swz(row)) * VE) * (int)sizeof(T);
        }
        // The loaders' own instructions are the critical path of staging (profiles/r01_conv_ablation.txt: one K chunk in flight
        // per workgroup, and every instruction of these waves is issued in the shadow of two MFMA-bound waves), so whatever does
        // not depend on the K chunk is hoisted: per wave the tile-independent byte offsets of its pieces, per TILE the offsets
        // with the image-border test folded in (t0 / t1: offset inside the tile's window of source 0 / 1, or OOB), computed in
        // the idle window after the tile's predecessor has issued its last chunk.  A full chunk then costs one v_add per piece.
        unsigned wfast[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) wfast[i] = (wch[i] != (1 << 28)) ? (unsigned)wrel[i] : OOB;
        int ihy[QI], ihx[QI], irel0[QI], irel1[QI], ich[QI];
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int row = (lwv + 4 * i) * 16 + drow;
            const int hy = row / HWD, hx = row - hy * HWD;
            ich[i] = (dslot ^ slot_swz(hx)) * VE;
            ihy[i] = (row < NHALO) ? hy - 1 : -100000;
            ihx[i] = hx - 1;
            irel0[i] = s2d ? ((2 * (hy - 1) * 2 * a.w + 2 * (hx - 1)) * a.c0 + ich[i]) * (int)sizeof(T)
                           : ((((hy - 1) >> a.up0) * w0 + ((hx - 1) >> a.up0)) * a.c0 + ich[i]) * (int)sizeof(T);
            irel1[i] = (((hy - 1) * a.w + (hx - 1)) * a.c1 + ich[i]) * (int)sizeof(T);
        }
        unsigned t0[QI], t1[QI];                         // prepared tile: per-piece offsets (border test folded in)
        int p_nmod = 0, p_tb0 = 0, p_tb1 = 0;
        const int img0_bytes = h0 * w0 * a.c0 * (int)sizeof(T), img1_bytes = a.h * a.w * a.c1 * (int)sizeof(T);
        // STATS == 3: the mask words of the tile's OUTPUT pixels, one 32-channel plane per channel tile, [TH][TW] words per plane, are
        // one 1 KiB piece per loader wave (NCT * NPIX / 256 <= 4 pieces), double-buffered by tile parity; issued with chunk 0 of the tile
        constexpr int MB_PLANE = TH * TW * 4, NPM = MB_PLANE / 1024, MB_TILE = NCT * MB_PLANE, MB_LPR = TW / 4;
        const bool mb_mine = STATS == 3 && lwv < NCT * NPM && co0 + (lwv / NPM) * 32 < a.mbits_c;
        const int mb_row = (lwv % NPM) * (64 / MB_LPR) + lane / MB_LPR, mb_x = (lane % MB_LPR) * 4;
        const i32x4 rsm = make_rsrc(STATS == 3 ? (const void*)a.mbits : (const void*)a.x0, STATS == 3 ? a.mbits_bytes : 0u);
        unsigned p_mb = OOB;
        int mb_par = 0;
        auto prep_tile = [&](int tile) __attribute__((always_inline)) {
            unsigned bx = (unsigned)tile;
            const unsigned tx_i = bx % (unsigned)a.tiles_x; bx /= (unsigned)a.tiles_x;
            const unsigned ty_i = bx % (unsigned)a.tiles_y;
            const int n_out = (int)(bx / (unsigned)a.tiles_y);
            const int ty0 = (int)ty_i * TH, tx0 = (int)tx_i * TW;
            p_nmod = a.depth > 1 ? n_out % a.depth : 0;
            p_tb0 = s2d ? ((n_out * 2 * a.h + 2 * ty0) * 2 * a.w + 2 * tx0) * a.c0 * (int)sizeof(T)
                        : ((n_out * h0 + (ty0 >> a.up0)) * w0 + (tx0 >> a.up0)) * a.c0 * (int)sizeof(T);
            p_tb1 = ((n_out * a.h + ty0) * a.w + tx0) * a.c1 * (int)sizeof(T);
            if constexpr (STATS == 3) {
                const bool ok = mb_mine && ty0 + mb_row < a.h && tx0 + mb_x < a.w;
                p_mb = ok ? (unsigned)(((co0 / 32 + lwv / NPM) * a.n * a.h + n_out * a.h + ty0 + mb_row) * a.w + tx0 + mb_x) * 4u : OOB;
            }
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int gy = ty0 + ihy[i], gx = tx0 + ihx[i];
                const bool ok = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w && !(a.dbg & 4);
                const bool ok0 = a.zs ? (ok && ((gy & gx) & 1)) : ok;
                t0[i] = ok0 ? (unsigned)irel0[i] : OOB;
                t1[i] = ok ? (unsigned)irel1[i] : OOB;
            }
        };
        auto issue_weights = [&](int kc, int wstage) __attribute__((always_inline)) {
            const int kdi = a.kd > 1 ? kc / nch : 0;
            const int cbase = (kc - kdi * nch) * KCE;
            const int tapbase = ((kdi + ph) * TAPS * a.cout * a.cin + cbase) * (int)sizeof(T);
            const unsigned lw0 = lds_base + a.nstg * IN_BYTES + wstage * W_BYTES + lwv * 1024; // + i * 4096 per piece
            if (a.cin - cbase >= KCE) {                  // full chunk: valid rows carry all their channels
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    if (i < QW - 1 || lwv + 4 * i < NQW) dma16(rsw, wfast[i] + (unsigned)tapbase, lw0 + i * 4096);   // OOB + tapbase stays >= 2^31
                }
            } else {
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    if (i < QW - 1 || lwv + 4 * i < NQW) {
                        const unsigned off = (wch[i] < a.cin - cbase) ? (unsigned)(wrel[i] + tapbase) : OOB;
                        dma16(rsw, off, lw0 + i * 4096);
                    }
                }
            }
        };
        // stages chunk kc of the PREPARED tile
        auto issue_input = [&](int kc, int stage) __attribute__((always_inline)) -> int {   // returns the VMEM instructions this wave issued
            int nvm = 0;
            const int kdi = a.kd > 1 ? kc / nch : 0;
            const int dsh = s2d ? 0 : kdi - (a.kd >> 1);     // depth tap: the image dsh slices away, zeros outside the volume
            const bool dok = (unsigned)(p_nmod + dsh) < (unsigned)a.depth;
            const int phoff = s2d ? ((kdi >> 1) * 2 * a.w + (kdi & 1)) * a.c0 * (int)sizeof(T) : 0;      // source phase (al, be) = (kdi >> 1, kdi & 1)
            const int cbase = (kc - kdi * nch) * KCE;
            const bool from0 = cbase < a.c0;                 // chunks never straddle the two sources (host checks)
            const int cb = from0 ? cbase : cbase - a.c0;
            const int crem = dok ? (from0 ? a.c0 : a.c1) - cb : 0;
            const unsigned li0 = lds_base + stage * IN_BYTES + lwv * 1024;                      // + i * 4096 per piece
            const unsigned base = (unsigned)((from0 ? p_tb0 + dsh * img0_bytes : p_tb1 + dsh * img1_bytes) + cb * (int)sizeof(T) + phoff);
            if constexpr (STATS == 3) {
                if (kc == 0) {                               // first chunk of a tile: its mask words ride along (wave-uniform)
                    if (lwv < NCT * NPM) { dma16(rsm, p_mb, lds_base + a.lds_mb_off + mb_par * MB_TILE + lwv * 1024); ++nvm; }
                    mb_par = mb_par + 1 == a.nstg ? 0 : mb_par + 1;      // one buffer of mask words per input stage
                }
            }
            if (crem >= KCE && !a.nt_in) {                   // full chunk inside the volume: offset = prepared + base
                if (from0) {
#pragma unroll
                    for (int i = 0; i < QI; ++i) {
                        if (i < QI - 1 || lwv + 4 * i < NQI) { dma16(rs0, t0[i] + base, li0 + i * 4096); ++nvm; }   // OOB + base stays >= 2^31
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < QI; ++i) {
                        if (i < QI - 1 || lwv + 4 * i < NQI) { dma16(rs1, t1[i] + base, li0 + i * 4096); ++nvm; }
                    }
                }
            } else {                                         // partial chunk / depth tap outside the volume / streaming hint
                const i32x4 rs = from0 ? rs0 : rs1;
#pragma unroll
                for (int i = 0; i < QI; ++i) {
                    if (i < QI - 1 || lwv + 4 * i < NQI) {
                        const unsigned t = from0 ? t0[i] : t1[i];
                        const unsigned off = (t != OOB && ich[i] < crem) ? t + base : OOB;
                        if (a.nt_in) dma16_nt(rs, off, li0 + i * 4096);
                        else dma16(rs, off, li0 + i * 4096);
                        ++nvm;
                    }
                }
            }
            return nvm;
        };
        if (resident) for (int kc = 0; kc < nchunks; ++kc) issue_weights(kc, kc);
        else issue_weights(0, 0);
        // The loaders run nstg - 1 items ahead of the compute waves.  Two stages (the original form): the next item is requested behind
        // the barrier that frees its stage and must have landed one item later -- a one-chunk tile then has ONE tile's bytes in flight
        // per CU and its compute waves wait for the DMA at every barrier.  Three stages (weights resident, LDS permitting): the item
        // requested here is consumed TWO barriers later, and the wait in front of a barrier is a COUNTED one -- s_waitcnt vmcnt(n) with
        // n = this wave's DMA instructions of the younger item, which may stay in flight (loads return in order).
        int itile = first_tile, ikc = 0, istg = 0;
        prep_tile(itile);
        auto issue_next = [&]() __attribute__((always_inline)) -> int {     // requests the cursor's item; returns this wave's DMA count for it (0: none)
            if (itile >= a.ntiles) return 0;
            // The count IS what was issued (every VMEM instruction of an item is counted where it is issued): the counted wait below
            // cannot drift from the request code.  Three stages imply resident weights (host), so the rotating-weight DMAs -- issued
            // only with two stages, where every wait is vmcnt(0) -- are not part of it.
            int cnt = 0;
            if (!resident && (a.dbg & 64)) issue_weights(ikc, istg);          // (ablation: weights first)
            if (!(a.dbg & 16)) cnt = issue_input(ikc, istg);
            if (!resident && !(a.dbg & (8 | 64))) issue_weights(ikc, istg);   // (rotating weights: two stages only, host-checked)
            if (++ikc == nchunks) {
                ikc = 0;
                itile += gdx_;
                if (itile < a.ntiles) prep_tile(itile);                        // its offsets, while the DMAs fly
            }
            istg = istg + 1 == a.nstg ? 0 : istg + 1;
            return cnt;
        };
        issue_next();
        if (a.dbg & 1) itile = a.ntiles;                  // (ablation: nothing is requested after the first item)
        int cyoung = a.nstg == 3 ? issue_next() : 0;      // DMAs of the item BEHIND the one the next barrier hands over
        for (int tile = first_tile; tile < a.ntiles; tile += gdx_) {
            for (int kc = 0; kc < nchunks; ++kc) {
                // my pieces of the item about to be consumed have landed; after the barrier: everybody's have, and the compute waves
                // are done with the item before it, whose stage the next request may now overwrite
                // cyoung = VMEM instructions of the YOUNGER item that may stay in flight; any count the cases below do not name waits for all
                if (cyoung == QI + 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(QI + 1) : "memory");
                else if (cyoung == QI) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(QI) : "memory");
                else if (QI > 1 && cyoung == QI - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(QI > 1 ? QI - 1 : 0) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                const int c = issue_next();
                if (a.nstg == 3) cyoung = c;
            }
        }
        if constexpr (STATS) {                           // the compute waves' reduction uses two more workgroup barriers
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        return;
    }

    // -------------------------------------------------- compute waves --------------------------------------------------
    // Fragment geometry of v_mfma_f32_16x16x32: lane = (i16 = row of the 16-row operand block, kq = which 16-byte slot of
    // the 64-byte K row); the 4 accumulator registers of a 16 x 16 tile hold output channels 4 kq + r of pixel i16.
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y1 ? a.y1 : a.y), 0, a.y1 ? a.y1_bytes : 0, 0x00020000);
    const int i16 = lane & 15, kq = lane >> 4;
    constexpr int NCB = 2 * NCT, NPB = 2 * NPT, HB = NPB / 2;           // 16-channel / 16-pixel blocks per wave; blocks per half step
    constexpr int BPR = TW / 16;                                          // pixel blocks per tile row
    constexpr int TXN = TAPS == 9 ? 3 : 2;
    const int row0 = (wv * (NPT * 32)) / TW;                              // first tile row of this wave
    // one address register per tap column / for the weights: a second pixel block of a tile row is 16 halo columns (1 KiB) further,
    // a further channel block 16 rows (1 KiB) further, and neither moves bit 2 of the swizzle key -- compile-time offsets
    int in_base3[3];                                                      // halo column i16 + 0 / 1 / 2
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
        const int hx = i16 + tx;
        in_base3[tx] = (row0 * HWD + hx) * 64 + ((kq ^ slot_swz(hx)) << 4);
    }
    // TAPS = 4: the window's first column is halo column pb (forward phase, per workgroup) / 1 - be (data gradient, per chunk)
    int in_base[TXN];
#pragma unroll
    for (int tx = 0; tx < TXN; ++tx) {
        const int hx = i16 + tx + (TAPS == 4 ? pb : 0);
        in_base[tx] = (row0 * HWD + hx) * 64 + ((kq ^ slot_swz(hx)) << 4);
    }
    const int w_addr = i16 * 64 + ((kq ^ slot_swz(i16)) << 4);
    // The accumulators START at the bias (and return to it in the epilogue) instead of having it added to every value there: the
    // epilogue is VALU-issue bound (two waves per SIMD), every instruction it loses shortens the tile.  Zero for the 2x2-sum form.
    f32x4 bias_r[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int c = co0 + cb * 16 + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[cb][r] = (a.bias && !a.down2 && c + r < a.cout) ? a.bias[c + r] : 0.f;
    }
    f32x4 acc[NCB][NPB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int p = 0; p < NPB; ++p) acc[cb][p] = bias_r[cb];
    // fused BatchNormalization statistics: lane (kq, i16) ends up with the sums of channel 4 kq + 2 bit3(i16) + bit2(i16) of every
    // 16-channel block over the wave's pixels (lane16_channel_sum), two registers per block
    float st_sum[NCB], st_sq[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) st_sum[cb] = st_sq[cb] = 0.f;
    constexpr int MB_PLANE = TH * TW * 4, MB_TILE = NCT * MB_PLANE;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(STATS == 0 && a.sbits ? (void*)a.sbits : (void*)a.y), 0,
                                                                          STATS == 0 && a.sbits ? a.sbits_bytes : 0u, 0x00020000);
    int mb_par = 0;

    int it = 0, stg = 0;
    for (int tile = first_tile; tile < a.ntiles; tile += gdx_, mb_par = (mb_par + 1 == a.nstg ? 0 : mb_par + 1)) {
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int n = bx / a.tiles_y;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        for (int kc = 0; kc < nchunks; ++kc, ++it) {
            asm volatile("s_barrier" ::: "memory");              // item `it` is in LDS (the loaders waited for their DMAs)
            const int cur = stg;
            stg = stg + 1 == a.nstg ? 0 : stg + 1;
            if (a.dbg & 2) continue;
            const unsigned char* sin = lin + cur * IN_BYTES + (TAPS == 4 ? pa * HWD * 64 : 0);
            if constexpr (TAPS == 4) {
                if (s2d) {                                           // source phase of this chunk -> window origin (wave-uniform)
                    const int kdi = kc / nch;
                    sin = lin + cur * IN_BYTES + (1 - (kdi >> 1)) * HWD * 64;
                    const bool c1st = (kdi & 1) != 0;                // be = 1: window starts at halo column 0
                    in_base[0] = c1st ? in_base3[0] : in_base3[1];
                    in_base[1] = c1st ? in_base3[1] : in_base3[2];
                }
            }
            const unsigned char* sw = lw + (resident ? kc : (it & 1)) * W_BYTES;
            // one tap = one K = 32 step, issued as two half steps (all channel blocks x half of the pixel blocks); the weight
            // fragments of the next tap and the pixel fragments of the next half step are read above the MFMAs of this one
            uint4 fa[2][NCB], fb[2][HB];
            auto load_a = [&](int tap, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) fa[buf][cb] = *reinterpret_cast<const uint4*>(sw + tap * BN * 64 + cb * 1024 + w_addr);
            };
            auto load_b = [&](int tap, int half, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int p = 0; p < HB; ++p) {
                    const int blk = half * HB + p;
                    fb[buf][p] = *reinterpret_cast<const uint4*>(sin + in_base[tap % TXN] + (blk % BPR) * 1024 + (tap / TXN + blk / BPR) * HWD * 64);
                }
            };
            load_a(0, 0);
            load_b(0, 0, 0);
#pragma unroll
            for (int ss = 0; ss < 2 * TAPS; ++ss) {
                const int tap = ss >> 1, half = ss & 1;
                if (half == 0) load_b(tap, 1, 1);
                else if (tap + 1 < TAPS) { load_a(tap + 1, (tap + 1) & 1); load_b(tap + 1, 0, 0); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int p = 0; p < HB; ++p) acc[cb][half * HB + p] = mfma16x16<T>(fa[tap & 1][cb], fb[half][p], acc[cb][half * HB + p]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue of this tile; compute waves never wait on their stores ----
        // Two CHANNEL blocks (A, B = 2 cp, 2 cp + 1) of one pixel block at a time: after v_permlane16_swap of the packed pairs a lane
        // with even kq holds 8 consecutive channels of block A, a lane with odd kq 8 of block B, all of the lane's own pixel -> one
        // 16-byte store per lane, and the four lanes of a pixel cover 32 consecutive channels = 64 contiguous bytes (a whole pixel
        // row for Cout = 32, where a store instruction is then 1 KiB contiguous).  Pairing two PIXEL blocks of one channel block
        // (the first form) left every instruction writing 32-byte pieces 64 / 128 bytes apart.
        // a0, a1 / b0, b1: the four channels of block A / B of the lane's pixel, packed (Vec<T>::pack2)
        auto store_cbpair = [&](int cp, unsigned a0, unsigned a1, unsigned b0, unsigned b1, unsigned pix, bool pix_ok) __attribute__((always_inline)) {
            const int cbase = co0 + cp * 32;                                   // wave-uniform
            const bool second = a.y1 && cbase >= a.csplit;                     // csplit % 32 == 0 (host)
            const int cstride = a.y1 ? (second ? a.cout - a.csplit : a.csplit) : a.cout;
            const int cshift = second ? a.csplit : 0;
            auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            const u32x4v dta = {s0[0], s1[0], s0[1], s1[1]};
            const int co = cbase + 16 * (kq & 1) + 8 * (kq >> 1);
            const unsigned off = (pix_ok && co < a.cout) ? (pix * cstride + (co - cshift)) * 2u : OOB;
            if (second) __builtin_amdgcn_raw_buffer_store_b128(dta, ry1, off, 0, RVIP_WT_AUX);
            else __builtin_amdgcn_raw_buffer_store_b128(dta, ry, off, 0, RVIP_WT_AUX);
        };
        auto block_pixel = [&](int blk, int& gy, int& gx) __attribute__((always_inline)) {
            gy = ty0 + row0 + blk / BPR;
            gx = tx0 + (blk % BPR) * 16 + i16;
        };
        auto epilogue = [&](auto actf) {
#pragma unroll
            for (int cp = 0; cp < NCB / 2; ++cp) {
                float qs[2][4], qq[2][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) qs[0][r] = qs[1][r] = qq[0][r] = qq[1][r] = 0.f;
                // column sums (STATS == 2): of the fp32 values in front of the storage rounding, only for the channel pairs somebody
                // reads (sums_from: the first half of a split result belongs to a stage without BatchNormalization), and without the
                // per-pixel validity select when the whole tile lies inside the image -- all wave-uniform
                const bool need_sums = STATS >= 2 && co0 + cp * 32 + 32 > a.sums_from;
                const bool interior = ty0 + TH <= a.h && tx0 + TW <= a.w;
                const bool gated = STATS == 3 && co0 + cp * 32 < a.mbits_c;      // this channel pair is gated by the tile's mask words
#pragma unroll
                for (int q = 0; q < NPB / 2; ++q) {
                    unsigned pk[2][2][2];                                       // [pixel block of the pair][channel block A / B][word]
                    bool ok[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int blk = 2 * q + s2;
                        int gy, gx;
                        block_pixel(blk, gy, gx);
                        ok[s2] = gy < a.h && gx < a.w;
                        const unsigned pix = (TAPS == 4 && !s2d) ? (unsigned)((n * 2 * a.h + 2 * gy + pa) * 2 * a.w + 2 * gx + pb)
                                                                 : (unsigned)((n * a.h + gy) * a.w + gx);
                        unsigned mword = 0, sword = 0;
                        if constexpr (STATS == 3) {
                            if (gated) mword = *reinterpret_cast<const unsigned*>(smem + a.lds_mb_off + mb_par * MB_TILE + cp * MB_PLANE +
                                                                                  ((row0 + blk / BPR) * TW + (blk % BPR) * 16 + i16) * 4);
                        }
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            const int cb = 2 * cp + c2;
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) { v[r] = actf(acc[cb][blk][r]); acc[cb][blk][r] = bias_r[cb][r]; }
                            if constexpr (STATS == 3) {      // Dropout / ReLU backward on the result: four bits of the pixel's mask word
                                if (gated) gate4(v, (mword >> (8 * kq + 4 * c2)) & 15u, a.mscale);
                            }
                            if constexpr (STATS == 0) {      // sign bits of what is stored (forward of a stage whose ReLU backward will be gated)
                                if (a.sbits) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) sword |= (Vec<T>::round(v[r]) > 0.f ? 1u : 0u) << (4 * c2 + r);
                                }
                            }
                            if constexpr (STATS >= 2) {
                                if (need_sums) {
                                    if (interior) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) qs[c2][r] += v[r];
                                    } else {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) qs[c2][r] += ok[s2] ? v[r] : 0.f;
                                    }
                                }
                            }
                            pk[s2][c2][0] = Vec<T>::pack2(v[0], v[1]);
                            pk[s2][c2][1] = Vec<T>::pack2(v[2], v[3]);
                        }
                        store_cbpair(cp, pk[s2][0][0], pk[s2][0][1], pk[s2][1][0], pk[s2][1][1], pix, ok[s2]);
                        if constexpr (STATS == 0) {
                            if (a.sbits) {                   // the four kq lanes of a pixel hold one byte each of its 32-channel word (rvip_bit_of_channel)
                                const int cbase = co0 + cp * 32;
                                const unsigned off = (ok[s2] && cbase < a.cout) ? ((unsigned)(cbase / 32) * (unsigned)((TAPS == 4 ? 4 : 1) * a.n * a.h * a.w) + pix) * 4u + (unsigned)kq : OOB;
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sword, rsb, off, 0, 0);
                            }
                        }
                    }
                    if constexpr (STATS == 1) {                                 // statistics of what is stored: the packed words, widened
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            // a pixel outside the image counts as zero: selected on the packed words (+0 is all-zero bits), one select per channel pair
                            const unsigned x0 = ok[0] ? pk[0][c2][0] : 0u, x1 = ok[0] ? pk[0][c2][1] : 0u;
                            const unsigned y0 = ok[1] ? pk[1][c2][0] : 0u, y1 = ok[1] ? pk[1][c2][1] : 0u;
                            const float ux[4] = {Vec<T>::lo(x0), Vec<T>::hi(x0), Vec<T>::lo(x1), Vec<T>::hi(x1)};
                            const float uy[4] = {Vec<T>::lo(y0), Vec<T>::hi(y0), Vec<T>::lo(y1), Vec<T>::hi(y1)};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float u0 = ux[r], u1 = uy[r];
                                qs[c2][r] += u0 + u1;
                                qq[c2][r] = fmaf(u0, u0, fmaf(u1, u1, qq[c2][r]));
                            }
                        }
                    }
                }
                if constexpr (STATS == 1) {
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        st_sum[2 * cp + c2] += lane16_channel_sum(qs[c2], i16);
                        st_sq[2 * cp + c2] += lane16_channel_sum(qq[c2], i16);
                    }
                }
                if constexpr (STATS >= 2) {
                    if (need_sums) {
                        st_sum[2 * cp] += lane16_channel_sum(qs[0], i16);
                        st_sum[2 * cp + 1] += lane16_channel_sum(qs[1], i16);
                    }
                }
            }
        };
        // data gradient of UpSampling2D -> conv: 2 x 2 block sums.  The block's rows are two pixel blocks of the same lane, its
        // columns lanes i16 and i16 ^ 1; units (block sums) are paired for the 16-byte store like the pixel blocks above.
        auto epilogue_down2_16 = [&]() {
            constexpr int NU = NPB / 2;                                        // 2-row units per wave
            const int hl = a.h >> 1, wl = a.w >> 1;
#pragma unroll
            for (int cp = 0; cp < NCB / 2; ++cp) {
                float qd[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    // TW = 32: unit = (row pair m, column block cx); TW = 16: unit = row pair m
                    const int m = BPR == 2 ? (u >> 1) : u, cx = BPR == 2 ? (u & 1) : 0;
                    const int top = (2 * m) * BPR + cx, bot = (2 * m + 1) * BPR + cx;
                    const int gyu = ty0 + row0 + 2 * m, gxu = tx0 + cx * 16 + i16;
                    float v[2][4];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const int cb = 2 * cp + c2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t = acc[cb][top][r] + acc[cb][bot][r];
                            acc[cb][top][r] = 0.f;
                            acc[cb][bot][r] = 0.f;
                            v[c2][r] = t + lane_xor1(t);
                        }
                    }
                    const bool keep = gyu < a.h && gxu < a.w && !(i16 & 1);
                    const unsigned pix = (unsigned)((n * hl + (gyu >> 1)) * wl + (gxu >> 1));
                    const unsigned w00 = Vec<T>::pack2(v[0][0], v[0][1]), w01 = Vec<T>::pack2(v[0][2], v[0][3]);
                    const unsigned w10 = Vec<T>::pack2(v[1][0], v[1][1]), w11 = Vec<T>::pack2(v[1][2], v[1][3]);
                    store_cbpair(cp, w00, w01, w10, w11, pix, keep);
                    if constexpr (STATS >= 2) {                                // column sums of the block sums (fp32, in front of the rounding)
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                            for (int r = 0; r < 4; ++r) qd[c2][r] += keep ? v[c2][r] : 0.f;
                    }
                }
                if constexpr (STATS >= 2) {
                    st_sum[2 * cp] += lane16_channel_sum(qd[0], i16);
                    st_sum[2 * cp + 1] += lane16_channel_sum(qd[1], i16);
                }
            }
        };
        if (a.dbg & 32) {}                                                       // (ablation: no epilogue)
        else if (a.down2) epilogue_down2_16();
        else if constexpr (STATS >= 2) epilogue([](float t) { return t; });      // data gradients carry no activation (host-checked)
        else if (a.act == RVIP_ACT_RELU) epilogue([](float t) { return relu1(t); });
        else if (a.act == RVIP_ACT_NONE) epilogue([](float t) { return t; });
        else epilogue([&](float t) { return act_fwd(t, a.act); });
    }
    if constexpr (STATS) {
        asm volatile("s_barrier" ::: "memory");                            // every stage has been consumed
        constexpr int KS = STATS == 1 ? 2 : 1;
        float* lst = reinterpret_cast<float*>(smem);                       // [NCW compute waves][KS][BN]
        if (!(i16 & 3)) {                                                  // four lanes hold the same channel
            const int r = 2 * ((i16 >> 3) & 1) + ((i16 >> 2) & 1);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int c = cb * 16 + 4 * kq + r;
                lst[(wv * KS + 0) * BN + c] = st_sum[cb];
                if constexpr (STATS == 1) lst[(wv * KS + 1) * BN + c] = st_sq[cb];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < KS * BN) {
            const int k = tid / BN, c = tid % BN;
            float t = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < NCW; ++w4) t += lst[(w4 * KS + k) * BN + c];
            if (co0 + c < a.cout) a.stats[((size_t)bx_ * KS + k) * a.cout + co0 + c] = t;
        }
    }
}

template <typename T, int TW, int NCT, int NPIX, int STATS, int TAPS = 9, int NCW = 4>
__global__ __launch_bounds__((NCW + 4) * 64, 1) void conv3x3_igemm_ws16(ConvArgs2 a) {
    igemm_ws16_body<T, TW, NCT, NPIX, STATS, TAPS, NCW>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x);
}

template <typename T, bool V5, int TW, int NCT, int NPIX, int STATS, int TAPS, int NCW>
static constexpr auto igemm_ws_kernel() {
    // (the four-compute-wave 512-pixel tiling holds 128 accumulators per lane: the wider fragment set of v5 would spill there)
    if constexpr (V5 && sizeof(T) == 2 && (NPIX / (32 * NCW) <= 2 || (TAPS == 9 && NCW == 4))) return &conv3x3_igemm_ws16<T, TW, NCT, NPIX, STATS, TAPS, NCW>;
    else return &conv3x3_igemm_ws<T, TW, NCT, NPIX, STATS, TAPS, NCW>;
}

// what launch_igemm_ws would launch (PLAN_ONLY: filled instead of launching -- the pair kernel of rvip_pair.hip runs the body itself)
struct IgemmPlan { ConvArgs2 args; int gx, gy, gz, lds; };

template <typename T, bool V5, int TW, int NCT, int NPIX, int TAPS = 9, int NCW = 4, bool PLAN_ONLY = false>
static int launch_igemm_ws(const ConvArgs& a0, hipStream_t s, bool& used, float* stats, int* rows_out, bool dry, int smode = 1, IgemmPlan* plan = nullptr) {
    constexpr int TH = NPIX / TW;
    constexpr int NHROWS = ((TW + 2) * (TH + 2) + 15) / 16 * 16;
    constexpr int IN_BYTES = NHROWS * 64, W_BYTES = TAPS * NCT * 32 * 64;
    constexpr int KCE = 64 / (int)sizeof(T);
    constexpr int LDS_MAX = 160 * 1024;
    used = false;
    const int nchunks = (int)cdiv(a0.cin, KCE) * a0.kd;
    const long long x0b = (a0.subpix == 2 ? 4LL : 1LL) * a0.n * (a0.h >> a0.up0) * (a0.w >> a0.up0) * a0.c0 * (long long)sizeof(T);
    const long long x1b = (long long)a0.n * a0.h * a0.w * a0.c1 * (long long)sizeof(T);
    const long long wpb = (TAPS == 4 ? 16LL : 9LL * a0.kd) * a0.cin * a0.cout * (long long)sizeof(T);
    if ((TAPS == 4) != (a0.subpix != 0)) return RVIP_OK;
    constexpr bool WS16 = V5 && sizeof(T) == 2 && (NPIX / (32 * NCW) <= 2 || (TAPS == 9 && NCW == 4));      // (igemm_ws_kernel's choice)
    if (a0.subpix == 2 && !WS16) return RVIP_OK;           // the data-gradient form of the sub-pixel up-conv lives in the 16-bit kernel only
    const bool sp_fwd = TAPS == 4 && a0.subpix == 1;        // forward form: four output phases = blockIdx.z, result at twice the grid
    if (x0b >= (1LL << 31) || x1b >= (1LL << 31) || wpb >= (1LL << 31)) return RVIP_OK;
    if (a0.c1 > 0 && a0.c0 % KCE) return RVIP_OK;
    if (a0.y1 && a0.csplit % 32) return RVIP_OK;
    const long long npx = (long long)a0.n * a0.h * a0.w;
    const long long yb = (a0.down2 ? npx / 4 : (sp_fwd ? npx * 4 : npx)) * (a0.y1 ? a0.csplit : a0.cout) * (long long)sizeof(T), y1b = a0.y1 ? npx * (a0.cout - a0.csplit) * (long long)sizeof(T) : 0;
    if (yb >= (1LL << 31) || y1b >= (1LL << 31)) return RVIP_OK;
    if (sizeof(T) == 2 && a0.cout % 8) return RVIP_OK;
    ConvArgs2 b;
    b.x0 = a0.x0; b.x1 = a0.x1; b.wp = a0.wp; b.bias = a0.bias; b.y = a0.y; b.y1 = a0.y1;
    b.x0_bytes = (unsigned)x0b; b.x1_bytes = (unsigned)x1b; b.wp_bytes = (unsigned)wpb; b.y_bytes = (unsigned)yb; b.y1_bytes = (unsigned)y1b;
    b.c0 = a0.c0; b.c1 = a0.c1; b.up0 = a0.up0; b.csplit = a0.csplit; b.zs = a0.zs; b.depth = a0.depth; b.kd = a0.kd; b.down2 = a0.down2; b.subpix = a0.subpix; b.nt_in = a0.nt_in;
    b.n = a0.n; b.h = a0.h; b.w = a0.w; b.cin = a0.cin; b.cout = a0.cout; b.act = a0.act;
    b.sums_from = a0.sums_from;
    const bool gated = stats && smode == 2 && a0.mbits;
    b.mbits = gated ? a0.mbits : nullptr; b.mbits_c = gated ? a0.mbits_c : 0; b.mscale = a0.mscale;
    b.mbits_bytes = gated ? (unsigned)((long long)cdiv(a0.mbits_c, 32) * npx * 4) : 0u;
    b.sbits = (!stats && a0.sbits) ? a0.sbits : nullptr;
    b.sbits_bytes = b.sbits ? (unsigned)((long long)cdiv(a0.cout, 32) * (sp_fwd ? npx * 4 : npx) * 4) : 0u;
    if ((gated || b.sbits) && (a0.cout % 8 || a0.down2 || (gated && a0.mbits_c % 32 && a0.mbits_c != a0.cout))) return RVIP_OK;      // (not served: the caller sees used == false)
    b.tiles_x = (int)cdiv(a0.w, TW); b.tiles_y = (int)cdiv(a0.h, TH);
    b.ntiles = a0.n * b.tiles_x * b.tiles_y;
    const int mb_tile = gated ? NCT * TH * TW * 4 : 0;           // one tile of mask words per input stage, behind the bias table
    // a third input stage (two items in flight per CU) where the 16-bit kernel keeps every weight chunk resident beside it
    const bool stg3_on = [] { const char* e = getenv("RVIP_IGEMM_STAGES"); return !(e && e[0] == '2'); }();      // (read per call: the host side of a launch; a test flips it)
    const bool res3 = WS16 && stg3_on && 3 * IN_BYTES + nchunks * W_BYTES + 256 + 3 * mb_tile <= LDS_MAX;
    b.nstg = res3 ? 3 : 2;
    const int mb_bytes = b.nstg * mb_tile;
    const bool res = res3 || 2 * IN_BYTES + nchunks * W_BYTES + 256 + mb_bytes <= LDS_MAX;
    b.wres = res ? nchunks : 0;
    b.lds_bias_off = b.nstg * IN_BYTES + (res ? nchunks : 2) * W_BYTES;
    b.lds_mb_off = b.lds_bias_off + 256;
    const int lds = b.lds_bias_off + 256 + mb_bytes;
    if (lds > LDS_MAX) return RVIP_OK;
    static std::atomic<int> attr_lds{0};             // idempotent attribute call; atomic so concurrent host threads do not race on the flag
    if constexpr (!PLAN_ONLY)
    if (!dry && lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ws_kernel<T, V5, TW, NCT, NPIX, 0, TAPS, NCW>()),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        if constexpr (TAPS == 4 && WS16) {
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ws_kernel<T, V5, TW, NCT, NPIX, 2, TAPS, NCW>()),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        }
        if constexpr (TAPS == 9) {
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ws_kernel<T, V5, TW, NCT, NPIX, 1, TAPS, NCW>()),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ws_kernel<T, V5, TW, NCT, NPIX, 2, TAPS, NCW>()),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ws_kernel<T, V5, TW, NCT, NPIX, 3, TAPS, NCW>()),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        }
        if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
        attr_lds = LDS_MAX;
    }
    const int cot = (int)cdiv(a0.cout, NCT * 32);
    if (cot > 1) b.nt_in = 0;       // several workgroup columns re-read the same input tile: keep it cached (measured)
    const int NZ = sp_fwd ? 4 : 1;
    int gx = a0.cus / (cot * NZ);                  // one workgroup per CU (LDS-limited), persistent over the pixel tiles
    if (gx < 1) gx = 1;
    if (gx > b.ntiles) gx = b.ntiles;
    b.stats = stats;
    { static const int dbg = [] { const char* e = getenv("RVIP_DBG"); return e ? atoi(e) : 0; }(); b.dbg = dbg; }
    if (rows_out) *rows_out = gx;
    if (stats && TAPS == 4 && (!WS16 || a0.subpix != 2 || smode != 2 || gated)) return dry ? RVIP_OK : RVIP_EUNSUPPORTED;
    if (dry) { used = true; return RVIP_OK; }
    if constexpr (PLAN_ONLY) {
        plan->args = b; plan->gx = gx; plan->gy = cot; plan->gz = NZ; plan->lds = lds;
        used = true;
        return RVIP_OK;
    } else {
    if (stats) {
        if constexpr (TAPS == 9) {
            if (smode == 2 && gated) hipLaunchKernelGGL((igemm_ws_kernel<T, V5, TW, NCT, NPIX, 3, TAPS, NCW>()), dim3((unsigned)gx, (unsigned)cot, (unsigned)NZ), dim3((NCW + 4) * 64), lds, s, b);
            else if (smode == 2) hipLaunchKernelGGL((igemm_ws_kernel<T, V5, TW, NCT, NPIX, 2, TAPS, NCW>()), dim3((unsigned)gx, (unsigned)cot, (unsigned)NZ), dim3((NCW + 4) * 64), lds, s, b);
            else hipLaunchKernelGGL((igemm_ws_kernel<T, V5, TW, NCT, NPIX, 1, TAPS, NCW>()), dim3((unsigned)gx, (unsigned)cot, (unsigned)NZ), dim3((NCW + 4) * 64), lds, s, b);
        } else if constexpr (WS16) {
            // TAPS = 4: only the data-gradient form carries partial rows, and only plain column sums
            if (a0.subpix != 2 || smode != 2 || gated) return RVIP_EUNSUPPORTED;
            hipLaunchKernelGGL((igemm_ws_kernel<T, V5, TW, NCT, NPIX, 2, TAPS, NCW>()), dim3((unsigned)gx, (unsigned)cot, (unsigned)NZ), dim3((NCW + 4) * 64), lds, s, b);
        } else return RVIP_EUNSUPPORTED;
    } else hipLaunchKernelGGL((igemm_ws_kernel<T, V5, TW, NCT, NPIX, 0, TAPS, NCW>()), dim3((unsigned)gx, (unsigned)cot, (unsigned)NZ), dim3((NCW + 4) * 64), lds, s, b);
    used = true;
    return check_launch();
    }
}

template <typename T, bool V5 = false>
static int dispatch_igemm_ws(const ConvArgs& a, hipStream_t s, bool& used, float* stats = nullptr, int* rows_out = nullptr, bool dry = false,
                             bool wide = false, int smode = 1) {
    bool two = a.cout > 32;
    if (two && !a.subpix) {
        // small maps: 64-channel tiles can leave half of the CUs without a workgroup (e.g. 256 -> 128 at 32 x 32: 64 tiles x 2
        // channel columns); 32-channel tiles double the workgroup count for a little more LDS traffic per FLOP
        const int tpx = (a.w > 16 && a.h >= 16) ? 512 : 256, tw = a.w > 16 ? 32 : 16;
        const long long ntiles = (long long)a.n * cdiv(a.w, tw) * cdiv(a.h, tpx / tw);
        if (ntiles * cdiv(a.cout, 64) <= a.cus / 2) two = false;
    }
    if (a.subpix) {                     // a.h, a.w = the low-resolution grid
        if (a.w > 16 && a.h >= 16) {
            // 16-bit types: eight compute waves (the 16x16x32 kernel; its four-wave form would hold 128 accumulators per lane): +0.4 % of the step
            if (sizeof(T) == 2) return two ? launch_igemm_ws<T, V5, 32, 2, 512, 4, 8>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 32, 1, 512, 4, 8>(a, s, used, stats, rows_out, dry, smode);
            return two ? launch_igemm_ws<T, V5, 32, 2, 512, 4>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 32, 1, 512, 4>(a, s, used, stats, rows_out, dry, smode);
        }
        if (a.w > 16) return two ? launch_igemm_ws<T, V5, 32, 2, 256, 4>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 32, 1, 256, 4>(a, s, used, stats, rows_out, dry, smode);
        return two ? launch_igemm_ws<T, V5, 16, 2, 256, 4>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 16, 1, 256, 4>(a, s, used, stats, rows_out, dry, smode);
    }
    if (a.w > 16 && a.h >= 16) {
        static const bool ncw4 = [] { const char* e = getenv("RVIP_IGEMM_NCW4"); return e && e[0] == '1'; }();      // (A/B: four compute waves of 128 pixels)
        if (ncw4) wide = false;
        if (wide) return two ? launch_igemm_ws<T, V5, 32, 2, 512, 9, 8>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 32, 1, 512, 9, 8>(a, s, used, stats, rows_out, dry, smode);
        return two ? launch_igemm_ws<T, V5, 32, 2, 512>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 32, 1, 512>(a, s, used, stats, rows_out, dry, smode);
    }
    if (a.w > 16) return two ? launch_igemm_ws<T, V5, 32, 2, 256>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 32, 1, 256>(a, s, used, stats, rows_out, dry, smode);
    return two ? launch_igemm_ws<T, V5, 16, 2, 256>(a, s, used, stats, rows_out, dry, smode) : launch_igemm_ws<T, V5, 16, 1, 256>(a, s, used, stats, rows_out, dry, smode);
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
            const uint16_t b = Vec<T>::enc(v);
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
template <> __device__ __forceinline__ float ld1<f16_t>(const f16_t* p) { return f16_to_f32(p->bits); }

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

// first layer, tiled form: the 8x32-pixel tile's 10x34 halo of the single input channel sits in LDS, a thread owns
// one channel vector for good (its 9 x VE weights stay in registers) and walks pixels/tiles; a wave stores 1 KiB
// of contiguous NHWC output per instruction.
// STATS: the BatchNormalization statistics of the stored output ride along -- every thread keeps the sum and the sum of squares
// of its channel vector over the pixels it produced, the workgroup folds its pixel slots through LDS and writes one partial
// row stats[blockIdx.x][2][cout] (finish with rvip_bn_stats_finalize): saves the 134 MB re-read of rvip_bn_train_stats.
template <typename T, bool STATS = false, int ACT = -1>      // ACT: compile-time activation (-1: the argument), as in the BN passes
__global__ __launch_bounds__(256) void conv3x3_c1_tiled(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, unsigned char* y,
                                                        int n, int h, int wd, int cout, int act, int tiles_x, int tiles_y,
                                                        float* __restrict__ stats = nullptr) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float xs[10 * 34];
    __shared__ float red[STATS ? 256 * VE : 1];
    float ssum[VE], ssq[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) ssum[e] = ssq[e] = 0.f;
    const int tid = threadIdx.x, cg = cout / VE, cv = tid % cg, ps = tid / cg, pps = 256 / cg;
    float wr[9][VE], br[VE];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VE; ++e) wr[t][e] = w[t * cout + cv * VE + e];
#pragma unroll
    for (int e = 0; e < VE; ++e) br[e] = bias ? bias[cv * VE + e] : 0.f;
    const int ntiles = n * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bx = tile;
        const int tx0 = (bx % tiles_x) * 32; bx /= tiles_x;
        const int ty0 = (bx % tiles_y) * 8;
        const long long img = bx / tiles_y;
        __syncthreads();
        for (int i = tid; i < 340; i += 256) {
            const int gy = ty0 - 1 + i / 34, gx = tx0 - 1 + i % 34;
            xs[i] = ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)wd) ? ld1<T>(x + (img * h + gy) * wd + gx) : 0.f;
        }
        __syncthreads();
        for (int p = ps; p < 256; p += pps) {
            const int py = p >> 5, px = p & 31;
            const int gy = ty0 + py, gx = tx0 + px;
            if (gy >= h || gx >= wd) continue;
            float xin[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) xin[t] = xs[(py + t / 3) * 34 + px + t % 3];
            float v[VE];
            // two channels per instruction (v_pk_fma_f32): the pass is VALU-bound -- 72 scalar FMAs per 16 bytes stored were most of it.
            // Per output the nine multiply-adds run in the same order as before: same bits.
            rvip_f32x2 acc2[VE / 2];
#pragma unroll
            for (int e2 = 0; e2 < VE / 2; ++e2) acc2[e2] = rvip_f32x2{br[2 * e2], br[2 * e2 + 1]};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const rvip_f32x2 x2 = {xin[t], xin[t]};
#pragma unroll
                for (int e2 = 0; e2 < VE / 2; ++e2)
                    acc2[e2] = __builtin_elementwise_fma(x2, rvip_f32x2{wr[t][2 * e2], wr[t][2 * e2 + 1]}, acc2[e2]);
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float acc = acc2[e >> 1][e & 1];
                v[e] = act_fwd(acc, ACT < 0 ? act : ACT);
                if constexpr (STATS) {
                    const float q = Vec<T>::round(v[e]);          // statistics of what is stored
                    ssum[e] += q;
                    ssq[e] = fmaf(q, q, ssq[e]);
                }
            }
            Vec<T>::store(y + ((((size_t)img * h + gy) * wd + gx) * cout + cv * VE) * sizeof(T), v);
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __syncthreads();
            if (ps < pps) {
#pragma unroll
                for (int e = 0; e < VE; ++e) red[ps * cout + cv * VE + e] = k == 0 ? ssum[e] : ssq[e];
            }
            __syncthreads();
            for (int col = tid; col < cout; col += 256) {
                float t = 0.f;
                for (int r = 0; r < pps; ++r) t += red[r * cout + col];
                stats[((size_t)blockIdx.x * 2 + k) * cout + col] = t;
            }
        }
    }
}

// First layer of the 3-D graph (Conv3D, Cin = 1, 27 taps): the three input slices d-1, d, d+1 of the tile in LDS, the
// [27][Cout] kernel in LDS as well (27 x VE registers per thread would not fit); otherwise as conv3x3_c1_tiled.
template <typename T>
__global__ __launch_bounds__(256) void conv3d_c1_tiled(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, unsigned char* y,
                                                       int n, int depth, int h, int wd, int cout, int act, int tiles_x, int tiles_y) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float xs[3][10 * 34];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* ws = reinterpret_cast<float*>(dyn);                  // [27][cout]
    const int tid = threadIdx.x, cg = cout / VE, cv = tid % cg, ps = tid / cg, pps = 256 / cg;
    for (int i = tid; i < 27 * cout; i += 256) ws[i] = w[i];
    float br[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) br[e] = bias ? bias[cv * VE + e] : 0.f;
    const int ntiles = n * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bx = tile;
        const int tx0 = (bx % tiles_x) * 32; bx /= tiles_x;
        const int ty0 = (bx % tiles_y) * 8;
        const int img = bx / tiles_y, dz = img % depth;
        __syncthreads();
        for (int i = tid; i < 3 * 340; i += 256) {
            const int k = i / 340, r = i - k * 340;
            const int gy = ty0 - 1 + r / 34, gx = tx0 - 1 + r % 34;
            const bool ok = (unsigned)(dz + k - 1) < (unsigned)depth && (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)wd;
            xs[k][r] = ok ? ld1<T>(x + ((long long)(img + k - 1) * h + gy) * wd + gx) : 0.f;
        }
        __syncthreads();
        // A thread owns one channel vector of FOUR vertically adjacent pixels: a tap's weights are read from LDS once for the four (the
        // one-pixel form read 9 LDS words per 8 multiply-adds and was LDS-bound: 502 us at config 5 against 54 us of stores), the six
        // input rows of a column are read once per tap column, and a wave still stores 1 KiB of contiguous NHWC output per instruction.
        // Per output the 27 multiply-adds run in the same order as before (bias, then k, kh, kw): same bits.
        for (int q = ps; q < 64; q += pps) {
            const int px = q & 31, py0 = (q >> 5) * 4;
            const int gx = tx0 + px;
            float v[4][VE];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < VE; ++e) v[j][e] = br[e];
            // rolled over the nine (slice, tap row) pairs on purpose: unrolled, the compiler hoists all 27 x VE tap weights out of the
            // pixel loop into registers -- 256 VGPRs + AGPR copies, ONE wave per SIMD and nothing to hide latency behind (502 us)
#pragma unroll 1
            for (int kr = 0; kr < 9; ++kr) {
                const int k = kr / 3, dy = kr - 3 * k;
                float xw[4][3];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) xw[j][dx] = xs[k][(py0 + j + dy) * 34 + px + dx];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float wt[VE];
#pragma unroll
                    for (int e = 0; e < VE; ++e) wt[e] = ws[(kr * 3 + dx) * cout + cv * VE + e];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < VE; ++e) v[j][e] = fmaf(xw[j][dx], wt[e], v[j][e]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = ty0 + py0 + j;
                if (gy >= h || gx >= wd) continue;
#pragma unroll
                for (int e = 0; e < VE; ++e) v[j][e] = act_fwd(v[j][e], act);
                Vec<T>::store(y + ((((size_t)img * h + gy) * wd + gx) * cout + cv * VE) * sizeof(T), v[j]);
            }
        }
    }
}

// First layer with IMG_CHANNELS = 2..4 (Unets.py:77; the reference's configs all use 1): the halo of every input channel in LDS,
// the [9][Cin][Cout] kernel in LDS; otherwise as conv3x3_c1_tiled.  x is NHWC [n][h][w][cin].
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_cn_tiled(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, unsigned char* y,
                                                        int n, int h, int wd, int cin, int cout, int act, int tiles_x, int tiles_y) {
    constexpr int VE = Vec<T>::VE;
    __shared__ float xs[4][10 * 34];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* ws = reinterpret_cast<float*>(dyn);                  // [9][cin][cout]
    const int tid = threadIdx.x, cg = cout / VE, cv = tid % cg, ps = tid / cg, pps = 256 / cg;
    for (int i = tid; i < 9 * cin * cout; i += 256) ws[i] = w[i];
    float br[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) br[e] = bias ? bias[cv * VE + e] : 0.f;
    const int ntiles = n * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bx = tile;
        const int tx0 = (bx % tiles_x) * 32; bx /= tiles_x;
        const int ty0 = (bx % tiles_y) * 8;
        const long long img = bx / tiles_y;
        __syncthreads();
        for (int i = tid; i < cin * 340; i += 256) {
            const int r = i / cin, ci = i - r * cin;
            const int gy = ty0 - 1 + r / 34, gx = tx0 - 1 + r % 34;
            const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)wd;
            xs[ci][r] = ok ? ld1<T>(x + ((img * h + gy) * wd + gx) * cin + ci) : 0.f;
        }
        __syncthreads();
        if (ps >= pps) continue;                                // 256 % cg != 0: the last threads own no channel vector
        for (int p = ps; p < 256; p += pps) {
            const int py = p >> 5, px = p & 31;
            const int gy = ty0 + py, gx = tx0 + px;
            if (gy >= h || gx >= wd) continue;
            float v[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] = br[e];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                for (int ci = 0; ci < cin; ++ci) {
                    const float xv = xs[ci][(py + t / 3) * 34 + px + t % 3];
                    const float* wt = ws + (t * cin + ci) * cout + cv * VE;
#pragma unroll
                    for (int e = 0; e < VE; ++e) v[e] = fmaf(xv, wt[e], v[e]);
                }
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] = act_fwd(v[e], act);
            Vec<T>::store(y + ((((size_t)img * h + gy) * wd + gx) * cout + cv * VE) * sizeof(T), v);
        }
    }
}

// Phase kernels of the sub-pixel form: w_phase[2a+b][2u+v][co][ci] = sum of W[kh][kw][ci][co] over the taps that land on
// low-resolution offset (u, v) for output phase (a, b): rows a=0: u=0 <- {0}, u=1 <- {1,2}; a=1: u=0 <- {0,1}, u=1 <- {2}
// (columns alike).  Summed in fp32, rounded once.
template <typename T>
__device__ __forceinline__ void pack_subpixel_range(const float* __restrict__ w, int cin, int cout, T* __restrict__ wp) {
    const long long total = 16LL * cin * cout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % cin);
        const int co = (int)((i / cin) % cout);
        const int tap = (int)((i / ((long long)cin * cout)) & 3), ph = (int)(i / (4LL * cin * cout));
        const int a = ph >> 1, b = ph & 1, u = tap >> 1, v = tap & 1;
        const int kh0 = a == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2), kh1 = a == 0 ? (u == 0 ? 0 : 2) : (u == 0 ? 1 : 2);
        const int kw0 = b == 0 ? (v == 0 ? 0 : 1) : (v == 0 ? 0 : 2), kw1 = b == 0 ? (v == 0 ? 0 : 2) : (v == 0 ? 1 : 2);
        float acc = 0.f;
        for (int kh = kh0; kh <= kh1; ++kh)
            for (int kw = kw0; kw <= kw1; ++kw) acc += w[((size_t)(kh * 3 + kw) * cin + ci) * cout + co];
        if constexpr (sizeof(T) == 4) wp[i] = acc;
        else wp[i].bits = Vec<T>::enc(acc);
    }
}

// The data gradient of the same layer (rvip_conv3x3_desc.subpix = 2): w_dphase[2 al + be][2 u + v][ci][co] (ci = the layer's INPUT
// channel = the result's channel, co = its output channel = the contraction) = sum of W[kh][kw][ci][co] over the taps through which
// source phase (al, be) of the gradient -- row 2y' + al -- at window position u (low-resolution row y - al + u) reaches the
// low-resolution pixel y:  al = 0: u = 0 <- kh {1, 2}, u = 1 <- {0};  al = 1: u = 0 <- {2}, u = 1 <- {0, 1}  (columns alike).
template <typename T>
__device__ __forceinline__ void pack_subpixel_dgrad_range(const float* __restrict__ w, int cin, int cout, T* __restrict__ wp) {
    const long long total = 16LL * cin * cout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % cout);
        const int ci = (int)((i / cout) % cin);
        const int tap = (int)((i / ((long long)cin * cout)) & 3), ph = (int)(i / (4LL * cin * cout));
        const int al = ph >> 1, be = ph & 1, u = tap >> 1, v = tap & 1;
        const int kh0 = al == 0 ? (u == 0 ? 1 : 0) : (u == 0 ? 2 : 0), kh1 = al == 0 ? (u == 0 ? 2 : 0) : (u == 0 ? 2 : 1);
        const int kw0 = be == 0 ? (v == 0 ? 1 : 0) : (v == 0 ? 2 : 0), kw1 = be == 0 ? (v == 0 ? 2 : 0) : (v == 0 ? 2 : 1);
        float acc = 0.f;
        for (int kh = kh0; kh <= kh1; ++kh)
            for (int kw = kw0; kw <= kw1; ++kw) acc += w[((size_t)(kh * 3 + kw) * cin + ci) * cout + co];
        if constexpr (sizeof(T) == 4) wp[i] = acc;
        else wp[i].bits = Vec<T>::enc(acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_subpixel_kernel(const float* __restrict__ w, int cin, int cout, T* __restrict__ wp, int dgrad) {
    if (dgrad) pack_subpixel_dgrad_range<T>(w, cin, cout, wp);
    else pack_subpixel_range<T>(w, cin, cout, wp);
}

// all 3x3 kernels of the model in ONE launch: table-driven re-layout (see pack_w_kernel)
struct PackEntry { long long w_off; long long f_off; long long d_off; int cin, cout; int taps, reserved; };
// A workgroup moves tiles of one tap: 32 input channels x 64 output channels, read as 256-byte rows of the HWIO master,
// written as 128-byte rows of w_dgrad (same orientation, taps reversed) and, transposed through LDS, as 64-byte rows of
// w_fwd.  (The element-per-thread form scattered 2-byte stores Cin apart: 1.5 ms for the 138 M parameters of cfg 4.)
// four consecutive output channels of one HWIO row; `ragged` (Cout % 4 != 0: rows are not 16-byte aligned and the last group is
// partial) takes the element path, bounded by the `nv` channels the row still has
__device__ __forceinline__ float4 pack_load4(const float* src, bool ragged, int nv) {
    if (!ragged) return *reinterpret_cast<const float4*>(src);
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (nv > 0) v.x = src[0];
    if (nv > 1) v.y = src[1];
    if (nv > 2) v.z = src[2];
    if (nv > 3) v.w = src[3];
    return v;
}
template <typename T>
__device__ __forceinline__ void pack_store4(T* dst, const float4& v, bool ragged, int nv) {
    if (!ragged) {
        if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(dst) = v;
        else {
            uint2 pk;
            pk.x = Vec<T>::pack2(v.x, v.y);
            pk.y = Vec<T>::pack2(v.z, v.w);
            *reinterpret_cast<uint2*>(dst) = pk;
        }
        return;
    }
    const float e[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < 4 && i < nv; ++i) {
        if constexpr (sizeof(T) == 4) dst[i] = e[i];
        else dst[i].bits = Vec<T>::enc(e[i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(const float* __restrict__ theta, const PackEntry* __restrict__ tab,
                                                       T* __restrict__ wf_base, T* __restrict__ wd_base, uint32_t* tick) {
    constexpr int TI = 32, TO = 64, LDW = TO + 1;
    __shared__ float tile[TI * LDW];
    // the optimiser step's closing launch also counts the step (nothing in this kernel reads the counter; its readers --
    // Adam's bias correction and the dropout keys of the next step -- are ordered before / after this launch by the stream)
    if (tick && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) tick[RVIP_STATE_STEP] += 1u;
    const PackEntry en = tab[blockIdx.y];
    const float* w = theta + en.w_off;
    T* wf = wf_base + en.f_off;
    T* wd = wd_base + en.d_off;
    const bool ragged = (en.cout & 3) != 0;
    if (en.reserved == 1) {
        // mode 1: the phase kernels of the sub-pixel up-conv form, [4][4][Cout][Cin] at f_off, and of its data gradient, [4][4][Cin][Cout]
        // at d_off (pack_subpixel_range / pack_subpixel_dgrad_range, same sums in the same order), tiled like mode 0: the element-per-
        // thread form read the master Cout floats apart for the forward set (56 us for the whole launch against 44 without the entries)
        const int nbi = (en.cin + TI - 1) / TI, nbo = (en.cout + TO - 1) / TO;
        const int tid = threadIdx.x;
        for (int tl = blockIdx.x; tl < 16 * nbi * nbo; tl += gridDim.x) {
            const int bo = tl % nbo, bi = (tl / nbo) % nbi, pt = tl / (nbo * nbi);      // pt = 4 * phase + tap
            const int ci0 = bi * TI, co0 = bo * TO;
            const int pa = pt >> 3, pb = (pt >> 2) & 1, u = (pt >> 1) & 1, v = pt & 1;
            // forward: phase (a, b), low-resolution offset (u, v);  data gradient: source phase (al, be), window position (u, v)
            const int fh0 = pa == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2), fh1 = pa == 0 ? (u == 0 ? 0 : 2) : (u == 0 ? 1 : 2);
            const int fw0 = pb == 0 ? (v == 0 ? 0 : 1) : (v == 0 ? 0 : 2), fw1 = pb == 0 ? (v == 0 ? 0 : 2) : (v == 0 ? 1 : 2);
            const int dh0 = pa == 0 ? (u == 0 ? 1 : 0) : (u == 0 ? 2 : 0), dh1 = pa == 0 ? (u == 0 ? 2 : 0) : (u == 0 ? 2 : 1);
            const int dw0 = pb == 0 ? (v == 0 ? 1 : 0) : (v == 0 ? 2 : 0), dw1 = pb == 0 ? (v == 0 ? 2 : 0) : (v == 0 ? 2 : 1);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = tid + q * 256, r = idx >> 4, c4 = (idx & 15) * 4;
                const bool ok = ci0 + r < en.cin && co0 + c4 < en.cout;
                float4 sf = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const float* src = w + (size_t)(ci0 + r) * en.cout + co0 + c4;
                    const int nv = en.cout - (co0 + c4);      // valid output channels of this group (< 4 only when Cout % 4 != 0)
                    for (int kh = fh0; kh <= fh1; ++kh)
                        for (int kw = fw0; kw <= fw1; ++kw) {
                            const float4 t4 = pack_load4(src + (size_t)(kh * 3 + kw) * en.cin * en.cout, ragged, nv);
                            sf.x += t4.x; sf.y += t4.y; sf.z += t4.z; sf.w += t4.w;
                        }
                    for (int kh = dh0; kh <= dh1; ++kh)
                        for (int kw = dw0; kw <= dw1; ++kw) {
                            const float4 t4 = pack_load4(src + (size_t)(kh * 3 + kw) * en.cin * en.cout, ragged, nv);
                            sd.x += t4.x; sd.y += t4.y; sd.z += t4.z; sd.w += t4.w;
                        }
                    pack_store4<T>(wd + ((size_t)pt * en.cin + ci0 + r) * en.cout + co0 + c4, sd, ragged, nv);
                }
                tile[r * LDW + c4] = sf.x; tile[r * LDW + c4 + 1] = sf.y; tile[r * LDW + c4 + 2] = sf.z; tile[r * LDW + c4 + 3] = sf.w;
            }
            __syncthreads();
            const int o = tid >> 2, i8 = (tid & 3) * 8;
            if (co0 + o < en.cout && ci0 + i8 < en.cin) {
                float x8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = tile[(i8 + e) * LDW + o];
                T* dst = wf + ((size_t)pt * en.cout + co0 + o) * en.cin + ci0 + i8;
                if (en.cin & 7) {
                    for (int e = 0; e < 8 && ci0 + i8 + e < en.cin; ++e) {
                        if constexpr (sizeof(T) == 4) dst[e] = x8[e];
                        else dst[e].bits = Vec<T>::enc(x8[e]);
                    }
                } else if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(dst) = float4{x8[0], x8[1], x8[2], x8[3]};
                    *reinterpret_cast<float4*>(dst + 4) = float4{x8[4], x8[5], x8[6], x8[7]};
                } else {
                    uint4 pk;
                    pk.x = Vec<T>::pack2(x8[0], x8[1]);
                    pk.y = Vec<T>::pack2(x8[2], x8[3]);
                    pk.z = Vec<T>::pack2(x8[4], x8[5]);
                    pk.w = Vec<T>::pack2(x8[6], x8[7]);
                    *reinterpret_cast<uint4*>(dst) = pk;
                }
            }
        }
        return;
    }
    const int nbi = (en.cin + TI - 1) / TI, nbo = (en.cout + TO - 1) / TO;
    const int taps = en.taps > 0 ? en.taps : 9;
    const int ntiles = taps * nbi * nbo;
    const int tid = threadIdx.x;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int bo = tl % nbo, bi = (tl / nbo) % nbi, t = tl / (nbo * nbi);
        const int ci0 = bi * TI, co0 = bo * TO;
        __syncthreads();
        // read: 32 rows x 16 float4
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * 256, r = idx >> 4, c4 = (idx & 15) * 4;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            const int nv = en.cout - (co0 + c4);
            if (ci0 + r < en.cin && nv > 0) v = pack_load4(w + ((size_t)t * en.cin + ci0 + r) * en.cout + co0 + c4, ragged, nv);
            tile[r * LDW + c4] = v.x; tile[r * LDW + c4 + 1] = v.y; tile[r * LDW + c4 + 2] = v.z; tile[r * LDW + c4 + 3] = v.w;
            // w_dgrad[8 - t][ci][co]: same row, 4 consecutive output channels
            if (ci0 + r < en.cin && nv > 0)
                pack_store4<T>(wd + ((size_t)(taps - 1 - t) * en.cin + ci0 + r) * en.cout + co0 + c4, v, ragged, nv);
        }
        __syncthreads();
        // w_fwd[t][co][ci]: thread = (output channel, 8 input channels)
        const int o = tid >> 2, i8 = (tid & 3) * 8;
        if (co0 + o < en.cout && ci0 + i8 < en.cin) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[(i8 + e) * LDW + o];
            T* dst = wf + ((size_t)t * en.cout + co0 + o) * en.cin + ci0 + i8;
            if (en.cin & 7) {                               // ragged channel count: element stores
                for (int e = 0; e < 8 && ci0 + i8 + e < en.cin; ++e) {
                    if constexpr (sizeof(T) == 4) dst[e] = v[e];
                    else dst[e].bits = Vec<T>::enc(v[e]);
                }
            } else if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(dst) = float4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4*>(dst + 4) = float4{v[4], v[5], v[6], v[7]};
            } else {
                uint4 pk;
                pk.x = Vec<T>::pack2(v[0], v[1]);
                pk.y = Vec<T>::pack2(v[2], v[3]);
                pk.z = Vec<T>::pack2(v[4], v[5]);
                pk.w = Vec<T>::pack2(v[6], v[7]);
                *reinterpret_cast<uint4*>(dst) = pk;
            }
        }
    }
}

// Kernel selection: the wave-specialised LDS-DMA implicit GEMM (conv3x3_igemm_ws16 for the 16-bit types, conv3x3_igemm_ws for
// f32 and for the four-compute-wave sub-pixel tiling) serves every shape it is eligible for; the register-staged conv3x3_igemm is
// the fallback for ragged channel counts (concat halves that are not whole 64-byte chunks, Cout % 8).  The generations measured
// and retired on the way (igemm v2: every wave stages and computes; two workgroups per CU on the one-chunk layers) are in DESIGN.md.
static int conv_args_from_desc(const rvip_conv3x3_desc* d, ConvArgs& a) {
    if (!d || !d->x0 || !d->w_packed || !d->y) return RVIP_EINVAL;
    const int ve = RVIP_VE(d->dtype);
    if (!RVIP_DT_OK(d->dtype)) return RVIP_EINVAL;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->c0 <= 0) return RVIP_EINVAL;
    if (d->c0 % ve || d->c1 % ve || d->cout % 4) return RVIP_EINVAL;
    if ((d->c1 > 0) != (d->x1 != nullptr)) return RVIP_EINVAL;
    if (d->up0 && ((d->h | d->w) & 1)) return RVIP_EINVAL;
    if (d->up0 < 0 || d->up0 > 2 || (d->up0 == 2 && d->c1 > 0)) return RVIP_EINVAL;
    if (d->y1 && (d->csplit <= 0 || d->csplit >= d->cout || d->csplit % 4)) return RVIP_EINVAL;
    if ((long long)d->n * d->h * d->w >= (1LL << 31)) return RVIP_EINVAL;
    a.x0 = (const unsigned char*)d->x0; a.x1 = (const unsigned char*)d->x1;
    a.wp = (const unsigned char*)d->w_packed; a.bias = d->bias;
    a.y = (unsigned char*)d->y; a.y1 = (unsigned char*)d->y1;
    a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0 ? 1 : 0; a.zs = d->up0 == 2; a.csplit = d->csplit;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->c0 + d->c1; a.cout = d->cout; a.act = d->act;
    a.tiles_x = a.tiles_y = 0;
    a.depth = d->depth > 0 ? d->depth : 1;
    a.kd = d->kd > 0 ? d->kd : 1;
    if ((a.kd != 1 && a.kd != 3) || d->n % a.depth) return RVIP_EINVAL;
    a.down2 = d->down2 ? 1 : 0;
    if (a.down2 && (d->y1 || d->bias || d->act != RVIP_ACT_NONE || ((d->h | d->w) & 1))) return RVIP_EINVAL;
    a.nt_in = d->stream_in ? 1 : 0;
    a.subpix = d->subpix == 2 ? 2 : (d->subpix ? 1 : 0);
    a.mbits = nullptr; a.mbits_c = 0; a.mscale = 1.f; a.sbits = nullptr;
    a.sums_from = (d->sums_from > 0 && d->sums_from % 32 == 0) ? d->sums_from : 0;
    a.cus = (d->cu_limit > 0 && d->cu_limit < 256) ? d->cu_limit : 256;
    if (a.subpix == 1) {                 // UpSampling2D -> conv as four phase convolutions on the low-resolution grid
        if (d->up0 != 1 || d->c1 || d->y1 || a.kd != 1 || a.down2) return RVIP_EINVAL;
        a.up0 = 0; a.h = d->h / 2; a.w = d->w / 2;
    } else if (a.subpix == 2) {          // its data gradient: x0 at the up-sampled size h x w, result on the h/2 x w/2 grid, K loop over four source phases
        if (d->up0 || d->c1 || d->y1 || a.kd != 1 || a.down2 || d->bias || d->act != RVIP_ACT_NONE || ((d->h | d->w) & 1) || d->dtype == RVIP_F32) return RVIP_EINVAL;
        a.h = d->h / 2; a.w = d->w / 2; a.kd = 4;
    }
    return RVIP_OK;
}

}  // namespace rvip

#ifndef RVIP_KERNELS_ONLY        /* rvip_pair.hip includes this file for its kernels and launch geometry only */
using namespace rvip;

extern "C" int rvip_abi_version(void) { return RVIP_ABI_VERSION; }     // 8: cu_limit of the conv / weight-gradient descriptors; 3: round-2 prune; 4: window argmax of the pooled stages (apply / BN-backward descriptors), IMG_CHANNELS 2..4 entry points
extern "C" const char* rvip_build_info(void) { return "rvip_hip gfx950 wave64 mfma"; }
extern "C" int rvip_last_hip_error(void) { return g_last_hip_error; }
extern "C" int rvip_device_check(void) {
    const hipError_t a = hipDeviceSynchronize();
    const hipError_t b = hipGetLastError();
    const hipError_t e = a != hipSuccess ? a : b;
    if (e != hipSuccess) g_last_hip_error = (int)e;
    return (int)e;
}

extern "C" int rvip_conv3x3_fwd(const rvip_conv3x3_desc* d, void* stream) {
    (void)hipGetLastError();
    ConvArgs a;
    int rc0 = conv_args_from_desc(d, a);
    if (rc0) return rc0;
    if (d->mask_bits) return RVIP_EINVAL;                                  // (rvip_conv3x3_fwd_sums)
    if (d->sign_bits) {
        if (d->y1 || d->down2 || d->cout % 8) return RVIP_EINVAL;
        a.sbits = d->sign_bits;
    }
    hipStream_t s = (hipStream_t)stream;
    {
        bool used = false;
        const int rc = by_dtype(d->dtype, [&](auto t) { return dispatch_igemm_ws<decltype(t), true>(a, s, used, nullptr, nullptr, false, !a.subpix); });
        if (rc || used) return rc;
    }
    if (a.kd > 1 || a.down2 || a.subpix || a.sbits) return RVIP_EUNSUPPORTED;        // the register-staged fallback is 2-D only and has the plain epilogue
    return by_dtype(d->dtype, [&](auto t) { return dispatch_igemm<decltype(t)>(a, s); });
}

// 1 if rvip_conv3x3_fwd writes d->sign_bits for this launch (the LDS-DMA kernels serve it), else 0
extern "C" int rvip_conv3x3_sign_bits_ok(const rvip_conv3x3_desc* d) {
    ConvArgs a;
    if (conv_args_from_desc(d, a) != RVIP_OK || !d->sign_bits || d->y1 || d->down2 || d->cout % 8) return 0;
    a.sbits = d->sign_bits;
    bool used = false;
    const int rc = by_dtype(d->dtype, [&](auto t) { return dispatch_igemm_ws<decltype(t), true>(a, nullptr, used, nullptr, nullptr, true, !a.subpix); });
    return (rc == RVIP_OK && used) ? 1 : 0;
}

// Number of partial-statistics rows rvip_conv3x3_fwd_stats will write for this shape (0 = this shape is served by the
// register-staged fallback kernel, which does not fuse statistics: run rvip_bn_train_stats instead).
extern "C" int rvip_conv3x3_fwd_stats_rows(const rvip_conv3x3_desc* d) {
    ConvArgs a;
    if (conv_args_from_desc(d, a) != RVIP_OK || d->y1 || d->down2) return 0;
    bool used = false; int rows = 0;
    const int rc = by_dtype(d->dtype, [&](auto t) { return dispatch_igemm_ws<decltype(t), true>(a, nullptr, used, nullptr, &rows, true, !a.subpix); });
    return (rc == RVIP_OK && used) ? rows : 0;
}

// conv + per-channel partial sums (sum, sum of squares) of the stored output: stats_ws[rows][2][cout]
extern "C" int rvip_conv3x3_fwd_stats(const rvip_conv3x3_desc* d, float* stats_ws, size_t stats_ws_bytes, void* stream) {
    (void)hipGetLastError();
    ConvArgs a;
    int rc = conv_args_from_desc(d, a);
    if (rc) return rc;
    const int rows = rvip_conv3x3_fwd_stats_rows(d);
    if (!stats_ws || rows <= 0) return RVIP_EUNSUPPORTED;
    if (d->mask_bits || d->sign_bits) return RVIP_EINVAL;     // (rvip_conv3x3_fwd_sums / rvip_conv3x3_fwd)
    if (stats_ws_bytes < (size_t)rows * 2 * d->cout * sizeof(float)) return RVIP_EWORKSPACE;
    bool used = false;
    hipStream_t s = (hipStream_t)stream;
    rc = by_dtype(d->dtype, [&](auto t) { return dispatch_igemm_ws<decltype(t), true>(a, s, used, stats_ws, nullptr, false, !a.subpix); });
    if (rc) return rc;
    return used ? RVIP_OK : RVIP_EUNSUPPORTED;
}

// Data-gradient launches: the per-channel sums of the STORED result as partial rows sums_ws[rows][cout] (include/rvip_hip.h)
extern "C" int rvip_conv3x3_fwd_sums_rows(const rvip_conv3x3_desc* d) {
    ConvArgs a;
    if (conv_args_from_desc(d, a) != RVIP_OK || d->subpix == 1 || (d->subpix == 2 && d->mask_bits)) return 0;
    if (d->mask_bits) { a.mbits = d->mask_bits; a.mbits_c = d->mask_channels; a.mscale = d->mask_scale; }
    bool used = false; int rows = 0;
    static float dummy;                                   // dry run: only tells the dispatcher that partial rows are wanted
    const int rc = by_dtype(d->dtype, [&](auto t) { return dispatch_igemm_ws<decltype(t), true>(a, nullptr, used, &dummy, &rows, true, true, 2); });
    return (rc == RVIP_OK && used) ? rows : 0;
}

extern "C" int rvip_conv3x3_fwd_sums(const rvip_conv3x3_desc* d, float* sums_ws, size_t sums_ws_bytes, void* stream) {
    (void)hipGetLastError();
    ConvArgs a;
    int rc = conv_args_from_desc(d, a);
    if (rc) return rc;
    const int rows = rvip_conv3x3_fwd_sums_rows(d);
    if (!sums_ws || rows <= 0) return RVIP_EUNSUPPORTED;
    if (d->act != RVIP_ACT_NONE || d->bias) return RVIP_EINVAL;          // a data gradient: no bias, no activation
    if (sums_ws_bytes < (size_t)rows * d->cout * sizeof(float)) return RVIP_EWORKSPACE;
    if (d->sign_bits) return RVIP_EINVAL;
    if (d->mask_bits) {                               // the result is gated element-wise by bit planes (Dropout / ReLU backward)
        if (d->mask_channels <= 0 || (d->mask_channels % 32 && d->mask_channels != d->cout) || d->mask_channels > d->cout || d->down2 || !(d->mask_scale > 0.f)) return RVIP_EINVAL;
        a.mbits = d->mask_bits; a.mbits_c = d->mask_channels; a.mscale = d->mask_scale;
    }
    bool used = false;
    hipStream_t s = (hipStream_t)stream;
    rc = by_dtype(d->dtype, [&](auto t) { return dispatch_igemm_ws<decltype(t), true>(a, s, used, sums_ws, nullptr, false, true, 2); });
    if (rc) return rc;
    return used ? RVIP_OK : RVIP_EUNSUPPORTED;
}

extern "C" int rvip_pack_conv3x3_weights(const float* w, int cin, int cout, int dtype, void* wf, void* wd, void* stream) {
    (void)hipGetLastError();
    if (!w || cin <= 0 || cout <= 0 || (!wf && !wd)) return RVIP_EINVAL;
    const long long total = 9LL * cin * cout;
    const int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(pack_w_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (bf16_t*)wf, (bf16_t*)wd);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(pack_w_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (f16_t*)wf, (f16_t*)wd);
    else if (dtype == RVIP_F32) hipLaunchKernelGGL(pack_w_kernel<float>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (float*)wf, (float*)wd);
    else return RVIP_EINVAL;
    return check_launch();
}

extern "C" int rvip_conv3x3_c1_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int cout,
                                   int act, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !y || n <= 0 || h <= 0 || w_ <= 0) return RVIP_EINVAL;
    const int ve = RVIP_VE(dtype);
    if (cout <= 0 || cout % ve) return RVIP_EINVAL;
    const long long total = (long long)n * h * w_ * (cout / ve);
    hipStream_t s = (hipStream_t)stream;
    if (!RVIP_DT_OK(dtype)) return RVIP_EINVAL;
    if (256 % (cout / ve) == 0) {
        const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
        long long nt = (long long)n * tx * ty;
        dim3 g2((unsigned)(nt < 2048 ? nt : 2048));
        if (dtype == RVIP_BF16) hipLaunchKernelGGL((conv3x3_c1_tiled<bf16_t, false>), g2, dim3(256), 0, s, (const bf16_t*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act, tx, ty, (float*)nullptr);
        else if (dtype == RVIP_F16) hipLaunchKernelGGL((conv3x3_c1_tiled<f16_t, false>), g2, dim3(256), 0, s, (const f16_t*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act, tx, ty, (float*)nullptr);
        else hipLaunchKernelGGL((conv3x3_c1_tiled<float, false>), g2, dim3(256), 0, s, (const float*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act, tx, ty, (float*)nullptr);
        return check_launch();
    }
    dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(conv3x3_c1_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(conv3x3_c1_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act);
    else if (dtype == RVIP_F32) hipLaunchKernelGGL(conv3x3_c1_kernel<float>, grid, dim3(256), 0, s, (const float*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act);
    else return RVIP_EINVAL;
    return check_launch();
}

static int c1_tiled_blocks(int n, int h, int w_) {
    const long long nt = (long long)n * cdiv(w_, 32) * cdiv(h, 8);
    // 1 024 workgroups (four per CU): the statistics fold behind the launch reads one partial row per workgroup, and at 2 048 that
    // single-workgroup fold was a chain of 32 round trips (conv 43.2 + fold 8.0 us against 41.5 + 5.3; 512: 47.7 + 3.9, 256: 67 + 4)
    return (int)(nt < 1024 ? nt : 1024);
}

// rows of partial statistics rvip_conv3x3_c1_fwd_stats writes (0 = the shape runs on the untiled kernel: use rvip_bn_train_stats)
extern "C" int rvip_conv3x3_c1_fwd_stats_rows(int n, int h, int w_, int cout, int dtype) {
    if (!RVIP_DT_OK(dtype) || n <= 0 || h <= 0 || w_ <= 0 || cout <= 0) return 0;
    const int ve = RVIP_VE(dtype);
    if (cout % ve || 256 % (cout / ve)) return 0;
    return c1_tiled_blocks(n, h, w_);
}

extern "C" int rvip_conv3x3_c1_fwd_stats(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int cout,
                                         int act, int dtype, float* stats_ws, size_t stats_ws_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !y || !stats_ws) return RVIP_EINVAL;
    const int rows = rvip_conv3x3_c1_fwd_stats_rows(n, h, w_, cout, dtype);
    if (rows <= 0) return RVIP_EUNSUPPORTED;
    if (stats_ws_bytes < (size_t)rows * 2 * cout * sizeof(float)) return RVIP_EWORKSPACE;
    const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
    hipStream_t s = (hipStream_t)stream;
    by_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        if (act == RVIP_ACT_RELU) hipLaunchKernelGGL((conv3x3_c1_tiled<T, true, RVIP_ACT_RELU>), dim3((unsigned)rows), dim3(256), 0, s, (const T*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act, tx, ty, stats_ws);
        else hipLaunchKernelGGL((conv3x3_c1_tiled<T, true>), dim3((unsigned)rows), dim3(256), 0, s, (const T*)x, w, bias, (unsigned char*)y, n, h, w_, cout, act, tx, ty, stats_ws);
        return 0;
    });
    return check_launch();
}

extern "C" int rvip_conv3d_c1_fwd(const void* x, const float* w, const float* bias, void* y, int n, int depth, int h, int w_,
                                  int cout, int act, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !y || n <= 0 || depth <= 0 || n % depth || h <= 0 || w_ <= 0) return RVIP_EINVAL;
    if (!RVIP_DT_OK(dtype)) return RVIP_EINVAL;
    const int ve = RVIP_VE(dtype);
    if (cout <= 0 || cout % ve || 256 % (cout / ve) || cout > 256) return RVIP_EINVAL;
    const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
    long long nt = (long long)n * tx * ty;
    dim3 grid((unsigned)(nt < 4096 ? nt : 4096));
    const size_t lds = (size_t)27 * cout * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(conv3d_c1_tiled<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)x, w, bias, (unsigned char*)y, n, depth, h, w_, cout, act, tx, ty);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(conv3d_c1_tiled<f16_t>, grid, dim3(256), lds, s, (const f16_t*)x, w, bias, (unsigned char*)y, n, depth, h, w_, cout, act, tx, ty);
    else hipLaunchKernelGGL(conv3d_c1_tiled<float>, grid, dim3(256), lds, s, (const float*)x, w, bias, (unsigned char*)y, n, depth, h, w_, cout, act, tx, ty);
    return check_launch();
}

extern "C" int rvip_conv3x3_cn_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int cin,
                                   int cout, int act, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !y || n <= 0 || h <= 0 || w_ <= 0 || cin < 1 || cin > 4) return RVIP_EINVAL;
    if (!RVIP_DT_OK(dtype)) return RVIP_EINVAL;
    const int ve = RVIP_VE(dtype);
    if (cout <= 0 || cout % ve || cout / ve > 256 || 9 * cin * cout * (int)sizeof(float) > 48 * 1024) return RVIP_EINVAL;
    const int tx = (int)cdiv(w_, 32), ty = (int)cdiv(h, 8);
    long long nt = (long long)n * tx * ty;
    dim3 grid((unsigned)(nt < 4096 ? nt : 4096));
    const size_t lds = (size_t)9 * cin * cout * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    by_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(conv3x3_cn_tiled<T>, grid, dim3(256), lds, s, (const T*)x, w, bias, (unsigned char*)y, n, h, w_, cin, cout, act, tx, ty);
        return 0;
    });
    return check_launch();
}

extern "C" int rvip_pack_subpixel_weights(const float* w, int cin, int cout, int dtype, void* w_phase, void* stream) {
    (void)hipGetLastError();
    if (!w || !w_phase || cin <= 0 || cout <= 0) return RVIP_EINVAL;
    const long long total = 16LL * cin * cout;
    const int blocks = (int)(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(pack_subpixel_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (bf16_t*)w_phase, 0);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(pack_subpixel_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (f16_t*)w_phase, 0);
    else if (dtype == RVIP_F32) hipLaunchKernelGGL(pack_subpixel_kernel<float>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (float*)w_phase, 0);
    else return RVIP_EINVAL;
    return check_launch();
}

extern "C" int rvip_pack_subpixel_dgrad_weights(const float* w, int cin, int cout, int dtype, void* w_phase, void* stream) {
    (void)hipGetLastError();
    if (!w || !w_phase || cin <= 0 || cout <= 0) return RVIP_EINVAL;
    const long long total = 16LL * cin * cout;
    const int blocks = (int)(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RVIP_BF16) hipLaunchKernelGGL(pack_subpixel_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (bf16_t*)w_phase, 1);
    else if (dtype == RVIP_F16) hipLaunchKernelGGL(pack_subpixel_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (f16_t*)w_phase, 1);
    else if (dtype == RVIP_F32) hipLaunchKernelGGL(pack_subpixel_kernel<float>, dim3(blocks), dim3(256), 0, s, w, cin, cout, (float*)w_phase, 1);
    else return RVIP_EINVAL;
    return check_launch();
}

// Host-side check of a pack table BEFORE it is uploaded (the launch below only sees the device copy): every entry must describe
// a kernel the table-driven re-layout and the conv kernels behind it take.
extern "C" int rvip_pack_table_check(const rvip_pack_entry* host_table, int entries, int dtype) {
    static_assert(sizeof(PackEntry) == sizeof(rvip_pack_entry), "the kernel's view of a table entry is the ABI's");
    if (!host_table || entries <= 0 || !RVIP_DT_OK(dtype)) return RVIP_EINVAL;
    const PackEntry* t = (const PackEntry*)host_table;
    for (int i = 0; i < entries; ++i) {
        const PackEntry& e = t[i];
        if (e.cin <= 0 || e.cout <= 0 || e.w_off < 0 || e.f_off < 0 || e.d_off < 0) return RVIP_EINVAL;
        if (e.reserved != 0 && e.reserved != 1) return RVIP_EINVAL;
        if (e.reserved == 0 && e.taps != 9 && e.taps != 27 && e.taps != 0) return RVIP_EINVAL;
        if (e.reserved == 1 && e.taps != 9 && e.taps != 0) return RVIP_EINVAL;          // phase kernels exist for 3x3 kernels only
        if ((e.w_off | e.f_off | e.d_off) & 3) return RVIP_EINVAL;                       // 16-byte rows start 16-byte aligned
        if (e.cout & 3) return RVIP_EINVAL;                                              // no conv kernel of this library takes such a layer
    }
    return RVIP_OK;
}

static int pack_all_launch(const float* theta, const void* table, int entries, int max_elems, int dtype,
                           void* wf_base, void* wd_base, uint32_t* tick, void* stream) {
    (void)hipGetLastError();
    if (!theta || !table || entries <= 0 || max_elems <= 0 || !wf_base || !wd_base || !RVIP_DT_OK(dtype)) return RVIP_EINVAL;
    long long nb = cdiv(max_elems, 32 * 64);                   // tiles of the largest kernel; smaller ones leave blocks idle
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    dim3 grid((unsigned)nb, (unsigned)entries);
    hipStream_t s = (hipStream_t)stream;
    by_dtype(dtype, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(pack_all_kernel<T>, grid, dim3(256), 0, s, theta, (const PackEntry*)table, (T*)wf_base, (T*)wd_base, tick);
        return 0;
    });
    return check_launch();
}
extern "C" int rvip_pack_all_conv3x3_weights(const float* theta, const void* table, int entries, int max_elems, int dtype,
                                             void* wf_base, void* wd_base, void* stream) {
    return pack_all_launch(theta, table, entries, max_elems, dtype, wf_base, wd_base, nullptr, stream);
}
extern "C" int rvip_pack_all_conv3x3_weights_tick(const float* theta, const void* table, int entries, int max_elems, int dtype,
                                                  void* wf_base, void* wd_base, uint32_t* state, void* stream) {
    if (!state) return RVIP_EINVAL;
    return pack_all_launch(theta, table, entries, max_elems, dtype, wf_base, wd_base, state, stream);
}
#endif  /* RVIP_KERNELS_ONLY */
