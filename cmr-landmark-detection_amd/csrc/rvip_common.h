// Shared device helpers for the gfx950 kernels (wave64, MFMA, 16-byte channel vectors).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rvip_hip.h"

namespace rvip {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct bf16_t { uint16_t bits; };
struct f16_t { uint16_t bits; };      // IEEE binary16 storage (RVIP_F16): same kernels, v_mfma_f32_32x32x16_f16, fp32 accumulate

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }   // RNE, NaN kept

__device__ __forceinline__ float f16_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
// RNE of the FP32 VALUE, overflow -> inf.  The empty asm pins that value: without it the compiler may fold a preceding fma into
// v_fma_mixlo_f16, which rounds the exact product-sum ONCE -- a different result in rare double-rounding cases, and then the
// fused and the separate kernels (and the storage-rounding emulation in tests/) no longer store the same bits.
// IEEE half SATURATES here (one v_med3_f32 in front of the convert) instead of overflowing to infinity: with the moving statistics of the
// first epochs an evaluation pass can compound a factor per BatchNormalization over 18 layers past 65504, and one infinity turns the
// next layer's sums into NaN (tools/soak_probe.py: val_loss 93.7, 6.9, nan, 0.15 ...).  A saturated value stays a large finite number
// in the fp32 accumulators behind it.
__device__ __forceinline__ float sat_f16(float f) { return __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { f = sat_f16(f); asm volatile("" : "+v"(f)); return __builtin_bit_cast(uint16_t, (_Float16)f); }

// two fp32 -> one packed 32-bit word (lo = a, hi = b), RNE: ONE v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32.  Converting the halves one by
// one and OR-ing them (the obvious form) is four instructions on gfx950: cvt, cvt, shift, or.
typedef float rvip_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rvip_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 rvip_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    const rvip_f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rvip_bf16x2)); }
__device__ __forceinline__ uint32_t pack2_f16(float a, float b) {
    a = sat_f16(a); b = sat_f16(b);
    asm volatile("" : "+v"(a), "+v"(b));              // see f32_to_f16
    const rvip_f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rvip_f16x2)); }

// 16-byte channel vector: VE elements of T
typedef unsigned rvip_u32x4 __attribute__((ext_vector_type(4)));
typedef float rvip_f32x4 __attribute__((ext_vector_type(4)));
// 16-byte store of a tensor the NEXT launch (or a later one) reads: write-through (sc1).  A kernel boundary writes back every dirty
// line of the eight L2s before the dependent launch may start (MI355X_MICROARCH.md, 'boundary': + B / 6 TB/s for B dirty bytes), and
// every launch of the step ends with all its workgroups storing at once; written through, the bytes leave L2 while the launch still
// runs.  Measured on the captured step, same box, alternating (round 5): igemm epilogues 4.427 -> 4.361 ms, + every element-wise
// pass 4.322 ms (-2.4 %); the 16-byte slab stores of the weight gradient on top: no change (left plain / non-temporal).
#define RVIP_WT_AUX 16            /* raw_buffer_store aux: bit 4 = sc1 */
__device__ __forceinline__ void store16(void* p, rvip_u32x4 v) {
    // s_nop 1: a store of more than 64 bits reads its data registers late; a VALU write to them within the next two wait states
    // corrupts it (the compiler pads its own stores, it cannot see into this statement: found as wrong fp16 skip tensors)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int VE = 4;
    // last use of a streamed tensor: do not keep the lines (nt = non-temporal)
    __device__ static __forceinline__ void load_nt(const void* p, float (&v)[4]) {
        const rvip_f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const rvip_f32x4*>(p)); v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
    __device__ static __forceinline__ void load(const void* p, float (&v)[4]) {
        float4 r = *reinterpret_cast<const float4*>(p); v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
    __device__ static __forceinline__ void store(void* p, const float (&v)[4]) {
        const rvip_f32x4 f = {v[0], v[1], v[2], v[3]}; store16(p, __builtin_bit_cast(rvip_u32x4, f)); }
    __device__ static __forceinline__ float round(float x) { return x; }
};
template <> struct Vec<bf16_t> {
    static constexpr int VE = 8;
    __device__ static __forceinline__ void load_nt(const void* p, float (&v)[8]) {
        const rvip_u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const rvip_u32x4*>(p));
        uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); } }
    __device__ static __forceinline__ void load(const void* p, float (&v)[8]) {
        uint4 r = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); } }
    __device__ static __forceinline__ void store(void* p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack2_bf16(v[2 * i], v[2 * i + 1]);
        { const rvip_u32x4 u = {w[0], w[1], w[2], w[3]}; store16(p, u); } }
    __device__ static __forceinline__ float round(float x) { return bf16_to_f32(f32_to_bf16(x)); }
    __device__ static __forceinline__ uint16_t enc(float x) { return f32_to_bf16(x); }
    __device__ static __forceinline__ uint32_t pack2(float a, float b) { return pack2_bf16(a, b); }
    __device__ static __forceinline__ float lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }           // halves of a packed word, widened
    __device__ static __forceinline__ float hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
    __device__ static __forceinline__ float dec(uint16_t b) { return bf16_to_f32(b); }
};
template <> struct Vec<f16_t> {
    static constexpr int VE = 8;
    __device__ static __forceinline__ void unpack(const uint32_t (&w)[4], float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = f16_to_f32((uint16_t)(w[i] & 0xffffu)); v[2 * i + 1] = f16_to_f32((uint16_t)(w[i] >> 16)); } }
    __device__ static __forceinline__ void load_nt(const void* p, float (&v)[8]) {
        const rvip_u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const rvip_u32x4*>(p));
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        unpack(w, v); }
    __device__ static __forceinline__ void load(const void* p, float (&v)[8]) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        unpack(w, v); }
    __device__ static __forceinline__ void store(void* p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack2_f16(v[2 * i], v[2 * i + 1]);
        { const rvip_u32x4 u = {w[0], w[1], w[2], w[3]}; store16(p, u); } }
    __device__ static __forceinline__ float round(float x) { return f16_to_f32(f32_to_f16(x)); }
    __device__ static __forceinline__ uint16_t enc(float x) { return f32_to_f16(x); }
    __device__ static __forceinline__ uint32_t pack2(float a, float b) { return pack2_f16(a, b); }
    __device__ static __forceinline__ float lo(uint32_t p) { return (float)__builtin_bit_cast(rvip_f16x2, p)[0]; }
    __device__ static __forceinline__ float hi(uint32_t p) { return (float)__builtin_bit_cast(rvip_f16x2, p)[1]; }
    __device__ static __forceinline__ float dec(uint16_t b) { return f16_to_f32(b); }
};
// MFMA on eight 16-bit K elements per lane (one uint4 fragment of each operand)
template <typename T> __device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& acc);
template <> __device__ __forceinline__ f32x16 mfma16<bf16_t>(const uint4& a, const uint4& b, const f32x16& acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma16<f16_t>(const uint4& a, const uint4& b, const f32x16& acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0); }
// 16 x 16 x 32 shape: the same eight K elements per lane, lane = (row mod 16, 16-byte slot of the 32-element K row)
template <typename T> __device__ __forceinline__ f32x4 mfma16x16(const uint4& a, const uint4& b, const f32x4& acc);
template <> __device__ __forceinline__ f32x4 mfma16x16<bf16_t>(const uint4& a, const uint4& b, const f32x4& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0); }
template <> __device__ __forceinline__ f32x4 mfma16x16<f16_t>(const uint4& a, const uint4& b, const f32x4& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0); }

// ReLU as ONE v_max_f32.  fmaxf() costs two -- the compiler first quiets a possible signalling NaN (v_max_f32 v, v, v), which no value
// computed on the device is; for every other input, quiet NaN included (-> 0), the value is fmaxf's.  The epilogues and the VALU-bound
// passes (first layer, head) run 64 of these per lane and tile.
__device__ __forceinline__ float relu1(float t) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(t));
    return r;
}
__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case RVIP_ACT_RELU: return relu1(x);
        case RVIP_ACT_ELU: return x > 0.f ? x : expm1f(x);
        case RVIP_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        default: return x;
    }
}
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_bwd(float y, int act) {
    switch (act) {
        case RVIP_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case RVIP_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        case RVIP_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ---- counter-based dropout stream (shared by forward and backward, and mirrored on the host in
// dropout_stream.py): one 32-bit hash per PAIR of consecutive elements, 16 bits each.
__host__ __device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__host__ __device__ __forceinline__ uint32_t dropout_key(uint32_t seed, uint32_t step, uint32_t layer_id) {
    return hash32(seed ^ hash32(step * 0x9E3779B9u + layer_id * 0x85EBCA6Bu + 0x27d4eb2fu));
}
__host__ __device__ __forceinline__ uint32_t dropout_thr(float rate) {       // keep iff bits16 < thr
    float keep = 1.f - rate;
    uint32_t t = (uint32_t)(keep * 65536.f + 0.5f);
    return t > 65536u ? 65536u : t;
}
// keep flags of the VE elements starting at element index e0 (e0 % VE == 0, VE even)
template <int VE>
__device__ __forceinline__ void dropout_keep(uint32_t key, unsigned long long e0, uint32_t thr, bool (&keep)[VE]) {
#pragma unroll
    for (int i = 0; i < VE / 2; ++i) {
        uint32_t h = hash32((uint32_t)((e0 >> 1) + i) ^ key);
        keep[2 * i] = (h & 0xffffu) < thr;
        keep[2 * i + 1] = (h >> 16) < thr;
    }
}

// values of lane ^ 1 / lane ^ 2 through DPP quad permutes (__shfl_xor compiles to ds_bpermute_b32, a trip through the LDS crossbar)
__device__ __forceinline__ float lane_xor1_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_xor2_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

extern int g_last_hip_error;
inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
    return RVIP_OK;
}

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// dtype tags of the ABI -> storage types
#define RVIP_DT_OK(dt) ((dt) == RVIP_BF16 || (dt) == RVIP_F16 || (dt) == RVIP_F32)
#define RVIP_VE(dt) ((dt) == RVIP_F32 ? 4 : 8)
#define RVIP_ESZ(dt) ((dt) == RVIP_F32 ? 4 : 2)
template <typename F> static inline int by_dtype(int dt, F&& f) {
    if (dt == RVIP_BF16) return f(bf16_t{});
    if (dt == RVIP_F16) return f(f16_t{});
    return f(float{});
}

}  // namespace rvip
