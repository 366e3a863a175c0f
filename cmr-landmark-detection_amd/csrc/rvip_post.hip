// Post-threshold of the predicted heat-maps on the device: the step right after the hot path.
//   flat label map   predict_model.py:149-156 / evaluate_cv.py (preds_flat): 0, then k+1 where pred[..., k] > thr, later
//                    channels overriding earlier ones
//   CC filter        Postprocess.py:108-120 clean_3d_prediction_2d_cc: per slice and label keep the largest 4-connected
//                    component (cv2.connectedComponentsWithStats(mask, 4); ties -> the component met first in raster order)
//   landmark         evaluate_cv.py:418-442 get_mean_rvip_2d: mean (y, x) of the label's pixels
// Both reference functions take `np.unique(x)[1:]` as "the labels without background": on a slice WITHOUT any background
// pixel that drops the smallest label present instead - the CC filter then erases that label, and get_mean_rvip_2d
// returns None for the smallest label of a (cleaned) slice that has no zero.  Reproduced here (flags per slice).
// One workgroup per (slice, label): labels live in global memory (L2-resident, H*W ints), propagated by alternating
// forward / backward raster sweeps of per-thread pixel runs until nothing changes.  Not a throughput kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rvip_hip.h"
#include "rvip_common.h"

namespace rvip {

__global__ __launch_bounds__(256) void post_flat_kernel(const float* __restrict__ pred, uint8_t* __restrict__ flat,
                                                        int* __restrict__ lab, int* __restrict__ present, long long npix, int hw, int k, float thr) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= npix) return;
    int v = 0;
    for (int c = 0; c < k; ++c) if (pred[p * k + c] > thr) v = c + 1;
    flat[p] = (uint8_t)v;
    const long long img = p / hw;
    present[img * (k + 1) + v] = 1;                     // value v occurs on this slice (idempotent store; zeroed by the host call)
    const int q = (int)(p - img * hw);
    for (int c = 0; c < k; ++c) lab[(img * k + c) * hw + q] = (v == c + 1) ? q + 1 : 0;
}

// grid = n * k workgroups of 1024 threads
__global__ __launch_bounds__(1024) void post_cc_kernel(uint8_t* __restrict__ flat, int* __restrict__ lab, int* __restrict__ cnt,
                                                       float* __restrict__ points, int* __restrict__ sizes,
                                                       const int* __restrict__ present, int* __restrict__ erased,
                                                       int h, int w, int k, int cc_filter) {
    __shared__ int changed;
    __shared__ int best_cnt, best_lab;
    __shared__ float red[3][16];
    const int hw = h * w, tid = threadIdx.x;
    const int img = blockIdx.x / k, c = blockIdx.x % k;
    int* L = lab + (size_t)blockIdx.x * hw;
    int* Cn = cnt + (size_t)blockIdx.x * hw;
    uint8_t* F = flat + (size_t)img * hw;
    const int run = (hw + 1023) / 1024, p0 = tid * run, p1 = (p0 + run < hw) ? p0 + run : hw;
    // np.unique(s)[1:] of the CC filter: without background on the slice the smallest label present is skipped (= erased)
    bool skipped = false;
    if (cc_filter && !present[img * (k + 1)]) {
        int first = 1;
        while (first <= k && !present[img * (k + 1) + first]) ++first;
        skipped = first == c + 1;
    }
    if (cc_filter && !skipped) {
        for (int sweep = 0;; ++sweep) {
            if (tid == 0) changed = 0;
            __syncthreads();
            bool ch = false;
            // alternate the direction so that a label also travels against the raster order within one sweep
            for (int i = 0; i < p1 - p0; ++i) {
                const int p = (sweep & 1) ? p1 - 1 - i : p0 + i;
                int l = __hip_atomic_load(&L[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!l) continue;
                const int y = p / w, x = p - y * w;
                int m = l;
                if (x > 0)     { const int t = __hip_atomic_load(&L[p - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (t && t < m) m = t; }
                if (x + 1 < w) { const int t = __hip_atomic_load(&L[p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (t && t < m) m = t; }
                if (y > 0)     { const int t = __hip_atomic_load(&L[p - w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (t && t < m) m = t; }
                if (y + 1 < h) { const int t = __hip_atomic_load(&L[p + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (t && t < m) m = t; }
                // a label is the raster index + 1 of a pixel of the same component: follow it once (pointer jumping)
                { const int t = __hip_atomic_load(&L[m - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (t && t < m) m = t; }
                if (m < l) { __hip_atomic_store(&L[p], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); ch = true; }
            }
            if (ch) changed = 1;
            __syncthreads();
            const int any = changed;
            __syncthreads();
            if (!any) break;
        }
        // component sizes: labels are 1 + the smallest raster index of the component
        for (int p = p0; p < p1; ++p) Cn[p] = 0;
        if (tid == 0) { best_cnt = 0; best_lab = 0x7fffffff; }
        __syncthreads();
        for (int p = p0; p < p1; ++p) { const int l = L[p]; if (l) atomicAdd(&Cn[l - 1], 1); }
        __syncthreads();
        int bc = 0;
        for (int p = p0; p < p1; ++p) bc = Cn[p] > bc ? Cn[p] : bc;
        atomicMax(&best_cnt, bc);
        __syncthreads();
        for (int p = p0; p < p1; ++p) if (Cn[p] == best_cnt && best_cnt > 0) atomicMin(&best_lab, p + 1);   // first in raster order
        __syncthreads();
    }
    // clean the flat map and reduce the centroid of what is kept
    float sy = 0.f, sx = 0.f, sn = 0.f;
    bool gone = false;
    for (int p = p0; p < p1; ++p) {
        if (F[p] != (uint8_t)(c + 1)) continue;
        const bool keep = !cc_filter || (!skipped && L[p] == best_lab);
        if (!keep) { F[p] = 0; gone = true; continue; }
        const int y = p / w;
        sy += (float)y; sx += (float)(p - y * w); sn += 1.f;
    }
    sy = wave_sum(sy); sx = wave_sum(sx); sn = wave_sum(sn);
    if ((tid & 63) == 0) { red[0][tid >> 6] = sy; red[1][tid >> 6] = sx; red[2][tid >> 6] = sn; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f, n = 0.f;
        for (int i = 0; i < 16; ++i) { a += red[0][i]; b += red[1][i]; n += red[2][i]; }
        const float nanv = __builtin_nanf("");
        points[(size_t)blockIdx.x * 2 + 0] = n > 0.f ? a / n : nanv;
        points[(size_t)blockIdx.x * 2 + 1] = n > 0.f ? b / n : nanv;
        sizes[blockIdx.x] = (int)n;
    }
    if (gone) erased[blockIdx.x] = 1;                   // the cleaned slice has background pixels now (idempotent store)
}

// get_mean_rvip_2d's np.unique(nda_2d)[1:]: on a (cleaned) slice without any zero the smallest label present has no point
__global__ void post_points_quirk(float* __restrict__ points, int* __restrict__ sizes_unused, const int* __restrict__ present,
                                  const int* __restrict__ erased, const int* __restrict__ sizes, int n, int k) {
    const int img = blockIdx.x * 64 + threadIdx.x;
    if (img >= n) return;
    bool zero = present[img * (k + 1)] != 0;
    for (int c = 0; c < k; ++c) zero = zero || erased[img * k + c];
    if (zero) return;
    for (int c = 0; c < k; ++c) {
        if (sizes[img * k + c] > 0) {
            points[(size_t)(img * k + c) * 2 + 0] = __builtin_nanf("");
            points[(size_t)(img * k + c) * 2 + 1] = __builtin_nanf("");
            return;
        }
    }
}

}  // namespace rvip

using namespace rvip;

extern "C" size_t rvip_postprocess_workspace(int n, int h, int w, int k) {
    return ((size_t)2 * n * k * h * w + (size_t)n * (2 * k + 1)) * sizeof(int);
}

extern "C" int rvip_postprocess(const float* pred, uint8_t* flat, float* points, int* sizes, int n, int h, int w, int k, float thr,
                                int cc_filter, void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!pred || !flat || !points || !sizes || !workspace || n <= 0 || h <= 0 || w <= 0 || k <= 0 || k > 254) return RVIP_EINVAL;
    if ((long long)h * w >= (1LL << 30) || (long long)n * k >= (1LL << 30)) return RVIP_EINVAL;
    if (workspace_bytes < rvip_postprocess_workspace(n, h, w, k)) return RVIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int* lab = (int*)workspace;
    int* cnt = lab + (size_t)n * k * h * w;
    int* present = cnt + (size_t)n * k * h * w;         // [n][k+1]
    int* erased = present + (size_t)n * (k + 1);        // [n][k]
    const long long npix = (long long)n * h * w;
    if (hipMemsetAsync(present, 0, (size_t)n * (2 * k + 1) * sizeof(int), s) != hipSuccess) return RVIP_ELAUNCH;
    hipLaunchKernelGGL(post_flat_kernel, dim3((unsigned)cdiv(npix, 256)), dim3(256), 0, s, pred, flat, lab, present, npix, h * w, k, thr);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(post_cc_kernel, dim3((unsigned)(n * k)), dim3(1024), 0, s, flat, lab, cnt, points, sizes, present, erased, h, w, k, cc_filter ? 1 : 0);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(post_points_quirk, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, s, points, nullptr, present, erased, sizes, n, k);
    return check_launch();
}
