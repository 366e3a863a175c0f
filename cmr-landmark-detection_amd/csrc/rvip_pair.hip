// Weight gradient and data gradient of one 3x3 layer in ONE launch (round 5).
//
// Both kernels of a layer's backward pass read the same gradient tensor dz and neither depends on the other (autodiff of Conv2D,
// KerasLayers.py:683,689).  Each is a persistent one-workgroup-per-CU kernel that owns its CU (LDS, registers), and each pays fixed
// costs nothing overlaps when it runs alone: launch, the first tile's DMA, the epilogue / slab store at the end, the kernel boundary.
// Side by side on a PARTITION of the CUs (rvip_*_desc.cu_limit) the fixed costs of one hide behind the other's matrix work -- but as
// two launches on two streams every layer pays a fork and a join of the captured graph, ~18 us of cross-queue synchronisation
// (profiles/r05_bwd_pair.txt).  Here the two run as the two halves of one grid: workgroups [0, nw) execute the body of wgrad3x3_ws,
// workgroups [nw, nw + nd) the body of conv3x3_igemm_ws16.  Both halves must be workgroups of one shape: 768 threads at <= 168 VGPRs
// for the 512-pixel tiles (the igemm's eight-compute-wave form; the weight gradient then splits its nine taps over two waves, TS = 2:
// 80 / 64 accumulator registers instead of 144), 512 threads for the 256-pixel tiles of the 16-wide maps.  Workgroups are dealt to the 8 XCDs round-robin by their linear id, so with nw a
// multiple of 8 both halves keep the placement their own launches have.  The arithmetic of either half is untouched (the tensors are
// and partial rows are bit-identical to the two launches with the same cu_limit).
// MEASURED (round 5, same box, captured step): one after the other 4.62 ms, two launches between a fork and a join 4.60, this 4.52.
// What the pair hides is the fixed cost of the two launches (~10 us per layer on the MFMA-bound layers); on the HBM-bound 256^2
// layers the two halves share the bandwidth and the pair takes as long as the two launches did.  (A first form with 512-thread
// halves -- the igemm's four-compute-wave tiling -- measured 4.66: that tiling is ~16 % slower on the MFMA-bound layers.)
#define RVIP_KERNELS_ONLY
#include "rvip_conv.hip"
#include "rvip_wgrad.hip"
#undef RVIP_KERNELS_ONLY

namespace rvip {

// NCW = compute waves of BOTH halves' workgroups: 8 (768 threads, <= 168 VGPRs: the igemm's eight-compute-wave form of the 512-pixel
// tiles + the weight gradient with its taps split over two waves, TS = 2) or 4 (512 threads: the forms the 256-pixel tiles of the
// 16-wide maps use anyway)
// TAPS = 4: the sub-pixel forms of an UpSampling2D -> conv layer (four-phase weight gradient, data gradient with subpix = 2);
// PBW = 1: the weight gradient's phase-PAIR form (both column phases per workgroup) beside the nine-tap data gradient with its 2x2 sums
template <typename T, int TW, int CIB, int COB, int NSTW, int NCT, int NPIX, int STATS, int NCW, int TAPS, int PBW = 0>
__global__ __launch_bounds__((NCW + 4) * 64, 1) void wgrad_dgrad_pair(WgArgs2 wa, ConvArgs2 ca, int nw, int wgx, int wgy, int dgx) {
    if ((int)blockIdx.x < nw) {
        const unsigned id = blockIdx.x, r = id / (unsigned)wgx;
        wgrad3x3_ws_body<T, TW, CIB, COB, NSTW, PBW ? 4 : TAPS, PBW, NCW / 4>(wa, id % (unsigned)wgx, r % (unsigned)wgy, r / (unsigned)wgy, (unsigned)wgx);
    } else {
        const int id = (int)blockIdx.x - nw;
        igemm_ws16_body<T, TW, NCT, NPIX, STATS, TAPS, NCW>(ca, id % dgx, id / dgx, 0, dgx);
    }
}

template <typename T, int TW, int CIB, int COB, int NCT, int NPIX, int STATS, int NCW, int TAPS = 9, int PBW = 0>
static int launch_pair(const WgArgs2& wa, const IgemmPlan& dp, hipStream_t s, bool dry) {
    // LDS of the weight-gradient half: launch_wgrad2x's arithmetic (the stage count is a template parameter of the body)
    constexpr int TH = 256 / TW;
    constexpr int NHROWS = ((TW + 2) * (TH + 2) + 15) / 16 * 16;
    constexpr int ST = NHROWS * CIB * 2 + (PBW ? 2 : 1) * 256 * COB * 2;
    constexpr int NSTW = 3 * ST <= 160 * 1024 ? 3 : 2;
    constexpr int FOLD = (CIB / 32) * (COB / 32) < 4 ? 4 * (PBW ? 2 * 4 : TAPS) * 32 * 32 * 4 : 0;
    constexpr int lds_w = NSTW * ST > FOLD ? NSTW * ST : FOLD;
    static_assert(lds_w <= 160 * 1024, "LDS");
    if (dp.gz != 1 || dp.lds > 160 * 1024) return RVIP_EUNSUPPORTED;
    if (dry) return RVIP_OK;
    const int lds = lds_w > dp.lds ? lds_w : dp.lds;
    auto kern = &wgrad_dgrad_pair<T, TW, CIB, COB, NSTW, NCT, NPIX, STATS, NCW, TAPS, PBW>;
    static std::atomic<bool> attr_done{false};
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return RVIP_ELAUNCH; }
        attr_done = true;
    }
    const int wgx = wa.nsplit, wgy = (int)cdiv(wa.cin, CIB), wgz = (int)cdiv(wa.cout, COB);
    const int nw = wgx * wgy * wgz, nd = dp.gx * dp.gy;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nw + nd)), dim3((NCW + 4) * 64), lds, s, wa, dp.args, nw, wgx, wgy, dp.gx);
    return check_launch();
}

// the data-gradient half: the tiling dispatch_igemm_ws takes for this shape, in its four-compute-wave (512-thread) instantiation
template <typename T, int TW, int NCT, int NPIX, int NCW, int TAPS = 9>
static int plan_dgrad(const ConvArgs& a, float* sums, IgemmPlan& p) {
    bool used = false;
    const int rc = launch_igemm_ws<T, true, TW, NCT, NPIX, TAPS, NCW, true>(a, nullptr, used, sums, nullptr, false, 2, &p);
    if (rc) return rc;
    return used ? RVIP_OK : RVIP_EUNSUPPORTED;
}

template <typename T>
static int pair_dispatch(const rvip_wgrad3x3_desc* wd, const rvip_conv3x3_desc* dd, float* sums_ws, size_t sums_ws_bytes, hipStream_t s, bool dry) {
    WgradPlan wp;
    int rc = wgrad_plan(wd, wp);
    if (rc) return rc;
    ConvArgs a;
    rc = conv_args_from_desc(dd, a);
    if (rc) return rc;
    const bool sub = wp.b.sp == 1;                          // four-phase weight gradient <-> data gradient in sub-pixel form (subpix = 2)
    const bool pbw = wp.b.sp == 2;                          // phase-pair weight gradient <-> nine-tap data gradient with the 2x2 sums (down2)
    if (sub != (a.subpix == 2) || a.subpix == 1 || (pbw && !a.down2)) return RVIP_EUNSUPPORTED;
    if ((!sub && a.kd != 1) || a.c1 || a.up0 || dd->dtype != wd->dtype || !sums_ws) return RVIP_EUNSUPPORTED;
    if (dd->n != wd->n || dd->h != wd->h || dd->w != wd->w || a.cin != wd->cout || dd->x0 != wd->dy) return RVIP_EINVAL;      // the same layer, the same dz
    if (dd->bias || dd->act != RVIP_ACT_NONE) return RVIP_EINVAL;
    const bool gated = dd->mask_bits != nullptr;
    if (dd->sign_bits) return RVIP_EINVAL;
    if (gated) {
        if (dd->mask_channels <= 0 || (dd->mask_channels % 32 && dd->mask_channels != dd->cout) || dd->mask_channels > dd->cout || dd->cout % 8 || a.down2 || !(dd->mask_scale > 0.f)) return RVIP_EINVAL;
        a.mbits = dd->mask_bits; a.mbits_c = dd->mask_channels; a.mscale = dd->mask_scale;
    }
    // the tile / channel-column choice of dispatch_igemm_ws
    bool two = a.cout > 32;
    if (two) {
        const int tpx = (a.w > 16 && a.h >= 16) ? 512 : 256, tw = a.w > 16 ? 32 : 16;
        const long long ntiles = (long long)a.n * cdiv(a.w, tw) * cdiv(a.h, tpx / tw);
        if (ntiles * cdiv(a.cout, 64) <= a.cus / 2) two = false;
    }
    if (sub) two = a.cout > 32;                            // (dispatch_igemm_ws: the sub-pixel forms keep 64-channel columns)
    if (sub && gated) return RVIP_EUNSUPPORTED;
    const int tw = wp.g.tw, cib = wp.g.cib, cob = wp.g.cob;
    if (a.w > 16 && a.h < 16) return RVIP_EUNSUPPORTED;    // (32-wide, 256-pixel tiles: no layer of the benchmark graphs)
    if ((a.w > 16 ? 32 : 16) != tw) return RVIP_EUNSUPPORTED;
    IgemmPlan dp;
#define RVIP_PAIR(TWv, CIBv, COBv, NCTv, NPIXv)                                                                                   \
    if (!sub && !pbw && tw == TWv && cib == CIBv && cob == COBv && two == (NCTv == 2)) {                                                            \
        rc = plan_dgrad<T, TWv, NCTv, NPIXv, (NPIXv == 512 ? 8 : 4)>(a, sums_ws, dp);                                              \
        if (rc) return rc;                                                                                                         \
        if (sums_ws_bytes < (size_t)dp.gx * a.cout * sizeof(float)) return RVIP_EWORKSPACE;                                        \
        return gated ? launch_pair<T, TWv, CIBv, COBv, NCTv, NPIXv, 3, (NPIXv == 512 ? 8 : 4)>(wp.b, dp, s, dry)                   \
                     : launch_pair<T, TWv, CIBv, COBv, NCTv, NPIXv, 2, (NPIXv == 512 ? 8 : 4)>(wp.b, dp, s, dry);                  \
    }
    // the (tile width, weight-gradient block, channel-column) combinations of the benchmark graphs (configs 2 and 4)
    RVIP_PAIR(32, 32, 32, 1, 512)
    RVIP_PAIR(32, 32, 32, 2, 512)
    RVIP_PAIR(32, 32, 64, 1, 512)
    RVIP_PAIR(32, 64, 64, 1, 512)
    RVIP_PAIR(32, 64, 64, 2, 512)
    RVIP_PAIR(16, 64, 64, 1, 256)
    RVIP_PAIR(16, 64, 64, 2, 256)
#undef RVIP_PAIR
#define RVIP_PAIR4(TWv, NCTv, NPIXv)                                                                                                \
    if (sub && tw == TWv && cib == 64 && cob == 64 && two == (NCTv == 2)) {                                                         \
        rc = plan_dgrad<T, TWv, NCTv, NPIXv, (NPIXv == 512 ? 8 : 4), 4>(a, sums_ws, dp);                                           \
        if (rc) return rc;                                                                                                         \
        if (sums_ws_bytes < (size_t)dp.gx * a.cout * sizeof(float)) return RVIP_EWORKSPACE;                                        \
        return launch_pair<T, TWv, 64, 64, NCTv, NPIXv, 2, (NPIXv == 512 ? 8 : 4), 4>(wp.b, dp, s, dry);                           \
    }
    RVIP_PAIR4(32, 2, 512)
    RVIP_PAIR4(16, 2, 256)
#undef RVIP_PAIR4
    if (pbw && !gated && tw == 32 && cib == 64 && cob == 32 && two) {           // the 64 -> 32 up-conv at the full resolution
        rc = plan_dgrad<T, 32, 2, 512, 8>(a, sums_ws, dp);
        if (rc) return rc;
        if (sums_ws_bytes < (size_t)dp.gx * a.cout * sizeof(float)) return RVIP_EWORKSPACE;
        return launch_pair<T, 32, 64, 32, 2, 512, 2, 8, 9, 1>(wp.b, dp, s, dry);
    }
    return RVIP_EUNSUPPORTED;
}

}  // namespace rvip

using namespace rvip;

static int pair_entry(const rvip_wgrad3x3_desc* wd, const rvip_conv3x3_desc* dd, float* sums_ws, size_t sums_ws_bytes, hipStream_t s, bool dry) {
    if (!wd || !dd) return RVIP_EINVAL;
    if (wd->dtype == RVIP_BF16) return pair_dispatch<bf16_t>(wd, dd, sums_ws, sums_ws_bytes, s, dry);

    if (wd->dtype == RVIP_F16) return pair_dispatch<f16_t>(wd, dd, sums_ws, sums_ws_bytes, s, dry);

    return RVIP_EUNSUPPORTED;
}

// 1 if rvip_conv3x3_wgrad_dgrad serves this pair of descriptors (same layer, same dz, both with their cu_limit set), else 0
extern "C" int rvip_conv3x3_wgrad_dgrad_ok(const rvip_wgrad3x3_desc* wd, const rvip_conv3x3_desc* dd) {
    float dummy;
    return pair_entry(wd, dd, &dummy, (size_t)1 << 40, nullptr, true) == RVIP_OK ? 1 : 0;
}

// rvip_conv3x3_wgrad(wd) and rvip_conv3x3_fwd_sums(dd, sums_ws) as one launch (+ the weight gradient's slab fold behind it)
extern "C" int rvip_conv3x3_wgrad_dgrad(const rvip_wgrad3x3_desc* wd, const rvip_conv3x3_desc* dd, float* sums_ws, size_t sums_ws_bytes, void* stream) {
    (void)hipGetLastError();
    hipStream_t s = (hipStream_t)stream;
    const int rc = pair_entry(wd, dd, sums_ws, sums_ws_bytes, s, false);
    if (rc) return rc;
    return wgrad_finish(wd, rvip_conv3x3_wgrad_splits(wd), rvip_conv3x3_wgrad_form(wd), s);
}
