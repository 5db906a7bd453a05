// ekf_capi.hip -- the C ABI of include/ekfslam.h over the gfx950 kernels (ekf_kernels.hip).
// Host runtime of the filter core: device pools, pinned staging, stream ordering, event timing.
// No CPU fallback exists: without a gfx950 device every entry point fails with EKF_ERR_NO_DEVICE.
#include "../../include/ekfslam.h"
#include "ekf_kernels.hpp"
#include "ekf_dense.hpp"
#include "ekf_sim.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

ekf_status fail(ekf_status st, const std::string& msg) {
    g_err = msg;
    return st;
}

#define HIPC(expr)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? EKF_ERR_NOMEM : EKF_ERR_HIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

#define EKFC(expr)                        \
    do {                                  \
        ekf_status s_ = (expr);           \
        if (s_ != EKF_OK) return s_;      \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// pinned host buffer whose last async use is guarded by an event
struct Staging {
    void* host = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;

    ekf_status reserve(size_t need) {
        if (need <= bytes) return EKF_OK;
        EKFC(wait());
        if (host) HIPC(hipHostFree(host));
        host = nullptr; bytes = 0;
        HIPC(hipHostMalloc(&host, need, hipHostMallocDefault));
        bytes = need;
        if (!ev) HIPC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        return EKF_OK;
    }
    ekf_status wait() {
        if (pending) { HIPC(hipEventSynchronize(ev)); pending = false; }
        return EKF_OK;
    }
    ekf_status mark(hipStream_t s) {
        HIPC(hipEventRecord(ev, s));
        pending = true;
        return EKF_OK;
    }
    void release() {
        if (host) (void)hipHostFree(host);
        if (ev) (void)hipEventDestroy(ev);
        host = nullptr; ev = nullptr; bytes = 0; pending = false;
    }
};

// ring of pinned staging buffers: an upload only waits for the copy issued kRing uploads ago, so the
// host keeps queueing work while the GPU is still busy with earlier calls
struct StagingRing {
    static constexpr int kRing = 8;
    Staging slot[kRing];
    int next = 0;
    Staging& acquire() {
        Staging& s = slot[next];
        next = (next + 1) % kRing;
        return s;
    }
    void release() { for (Staging& s : slot) s.release(); }
};

struct Pool {
    int device = -1;
    hipStream_t stream = nullptr;
    ekf::PoolView pv{};
    ekf::Rank2Tuning tuning{0, -1, 0};
    size_t dev_bytes = 0;
    int init_flag = 0;  // landmark_init_flag, ekf_slam.hpp:65

    // association / single-filter staging (device)
    double* scores = nullptr;    // [B][n]
    double* meas_dev = nullptr;  // [jcap][2]
    int* assoc_out_dev = nullptr;  // [jcap]
    int jcap = 0;
    double* sensor_dev = nullptr;  // [2n] (single filter)
    double* digest_dev = nullptr;  // [B][4]
    double* poses_dev = nullptr;   // [B][3]
    StagingRing stage_in;
    Staging stage_out;

    // uploaded known-association log (device) + per-(step, slot) active-filter counts (host)
    int T = 0, vmax = 0;
    double* log_twist = nullptr;
    int* log_lm = nullptr;
    double* log_z = nullptr;
    double* log_init = nullptr;
    double* log_truth = nullptr;  // [T][B][3], simulated logs only
    size_t log_bytes = 0;
    std::vector<int> slot_active;  // [T][vmax]

    // uploaded unknown-association log
    int uT = 0, ujmax = 0;
    double* ulog_twist = nullptr;  // [T][B][2]
    int* ulog_count = nullptr;     // [T][B]
    double* ulog_meas = nullptr;   // [T][B][jmax][2]
    int* ulog_assoc = nullptr;     // [T][B][jmax] decisions
    double* ulog_truth = nullptr;  // [T][B][3], simulated logs only
    int truth_is_unknown_log = 0;  // which simulated log ekf_batch_mc_stats refers to (the latest)
    unsigned long long* corr_counter = nullptr;
    std::vector<int> ucount_host;  // [T][B]
    size_t ulog_bytes = 0;

    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;

    // delayed rank-2k update (0 = eager): pending factor store + ping-pong state buffer
    double* Uf = nullptr;
    double* Vf = nullptr;
    double* state_alt = nullptr;
    int pend_cap = 0, pend_count = 0;
    int pend_symmetric = 0;
    int active_prefix = 1;  // data_association(): restrict corrections to the discovered prefix of the state
    int touched_hwm = 0;    // landmarks [0, touched_hwm) may carry non-constructor covariance (single filter)
    int small_path = 1;     // measurement() of a small map runs as one LDS-resident launch (ekf_small.hip)
    int active_set = 0;     // eager corrections stream only the rows of the touched set (opt-in)
    int touched_bound = 0;  // host-side upper bound of the device touch_count over the pool
    std::vector<unsigned char> host_touched;  // single filter: exact host copy of the touched flags
    std::vector<int> log_touch_bound;         // batch: bound after step t of the uploaded log
    int touch_bound_base = 0;                 // touched_bound when that log arrived (filters may not be fresh)
    unsigned char* visible_dev = nullptr;  // [n] (single filter)

    // fused single-launch correction (single filter): the second covariance / state buffer it writes into
    int fused = 1;
    double* sigma_alt = nullptr;
    double* state_fz = nullptr;
    bool alt_synced = false;  // sigma_alt equals sigma outside the region the next fused correction rewrites

    bool fused_ok() const { return fused && pv.B == 1 && pend_cap == 0 && !active_set; }
    ekf_status ensure_alt() {
        if (!sigma_alt) {
            EKFC(dalloc(&sigma_alt, (size_t)pv.B * pv.sigma_stride));
            EKFC(dalloc(&state_fz, (size_t)pv.B * pv.ld));
            alt_synced = false;
        }
        if (!alt_synced) {
            // Both buffers must agree wherever a (prefix-confined) correction does not write.  Anything that
            // rewrites Sigma in place outside this path clears alt_synced; prediction() needs no copy -- beyond
            // the discovered prefix it maps zeros to zeros, inside it the next fused correction rewrites all.
            HIPC(hipMemcpyAsync(sigma_alt, pv.sigma, sizeof(double) * pv.B * pv.sigma_stride, hipMemcpyDeviceToDevice, stream));
            alt_synced = true;
        }
        return EKF_OK;
    }

    ekf::Pending pending() const { return ekf::Pending{Uf, Vf, pend_cap, pend_count, pend_symmetric}; }

    ekf_status set_update_mode(int max_pending_corrections, int symmetric_gather) {
        EKFC(use());
        EKFC(flush());
        pend_symmetric = symmetric_gather ? 1 : 0;
        HIPC(hipStreamSynchronize(stream));
        for (double** p : {&Uf, &Vf, &state_alt})
            if (*p) { HIPC(hipFree(*p)); *p = nullptr; }
        pend_cap = 0;
        if (max_pending_corrections <= 0) return EKF_OK;
        int cap = 2 * max_pending_corrections;
        if (cap > ekf::max_pending()) cap = ekf::max_pending();
        const size_t cnt = (size_t)pv.B * cap * pv.ld;
        HIPC(hipMalloc((void**)&Uf, cnt * sizeof(double)));
        HIPC(hipMalloc((void**)&Vf, cnt * sizeof(double)));
        HIPC(hipMalloc((void**)&state_alt, (size_t)pv.B * pv.ld * sizeof(double)));
        HIPC(hipMemsetAsync(state_alt, 0, (size_t)pv.B * pv.ld * sizeof(double), stream));
        pend_cap = cap;
        return EKF_OK;
    }

    // fold every pending correction into Sigma_base (no-op in eager mode)
    ekf_status flush() {
        if (pend_count > 0) {
            alt_synced = false;
            ekf::launch_flush(pv, pending(), tuning, stream);
            HIPC(hipGetLastError());
            pend_count = 0;
        }
        return EKF_OK;
    }

    // one landmark correction, eager (gain + covariance stream) or delayed (gain only, factors appended).
    // active_N > 0: the correction is exactly confined to the leading active_N block (data_association()).
    ekf_status correct(const ekf::CmdSrc& src, int active_N = 0) {
        if (pend_cap > 0 && src.mode != ekf::SRC_ASSOC) {
            if (pend_count + 2 > pend_cap) EKFC(flush());
            ekf::launch_gain_delayed(pv, src, pending(), state_alt, stream);
            std::swap(pv.state, state_alt);
            pend_count += 2;
            return EKF_OK;
        }
        EKFC(flush());
        ekf::PoolView view = pv;
        if (active_N > 0 && active_N < pv.N) view.N = active_N;
        if (fused_ok()) {  // single filter: gain + state + covariance in one launch, out of place
            EKFC(ensure_alt());
            ekf::launch_correct_fused(view, src, sigma_alt, state_fz, stream);
            std::swap(pv.sigma, sigma_alt);
            std::swap(pv.state, state_fz);
            return EKF_OK;
        }
        alt_synced = false;
        ekf::launch_gain(view, src, stream);
        if (active_set && active_N == 0) ekf::launch_rank2_active(pv, tuning, touched_bound, stream);
        else ekf::launch_rank2(view, tuning, stream);
        return EKF_OK;
    }

    // single filter: landmark lm is about to be corrected
    void note_touched(int lm) {
        if (host_touched.size() != (size_t)pv.n) host_touched.assign(pv.n, 0);
        if (lm >= 0 && lm < pv.n && !host_touched[lm]) { host_touched[lm] = 1; touched_bound++; }
        if (touched_bound > pv.n) touched_bound = pv.n;
    }

    // batch: bound of the touched-set size after every step of a known-association log
    void compute_log_touch_bound(const int* lm_idx, int T, int vmax) {
        const int B = pv.B, n = pv.n;
        std::vector<unsigned char> seen((size_t)B * (n > 0 ? n : 1), 0);
        std::vector<int> cnt(B, 0);
        log_touch_bound.assign(T, 0);
        touch_bound_base = touched_bound;  // |old set UNION new landmarks| <= old bound + new count
        int best = 0;
        for (int t = 0; t < T; t++) {
            for (int b = 0; b < B; b++)
                for (int v = 0; v < vmax; v++) {
                    const int lm = lm_idx[((size_t)t * B + b) * vmax + v];
                    if (lm < 0 || lm >= n) continue;
                    unsigned char& sflag = seen[(size_t)b * n + lm];
                    if (!sflag) { sflag = 1; if (++cnt[b] > best) best = cnt[b]; }
                }
            log_touch_bound[t] = best;
        }
    }

    ekf_status use() {
        HIPC(hipSetDevice(device));
        return EKF_OK;
    }

    template <class Tp>
    ekf_status dalloc(Tp** p, size_t count) {
        HIPC(hipMalloc((void**)p, count * sizeof(Tp)));
        dev_bytes += count * sizeof(Tp);
        return EKF_OK;
    }

    ekf_status create(int B, int n, const ekf_params* params, int dev) {
        if (B <= 0 || n < 0) return fail(EKF_ERR_INVALID, "B must be > 0 and n >= 0");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
        if (dev < 0) HIPC(hipGetDevice(&dev));
        if (dev >= count) return fail(EKF_ERR_INVALID, "device index out of range");
        hipDeviceProp_t prop;
        HIPC(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(EKF_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
        device = dev;
        EKFC(use());
        HIPC(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        ekf_params p;
        ekf_default_params(&p);
        if (params) p = *params;
        pv.p = ekf::Params{p.sigma0_landmark, p.q_pose, p.r_meas, p.gate_new, p.gate_update, p.straight_eps};
        pv.n = n;
        pv.N = 3 + 2 * n;
        pv.ld = round_up(pv.N, 16);
        pv.B = B;
        pv.sigma_stride = (size_t)pv.N * pv.ld;
        EKFC(dalloc(&pv.sigma, (size_t)B * pv.sigma_stride));
        EKFC(dalloc(&pv.state, (size_t)B * pv.ld));
        EKFC(dalloc(&pv.Kg, (size_t)B * 2 * pv.ld));
        EKFC(dalloc(&pv.Gh, (size_t)B * 2 * pv.ld));
        EKFC(dalloc(&pv.snap, (size_t)B * 4));
        EKFC(dalloc(&pv.rec, (size_t)B));
        EKFC(dalloc(&pv.assoc, (size_t)B));
        EKFC(dalloc(&pv.touch_flag, (size_t)B * (n > 0 ? n : 1)));
        EKFC(dalloc(&pv.touch_list, (size_t)B * (n > 0 ? n : 1)));
        EKFC(dalloc(&pv.touch_count, (size_t)B));
        EKFC(dalloc(&scores, (size_t)B * (n > 0 ? n : 1)));
        EKFC(dalloc(&digest_dev, (size_t)B * 4));
        EKFC(dalloc(&poses_dev, (size_t)B * 3));
        HIPC(hipEventCreate(&ev_begin));
        HIPC(hipEventCreate(&ev_end));
        return reset();
    }

    ekf_status reset() {
        EKFC(use());
        pend_count = 0;  // pending factors of the old run are dropped with it
        touched_hwm = 0;
        touched_bound = 0;
        touch_bound_base = 0;
        std::fill(host_touched.begin(), host_touched.end(), 0);
        ekf::launch_init(pv, stream);
        HIPC(hipGetLastError());
        init_flag = 0;
        alt_synced = false;
        return EKF_OK;
    }

    void destroy() {
        if (device >= 0) (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        void* ptrs[] = {pv.sigma, pv.state, pv.Kg, pv.Gh, pv.snap, pv.rec, pv.assoc, pv.touch_flag, pv.touch_list,
                        pv.touch_count, scores, meas_dev,
                        assoc_out_dev, sensor_dev, digest_dev, poses_dev, log_twist, log_lm, log_z, log_init,
                        Uf, Vf, state_alt, sigma_alt, state_fz, log_truth, ulog_twist, ulog_count, ulog_meas, ulog_assoc, ulog_truth, corr_counter};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        stage_in.release();
        stage_out.release();
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }

    ekf_status sync() {
        EKFC(use());
        HIPC(hipStreamSynchronize(stream));
        return EKF_OK;
    }

    // host -> device through the pinned staging buffer, ordered on the stream
    ekf_status upload(void* dst, const void* src, size_t bytes) { return upload2(dst, src, bytes, nullptr, 0); }

    // one H2D copy of two host pieces laid out back to back (piece 2 lands at dst + bytes1)
    ekf_status upload2(void* dst, const void* src1, size_t bytes1, const void* src2, size_t bytes2) {
        if (bytes1 + bytes2 == 0) return EKF_OK;
        Staging& sg = stage_in.acquire();
        EKFC(sg.reserve(bytes1 + bytes2));
        EKFC(sg.wait());
        std::memcpy(sg.host, src1, bytes1);
        if (bytes2) std::memcpy(static_cast<char*>(sg.host) + bytes1, src2, bytes2);
        HIPC(hipMemcpyAsync(dst, sg.host, bytes1 + bytes2, hipMemcpyHostToDevice, stream));
        return sg.mark(stream);
    }

    // device -> host, blocking
    ekf_status download(void* dst, const void* src, size_t bytes) {
        if (bytes == 0) return EKF_OK;
        EKFC(stage_out.reserve(bytes));
        HIPC(hipMemcpyAsync(stage_out.host, src, bytes, hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        std::memcpy(dst, stage_out.host, bytes);
        return EKF_OK;
    }

    ekf_status get_state(int b, double* out) {
        if (!out || b < 0 || b >= pv.B) return fail(EKF_ERR_INVALID, "get_state: bad argument");
        EKFC(use());
        return download(out, pv.state + (size_t)b * pv.ld, sizeof(double) * pv.N);
    }

    ekf_status set_state(int b, const double* in) {
        if (!in || b < 0 || b >= pv.B) return fail(EKF_ERR_INVALID, "set_state: bad argument");
        EKFC(use());
        return upload(pv.state + (size_t)b * pv.ld, in, sizeof(double) * pv.N);
    }

    ekf_status get_cov(int b, double* out) {
        if (!out || b < 0 || b >= pv.B) return fail(EKF_ERR_INVALID, "get_cov: bad argument");
        EKFC(use());
        EKFC(flush());
        const size_t w = sizeof(double) * pv.N;
        EKFC(stage_out.reserve(w * pv.N));
        HIPC(hipMemcpy2DAsync(stage_out.host, w, pv.sigma + (size_t)b * pv.sigma_stride, sizeof(double) * pv.ld, w,
                              pv.N, hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        std::memcpy(out, stage_out.host, w * pv.N);
        return EKF_OK;
    }

    ekf_status set_cov(int b, const double* in) {
        if (!in || b < 0 || b >= pv.B) return fail(EKF_ERR_INVALID, "set_cov: bad argument");
        EKFC(use());
        EKFC(flush());
        alt_synced = false;
        touched_hwm = pv.n;  // caller-supplied covariance: no structure may be assumed any more
        touched_bound = pv.n;
        std::fill(host_touched.begin(), host_touched.end(), 1);
        ekf::launch_touch_all(pv, stream);
        const size_t w = sizeof(double) * pv.N;
        Staging& sg = stage_in.acquire();
        EKFC(sg.reserve(w * pv.N));
        EKFC(sg.wait());
        std::memcpy(sg.host, in, w * pv.N);
        HIPC(hipMemcpy2DAsync(pv.sigma + (size_t)b * pv.sigma_stride, sizeof(double) * pv.ld, sg.host, w, w,
                              pv.N, hipMemcpyHostToDevice, stream));
        return sg.mark(stream);
    }

    ekf_status ensure_meas_capacity(int J) {
        if (J <= jcap) return EKF_OK;
        HIPC(hipStreamSynchronize(stream));
        if (meas_dev) HIPC(hipFree(meas_dev));
        if (assoc_out_dev) HIPC(hipFree(assoc_out_dev));
        meas_dev = nullptr; assoc_out_dev = nullptr;
        const int cap = J < 64 ? 64 : round_up(J, 64);
        EKFC(dalloc(&meas_dev, (size_t)cap * 2));
        EKFC(dalloc(&assoc_out_dev, (size_t)cap));
        jcap = cap;
        return EKF_OK;
    }

    hipEvent_t* events(size_t need) {
        while (ev_pool.size() < need) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool.data();
    }
};

ekf_status checked_launch() {
    HIPC(hipGetLastError());
    return EKF_OK;
}

}  // namespace

struct ekf_filter_s { Pool pool; };
struct ekf_batch_s { Pool pool; };

struct ekf_dense_s {
    int device = -1, N = 0, ld = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;
    float *F = nullptr, *S = nullptr, *T = nullptr, *Q = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, j1 = nullptr, j2 = nullptr;
};

extern "C" {

const char* ekf_last_error(void) { return g_err.c_str(); }

void ekf_default_params(ekf_params* out) {
    if (!out) return;
    out->sigma0_landmark = 100.0;  // ekf_slam.cpp:32
    out->q_pose = 0.0001;          // ekf_slam.cpp:41-43
    out->r_meas = 0.01;            // ekf_slam.cpp:174-175
    out->gate_new = 10.0;          // ekf_slam.cpp:293
    out->gate_update = 1.0;        // ekf_slam.cpp:330
    out->straight_eps = 0.000001;  // ekf_slam.cpp:79
}

int ekf_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

// ---- single filter -------------------------------------------------------------------------

ekf_status ekf_create(int n, const ekf_params* params, int device, ekf_handle* out) {
    if (!out) return fail(EKF_ERR_INVALID, "ekf_create: out is null");
    *out = nullptr;
    ekf_filter_s* f = new (std::nothrow) ekf_filter_s();
    if (!f) return fail(EKF_ERR_NOMEM, "host allocation failed");
    ekf_status st = f->pool.create(1, n, params, device);
    // sensor_reading (2n doubles) and visible_list (n bytes) share one buffer so one copy brings both
    if (st == EKF_OK) st = f->pool.dalloc(&f->pool.sensor_dev, (size_t)(n > 0 ? 2 * n : 2) + (size_t)(n + 15) / 8);
    if (st == EKF_OK) f->pool.visible_dev = reinterpret_cast<unsigned char*>(f->pool.sensor_dev + (n > 0 ? 2 * n : 2));
    if (st == EKF_OK && ekf::small_prepare() != hipSuccess) st = fail(EKF_ERR_HIP, "hipFuncSetAttribute failed");
    if (st != EKF_OK) {
        f->pool.destroy();
        delete f;
        return st;
    }
    *out = f;
    return EKF_OK;
}

ekf_status ekf_destroy(ekf_handle h) {
    if (!h) return EKF_OK;
    h->pool.destroy();
    delete h;
    return EKF_OK;
}

ekf_status ekf_clone(ekf_handle h, ekf_handle* out) {
    if (!h || !out) return fail(EKF_ERR_INVALID, "ekf_clone: null argument");
    Pool& a = h->pool;
    ekf_params p{a.pv.p.sigma0_landmark, a.pv.p.q_pose, a.pv.p.r_meas, a.pv.p.gate_new, a.pv.p.gate_update,
                 a.pv.p.straight_eps};
    EKFC(ekf_create(a.pv.n, &p, a.device, out));
    Pool& c = (*out)->pool;
    EKFC(a.flush());
    EKFC(a.sync());
    if (a.pend_cap > 0) EKFC(c.set_update_mode(a.pend_cap / 2, a.pend_symmetric));
    HIPC(hipMemcpyAsync(c.pv.sigma, a.pv.sigma, sizeof(double) * a.pv.sigma_stride, hipMemcpyDeviceToDevice, c.stream));
    HIPC(hipMemcpyAsync(c.pv.state, a.pv.state, sizeof(double) * a.pv.ld, hipMemcpyDeviceToDevice, c.stream));
    c.init_flag = a.init_flag;
    c.tuning = a.tuning;
    c.touched_hwm = a.touched_hwm;
    c.small_path = a.small_path;
    c.active_set = a.active_set;
    c.pv.active_set = a.active_set;
    c.touched_bound = a.touched_bound;
    c.host_touched = a.host_touched;
    HIPC(hipMemcpyAsync(c.pv.touch_flag, a.pv.touch_flag, (size_t)(a.pv.n > 0 ? a.pv.n : 1), hipMemcpyDeviceToDevice, c.stream));
    HIPC(hipMemcpyAsync(c.pv.touch_list, a.pv.touch_list, sizeof(int) * (size_t)(a.pv.n > 0 ? a.pv.n : 1), hipMemcpyDeviceToDevice, c.stream));
    HIPC(hipMemcpyAsync(c.pv.touch_count, a.pv.touch_count, sizeof(int), hipMemcpyDeviceToDevice, c.stream));
    c.active_prefix = a.active_prefix;
    c.fused = a.fused;
    return c.sync();
}

ekf_status ekf_predict(ekf_handle h, double dtheta, double dx) {
    if (!h) return fail(EKF_ERR_INVALID, "ekf_predict: null handle");
    Pool& P = h->pool;
    EKFC(P.use());
    // Rows/columns of landmarks this object never corrected are exactly zero against the pose block
    // (constructor values), and At*0*At^T + 0 = 0: the propagation is confined to the touched prefix.
    ekf::PoolView view = P.pv;
    if (P.active_prefix && P.pend_cap == 0 && P.touched_hwm < P.pv.n) view.N = 3 + 2 * P.touched_hwm;
    ekf::launch_predict(view, nullptr, dtheta, dx, P.pending(), P.stream);
    return checked_launch();
}

ekf_status ekf_measure_known(ekf_handle h, const double* sensor_xy, const uint8_t* visible) {
    if (!h || !sensor_xy || !visible) return fail(EKF_ERR_INVALID, "ekf_measure_known: null argument");
    Pool& P = h->pool;
    EKFC(P.use());
    const int n = P.pv.n;
    EKFC(P.upload2(P.sensor_dev, sensor_xy, sizeof(double) * 2 * n, visible, (size_t)n));
    if (P.small_path && P.pend_cap == 0 && P.pv.N <= ekf::small_max_dim() && n > 0) {
        // small map (the reference runs n = 20): the whole call in one LDS-resident launch
        P.alt_synced = false;
        ekf::launch_small_measure(P.pv, P.sensor_dev, P.visible_dev, !P.init_flag, P.stream);
        P.init_flag = 1;
        for (int i = n - 1; i >= 0; i--)
            if (visible[i]) { if (i + 1 > P.touched_hwm) P.touched_hwm = i + 1; break; }
        for (int i = 0; i < n; i++)
            if (visible[i]) P.note_touched(i);
        return checked_launch();
    }
    // ekf_slam.cpp:109-128.  The pose capture needs its own launch only together with the first-call
    // landmark initialisation; afterwards the first correction of the call records the pose it reads
    // (no correction has moved it yet) and the later ones use that record.
    const bool need_begin = !P.init_flag;
    if (need_begin) ekf::launch_measure_begin(P.pv, P.sensor_dev, 1, P.stream);
    P.init_flag = 1;
    ekf::CmdSrc src{};
    src.mode = ekf::SRC_SENSOR_VECTOR;
    src.sensor = P.sensor_dev;
    bool first = !need_begin;
    for (int i = 0; i < n; i++) {  // ekf_slam.cpp:132-194, ascending landmark order
        if (!visible[i]) continue;
        if (i + 1 > P.touched_hwm) P.touched_hwm = i + 1;
        P.note_touched(i);
        src.lm_imm = i;
        src.fresh_pose = first ? 1 : 0;
        src.write_snap = first ? 1 : 0;
        first = false;
        EKFC(P.correct(src));
    }
    return checked_launch();
}

ekf_status ekf_associate(ekf_handle h, const double* meas_xy, int J, uint8_t* known, int* assoc_out) {
    if (!h || !known || J < 0 || (J > 0 && !meas_xy)) return fail(EKF_ERR_INVALID, "ekf_associate: bad argument");
    Pool& P = h->pool;
    EKFC(P.use());
    const int n = P.pv.n;
    int known_count = 0;  // ekf_slam.cpp:281-288: leading run of true
    for (int i = 0; i < n; i++) {
        if (known[i]) known_count++; else break;
    }
    if (J == 0) return EKF_OK;
    EKFC(P.flush());  // the scoring kernel reads Sigma directly
    EKFC(P.ensure_meas_capacity(J));
    EKFC(P.upload(P.meas_dev, meas_xy, sizeof(double) * 2 * J));
    if (P.small_path && P.pv.N <= ekf::small_max_dim() && n > 0 && n <= 128) {
        // small map: scores, decisions and corrections of all J measurements in one LDS-resident launch
        P.alt_synced = false;
        ekf::launch_small_associate(P.pv, P.meas_dev, J, known_count, P.assoc_out_dev, P.stream);
        EKFC(checked_launch());
        ekf::AssocRec rec_s;
        EKFC(P.download(&rec_s, P.pv.assoc, sizeof(rec_s)));
        for (int i = known_count; i < rec_s.known_count && i < n; i++) known[i] = 1;  // :323
        if (rec_s.known_count > P.touched_hwm) P.touched_hwm = rec_s.known_count < n ? rec_s.known_count : n;
        for (int i = 0; i < rec_s.known_count && i < n; i++) P.note_touched(i);
        if (assoc_out) EKFC(P.download(assoc_out, P.assoc_out_dev, sizeof(int) * J));
        return EKF_OK;
    }
    ekf::launch_assoc_begin(P.pv, nullptr, known_count, P.stream);
    ekf::CmdSrc src{};
    src.mode = ekf::SRC_ASSOC;
    src.assoc = P.pv.assoc;
    src.fresh_pose = 1;
    // Landmarks are appended in discovery order (:318-327), so rows/columns beyond 3 + 2*(known_count + j + 1)
    // -- and beyond every landmark this object has ever corrected (touched_hwm: known_list is caller-owned
    // and may have holes) -- still hold their constructor values, and every correction of this call is
    // exactly confined to that leading block: K and H*Sigma are exact zeros outside it.  (Non-finite
    // states void this; they are already garbage in the reference.)
    for (int j = 0; j < J; j++) {  // ekf_slam.cpp:291: sequential, state-carrying
        const double* mj = P.meas_dev + 2 * (size_t)j;
        int active_N = 0;
        if (P.active_prefix) {
            int m = known_count + j + 1 < n ? known_count + j + 1 : n;
            if (P.touched_hwm > m) m = P.touched_hwm;
            active_N = 3 + 2 * m;
        }
        const ekf::MeasSrc ms{mj, 2, nullptr, 0};
        ekf::launch_maha(P.pv, ms, P.scores, -1, known_count + j < n ? known_count + j : n, P.stream);  // :300-309
        ekf::launch_assoc_decide(P.pv, ms, P.scores, P.assoc_out_dev, 0, j, nullptr, P.stream);    // :293-330
        src.meas = mj;
        src.meas_stride = 2;
        EKFC(P.correct(src, active_N));                                                   // :331-390
    }
    EKFC(checked_launch());
    ekf::AssocRec rec;
    EKFC(P.download(&rec, P.pv.assoc, sizeof(rec)));
    for (int i = known_count; i < rec.known_count && i < n; i++) known[i] = 1;  // :323
    if (rec.known_count > P.touched_hwm) P.touched_hwm = rec.known_count < n ? rec.known_count : n;
    // for the host-side bound every landmark below the new known_count counts as touched (a superset
    // of what this call's decisions actually corrected)
    for (int i = 0; i < rec.known_count && i < n; i++) P.note_touched(i);
    if (assoc_out) EKFC(P.download(assoc_out, P.assoc_out_dev, sizeof(int) * J));
    return EKF_OK;
}

ekf_status ekf_maha_scores(ekf_handle h, double meas_x, double meas_y, int M, double* scores_out) {
    if (!h || !scores_out || M < 0) return fail(EKF_ERR_INVALID, "ekf_maha_scores: bad argument");
    Pool& P = h->pool;
    if (M > P.pv.n) return fail(EKF_ERR_INVALID, "ekf_maha_scores: M exceeds the number of landmarks");
    if (M == 0) return EKF_OK;
    EKFC(P.use());
    EKFC(P.flush());
    EKFC(P.ensure_meas_capacity(1));
    const double m[2] = {meas_x, meas_y};
    EKFC(P.upload(P.meas_dev, m, sizeof(m)));
    ekf::launch_maha(P.pv, ekf::MeasSrc{P.meas_dev, 2, nullptr, 0}, P.scores, M, -1, P.stream);
    EKFC(checked_launch());
    return P.download(scores_out, P.scores, sizeof(double) * M);
}

ekf_status ekf_get_pose(ekf_handle h, double out[3]) {
    if (!h || !out) return fail(EKF_ERR_INVALID, "ekf_get_pose: null argument");
    EKFC(h->pool.use());
    return h->pool.download(out, h->pool.pv.state, sizeof(double) * 3);
}

ekf_status ekf_get_landmarks(ekf_handle h, double* out) {
    if (!h || !out) return fail(EKF_ERR_INVALID, "ekf_get_landmarks: null argument");
    EKFC(h->pool.use());
    return h->pool.download(out, h->pool.pv.state + 3, sizeof(double) * 2 * h->pool.pv.n);
}

ekf_status ekf_dim(ekf_handle h, int* n, int* N) {
    if (!h) return fail(EKF_ERR_INVALID, "ekf_dim: null handle");
    if (n) *n = h->pool.pv.n;
    if (N) *N = h->pool.pv.N;
    return EKF_OK;
}

ekf_status ekf_get_state(ekf_handle h, double* out) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.get_state(0, out);
}
ekf_status ekf_set_state(ekf_handle h, const double* in) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.set_state(0, in);
}
ekf_status ekf_get_cov(ekf_handle h, double* out) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.get_cov(0, out);
}
ekf_status ekf_set_cov(ekf_handle h, const double* in) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.set_cov(0, in);
}
ekf_status ekf_get_init_flag(ekf_handle h, int* flag) {
    if (!h || !flag) return fail(EKF_ERR_INVALID, "null argument");
    *flag = h->pool.init_flag;
    return EKF_OK;
}
ekf_status ekf_set_init_flag(ekf_handle h, int flag) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.init_flag = flag ? 1 : 0;
    return EKF_OK;
}
ekf_status ekf_set_active_set(ekf_handle h, int enable) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.active_set = enable ? 1 : 0;
    h->pool.pv.active_set = h->pool.active_set;
    return EKF_OK;
}
ekf_status ekf_batch_set_active_set(ekf_batch_handle hb, int enable) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    hb->pool.active_set = enable ? 1 : 0;
    hb->pool.pv.active_set = hb->pool.active_set;
    return EKF_OK;
}
ekf_status ekf_batch_get_touched(ekf_batch_handle hb, int* counts_out) {
    if (!hb || !counts_out) return fail(EKF_ERR_INVALID, "null argument");
    EKFC(hb->pool.use());
    return hb->pool.download(counts_out, hb->pool.pv.touch_count, sizeof(int) * hb->pool.pv.B);
}
ekf_status ekf_set_small_map_path(ekf_handle h, int enable) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.small_path = enable ? 1 : 0;
    return EKF_OK;
}
ekf_status ekf_set_active_prefix(ekf_handle h, int enable) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.active_prefix = enable ? 1 : 0;
    return EKF_OK;
}
ekf_status ekf_set_fused_correction(ekf_handle h, int enable) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.fused = enable ? 1 : 0;
    h->pool.alt_synced = false;
    return EKF_OK;
}
ekf_status ekf_batch_set_small_map_path(ekf_batch_handle hb, int enable) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    hb->pool.small_path = enable ? 1 : 0;
    return EKF_OK;
}
ekf_status ekf_batch_set_active_prefix(ekf_batch_handle hb, int enable) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    hb->pool.active_prefix = enable ? 1 : 0;
    return EKF_OK;
}
ekf_status ekf_sync(ekf_handle h) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.sync();
}
ekf_status ekf_set_tuning(ekf_handle h, int rows_per_block, int nontemporal, int group_rows) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.tuning = ekf::Rank2Tuning{rows_per_block, nontemporal, group_rows};
    return EKF_OK;
}

// ---- batch ---------------------------------------------------------------------------------

ekf_status ekf_batch_create(int B, int n, const ekf_params* params, int device, ekf_batch_handle* out) {
    if (!out) return fail(EKF_ERR_INVALID, "ekf_batch_create: out is null");
    *out = nullptr;
    if (B > 65535) return fail(EKF_ERR_INVALID, "B must be <= 65535 (grid dimension)");
    ekf_batch_s* f = new (std::nothrow) ekf_batch_s();
    if (!f) return fail(EKF_ERR_NOMEM, "host allocation failed");
    ekf_status st = f->pool.create(B, n, params, device);
    if (st == EKF_OK && ekf::small_prepare() != hipSuccess) st = fail(EKF_ERR_HIP, "hipFuncSetAttribute failed");
    if (st != EKF_OK) {
        f->pool.destroy();
        delete f;
        return st;
    }
    *out = f;
    return EKF_OK;
}

ekf_status ekf_batch_destroy(ekf_batch_handle hb) {
    if (!hb) return EKF_OK;
    hb->pool.destroy();
    delete hb;
    return EKF_OK;
}

ekf_status ekf_batch_reset(ekf_batch_handle hb) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    return hb->pool.reset();
}

ekf_status ekf_batch_device_bytes(ekf_batch_handle hb, size_t* bytes) {
    if (!hb || !bytes) return fail(EKF_ERR_INVALID, "null argument");
    *bytes = hb->pool.dev_bytes + hb->pool.log_bytes + hb->pool.ulog_bytes;
    return EKF_OK;
}

ekf_status ekf_batch_set_tuning(ekf_batch_handle hb, int rows_per_block, int nontemporal, int group_rows) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    hb->pool.tuning = ekf::Rank2Tuning{rows_per_block, nontemporal, group_rows};
    return EKF_OK;
}

void ekf_default_sim_params(ekf_sim_params* out) {
    if (!out) return;
    out->seed = 5000000ull;
    out->first_filter_id = 0;
    out->v_cmd = 0.5; out->w_cmd = 0.06;
    out->vx_std = 0.01; out->the_std = 0.01;      // noise_param.yaml:3,5
    out->slip_min = 0.90; out->slip_max = 1.10;   // noise_param.yaml:6-7
    out->sensor_std = 0.005;                      // noise_param.yaml:8-9
    out->max_visible_dis = 0.7;                   // noise_param.yaml:10
    out->wheel_base = 0.16; out->wheel_radius = 0.033;  // fake_turtle_param.yaml:6-7
    out->ticks_per_step = 10;
}

static ekf_status free_log(Pool& P) {
    HIPC(hipStreamSynchronize(P.stream));
    for (void* p : {(void*)P.log_twist, (void*)P.log_lm, (void*)P.log_z, (void*)P.log_init, (void*)P.log_truth})
        if (p) HIPC(hipFree(p));
    P.log_twist = nullptr; P.log_lm = nullptr; P.log_z = nullptr; P.log_init = nullptr; P.log_truth = nullptr;
    P.T = 0; P.vmax = 0; P.log_bytes = 0;
    return EKF_OK;
}

ekf_status ekf_batch_simulate_known_log(ekf_batch_handle hb, const ekf_sim_params* sp, const double* world_xy, int T,
                                        int vmax) {
    if (!hb || !sp || !world_xy || T <= 0 || vmax < 0 || vmax > 64 || sp->ticks_per_step < 1)
        return fail(EKF_ERR_INVALID, "ekf_batch_simulate_known_log: bad argument (vmax <= 64)");
    Pool& P = hb->pool;
    EKFC(P.use());
    EKFC(free_log(P));
    const int B = P.pv.B, n = P.pv.n;
    const size_t n_tw = (size_t)T * B * 2, n_lm = (size_t)T * B * vmax, n_z = n_lm * 2, n_in = (size_t)B * 2 * n,
                 n_tr = (size_t)T * B * 3;
    HIPC(hipMalloc((void**)&P.log_twist, sizeof(double) * n_tw));
    HIPC(hipMalloc((void**)&P.log_lm, sizeof(int) * (n_lm ? n_lm : 1)));
    HIPC(hipMalloc((void**)&P.log_z, sizeof(double) * (n_z ? n_z : 1)));
    HIPC(hipMalloc((void**)&P.log_init, sizeof(double) * (n_in ? n_in : 1)));
    HIPC(hipMalloc((void**)&P.log_truth, sizeof(double) * n_tr));
    P.log_bytes = sizeof(double) * (n_tw + n_z + n_in + n_tr) + sizeof(int) * n_lm;
    double* d_world = nullptr;
    int* d_active = nullptr;
    const size_t n_act = (size_t)T * (vmax > 0 ? vmax : 1);
    HIPC(hipMalloc((void**)&d_world, sizeof(double) * 2 * (n > 0 ? n : 1)));
    HIPC(hipMalloc((void**)&d_active, sizeof(int) * n_act));
    ekf_status st = EKF_OK;
    auto body = [&]() -> ekf_status {
        if (n > 0) HIPC(hipMemcpyAsync(d_world, world_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, P.stream));
        HIPC(hipMemsetAsync(d_active, 0, sizeof(int) * n_act, P.stream));
        ekf::SimParams p{sp->seed, sp->first_filter_id, sp->v_cmd, sp->w_cmd, sp->vx_std, sp->the_std, sp->slip_min,
                         sp->slip_max, sp->sensor_std, sp->max_visible_dis, sp->wheel_base, sp->wheel_radius,
                         sp->ticks_per_step};
        ekf::launch_sim(p, B, n, T, vmax, d_world, P.log_twist, P.log_truth, P.log_lm, P.log_z, P.log_init, d_active,
                        P.stream);
        HIPC(hipGetLastError());
        std::vector<int> active(n_act, 0);
        HIPC(hipMemcpyAsync(active.data(), d_active, sizeof(int) * n_act, hipMemcpyDeviceToHost, P.stream));
        HIPC(hipStreamSynchronize(P.stream));
        P.slot_active.swap(active);
        std::vector<int> lm_host(n_lm ? n_lm : 1, -1);
        if (n_lm) HIPC(hipMemcpy(lm_host.data(), P.log_lm, sizeof(int) * n_lm, hipMemcpyDeviceToHost));
        P.compute_log_touch_bound(lm_host.data(), T, vmax);
        return EKF_OK;
    };
    st = body();
    (void)hipFree(d_world);
    (void)hipFree(d_active);
    if (st != EKF_OK) return st;
    P.T = T;
    P.truth_is_unknown_log = 0;
    P.vmax = vmax;
    return EKF_OK;
}

ekf_status ekf_batch_download_log(ekf_batch_handle hb, double* twist, int* lm_idx, double* z_xy, double* init_xy,
                                  double* true_pose) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.T <= 0) return fail(EKF_ERR_STATE, "no log on the device");
    if (true_pose && !P.log_truth) return fail(EKF_ERR_STATE, "the uploaded log carries no simulated truth");
    EKFC(P.use());
    HIPC(hipStreamSynchronize(P.stream));
    const size_t B = P.pv.B, T = P.T, vmax = P.vmax, n = P.pv.n;
    if (twist) HIPC(hipMemcpy(twist, P.log_twist, sizeof(double) * T * B * 2, hipMemcpyDeviceToHost));
    if (lm_idx && vmax) HIPC(hipMemcpy(lm_idx, P.log_lm, sizeof(int) * T * B * vmax, hipMemcpyDeviceToHost));
    if (z_xy && vmax) HIPC(hipMemcpy(z_xy, P.log_z, sizeof(double) * T * B * vmax * 2, hipMemcpyDeviceToHost));
    if (init_xy && n) HIPC(hipMemcpy(init_xy, P.log_init, sizeof(double) * B * 2 * n, hipMemcpyDeviceToHost));
    if (true_pose) HIPC(hipMemcpy(true_pose, P.log_truth, sizeof(double) * T * B * 3, hipMemcpyDeviceToHost));
    return EKF_OK;
}

ekf_status ekf_batch_mc_stats(ekf_batch_handle hb, int t, double out[6]) {
    if (!hb || !out) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    const double* truth = P.truth_is_unknown_log ? P.ulog_truth : P.log_truth;
    const int Tl = P.truth_is_unknown_log ? P.uT : P.T;
    if (!truth) return fail(EKF_ERR_STATE, "ekf_batch_mc_stats needs a simulated log (ground truth)");
    if (t < 0 || t >= Tl) return fail(EKF_ERR_INVALID, "step outside the log");
    EKFC(P.use());
    EKFC(P.flush());
    ekf::launch_mc_stats(P.pv, truth + (size_t)t * P.pv.B * 3, P.digest_dev, P.stream);
    EKFC(checked_launch());
    std::vector<double> h((size_t)4 * P.pv.B);
    EKFC(P.download(h.data(), P.digest_dev, sizeof(double) * h.size()));
    double nees = 0, nmax = 0, p2 = 0, a2 = 0, tr = 0, inside = 0;
    for (int b = 0; b < P.pv.B; b++) {
        const double v = h[(size_t)b * 4];
        nees += v; if (v > nmax) nmax = v;
        p2 += h[(size_t)b * 4 + 1]; a2 += h[(size_t)b * 4 + 2]; tr += h[(size_t)b * 4 + 3];
        if (v < 7.815) inside += 1.0;
    }
    const double Bn = (double)P.pv.B;
    out[0] = nees / Bn; out[1] = nmax; out[2] = std::sqrt(p2 / Bn); out[3] = std::sqrt(a2 / Bn); out[4] = tr / Bn;
    out[5] = inside / Bn;
    return EKF_OK;
}

ekf_status ekf_batch_set_update_mode(ekf_batch_handle hb, int max_pending_corrections, int symmetric_gather) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    return hb->pool.set_update_mode(max_pending_corrections, symmetric_gather);
}

ekf_status ekf_set_update_mode(ekf_handle h, int max_pending_corrections, int symmetric_gather) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.set_update_mode(max_pending_corrections, symmetric_gather);
}

ekf_status ekf_batch_upload_known_log(ekf_batch_handle hb, const ekf_known_log* log) {
    if (!hb || !log || !log->twist || !log->lm_idx || !log->z_xy || !log->init_xy || log->T <= 0 || log->vmax < 0)
        return fail(EKF_ERR_INVALID, "ekf_batch_upload_known_log: bad argument");
    Pool& P = hb->pool;
    EKFC(P.use());
    const int B = P.pv.B, n = P.pv.n, T = log->T, vmax = log->vmax;
    // validate: indices in range, ascending within a step (the loop order of ekf_slam.cpp:132)
    std::vector<int> active((size_t)T * (vmax > 0 ? vmax : 1), 0);
    for (int t = 0; t < T; t++)
        for (int b = 0; b < B; b++) {
            int prev = -1;
            bool ended = false;
            for (int v = 0; v < vmax; v++) {
                const int lm = log->lm_idx[((size_t)t * B + b) * vmax + v];
                if (lm < 0) { ended = true; continue; }
                if (ended || lm >= n || lm <= prev)
                    return fail(EKF_ERR_INVALID, "known log: landmark indices must be < n, strictly ascending, -1 padded");
                prev = lm;
                active[(size_t)t * vmax + v]++;
            }
        }
    EKFC(free_log(P));
    const size_t n_tw = (size_t)T * B * 2, n_lm = (size_t)T * B * vmax, n_z = n_lm * 2, n_in = (size_t)B * 2 * n;
    HIPC(hipMalloc((void**)&P.log_twist, sizeof(double) * (n_tw ? n_tw : 1)));
    HIPC(hipMalloc((void**)&P.log_lm, sizeof(int) * (n_lm ? n_lm : 1)));
    HIPC(hipMalloc((void**)&P.log_z, sizeof(double) * (n_z ? n_z : 1)));
    HIPC(hipMalloc((void**)&P.log_init, sizeof(double) * (n_in ? n_in : 1)));
    P.log_bytes = sizeof(double) * (n_tw + n_z + n_in) + sizeof(int) * n_lm;
    HIPC(hipMemcpy(P.log_twist, log->twist, sizeof(double) * n_tw, hipMemcpyHostToDevice));
    if (n_lm) HIPC(hipMemcpy(P.log_lm, log->lm_idx, sizeof(int) * n_lm, hipMemcpyHostToDevice));
    if (n_z) HIPC(hipMemcpy(P.log_z, log->z_xy, sizeof(double) * n_z, hipMemcpyHostToDevice));
    if (n_in) HIPC(hipMemcpy(P.log_init, log->init_xy, sizeof(double) * n_in, hipMemcpyHostToDevice));
    P.T = T;
    P.vmax = vmax;
    P.slot_active.swap(active);
    P.compute_log_touch_bound(log->lm_idx, T, vmax);
    return EKF_OK;
}

ekf_status ekf_batch_run_known(ekf_batch_handle hb, int t_begin, int t_end, int time_kernels, ekf_run_stats* stats) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.T <= 0) return fail(EKF_ERR_STATE, "ekf_batch_run_known: no log uploaded");
    if (t_begin < 0 || t_end > P.T || t_begin > t_end) return fail(EKF_ERR_INVALID, "step range outside the uploaded log");
    EKFC(P.use());
    const int B = P.pv.B, vmax = P.vmax;
    P.touched_hwm = P.pv.n;  // a known log corrects arbitrary indices: no discovered-prefix structure afterwards
    P.alt_synced = false;
    size_t launches = 0;
    long long corrections = 0;
    for (int t = t_begin; t < t_end; t++)
        for (int v = 0; v < vmax; v++)
            if (P.slot_active[(size_t)t * vmax + v] > 0) {
                launches++;
                corrections += P.slot_active[(size_t)t * vmax + v];
            }
    const bool delayed = P.pend_cap > 0;
    // worst-case number of covariance passes (rank-2 launches, or flushes in delayed mode) for the events
    const size_t max_passes = delayed ? launches / (size_t)(P.pend_cap / 2) + 2 : launches;
    hipEvent_t* ev = nullptr;
    if (time_kernels && launches) {
        ev = P.events(2 * max_passes);
        if (!ev) return fail(EKF_ERR_HIP, "hipEventCreate failed");
    }
    HIPC(hipEventRecord(P.ev_begin, P.stream));
    size_t k = 0;
    auto timed_flush = [&]() -> ekf_status {
        if (P.pend_count == 0) return EKF_OK;
        if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
        EKFC(P.flush());
        if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
        k++;
        return EKF_OK;
    };
    ekf::CmdSrc src{};
    src.mode = ekf::SRC_COMPACT_LOG;
    src.vmax = vmax;
    src.fresh_pose = 0;
    // Small maps (the reference's own n = 20): the whole step range in ONE launch, every filter's Sigma resident
    // in LDS from the first step to the last (k_pool_run_known); bit-identical to the replay below.
    const bool small_run = P.small_path && !delayed && !P.active_set && P.pv.n > 0 && P.pv.N <= ekf::small_max_dim() &&
                           vmax <= 64 && t_end > t_begin;
    if (small_run) {
        if (ev) HIPC(hipEventRecord(ev[0], P.stream));
        ekf::launch_pool_run_known(P.pv, P.log_twist, P.log_lm, P.log_z, P.log_init, vmax, t_begin, t_end, !P.init_flag,
                                   P.stream);
        if (ev) HIPC(hipEventRecord(ev[1], P.stream));
        k = 1;
        P.init_flag = 1;
        if ((size_t)(t_end - 1) < P.log_touch_bound.size()) {
            int cand = P.touch_bound_base + P.log_touch_bound[t_end - 1];
            if (cand > P.pv.n) cand = P.pv.n;
            if (cand > P.touched_bound) P.touched_bound = cand;
        }
    }
    for (int t = small_run ? t_end : t_begin; t < t_end; t++) {
        ekf::launch_predict(P.pv, P.log_twist + (size_t)t * B * 2, 0.0, 0.0, P.pending(), P.stream);  // prediction()
        ekf::launch_measure_begin(P.pv, P.log_init, !P.init_flag, P.stream);               // measurement() top
        P.init_flag = 1;
        src.lm_idx = P.log_lm + (size_t)t * B * vmax;
        src.z_xy = P.log_z + (size_t)t * B * vmax * 2;
        if ((size_t)t < P.log_touch_bound.size()) {
            int cand = P.touch_bound_base + P.log_touch_bound[t];
            if (cand > P.pv.n) cand = P.pv.n;
            if (cand > P.touched_bound) P.touched_bound = cand;
        }
        for (int v = 0; v < vmax; v++) {
            if (P.slot_active[(size_t)t * vmax + v] == 0) continue;
            src.v = v;
            if (delayed) {
                if (P.pend_count + 2 > P.pend_cap) EKFC(timed_flush());
                EKFC(P.correct(src));
            } else {
                ekf::launch_gain(P.pv, src, P.stream);
                if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
                if (P.active_set) ekf::launch_rank2_active(P.pv, P.tuning, P.touched_bound, P.stream);
                else ekf::launch_rank2(P.pv, P.tuning, P.stream);
                if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
                k++;
            }
        }
    }
    if (delayed) EKFC(timed_flush());  // every run leaves Sigma materialised
    const size_t passes = k;
    HIPC(hipEventRecord(P.ev_end, P.stream));
    EKFC(checked_launch());
    HIPC(hipStreamSynchronize(P.stream));
    if (stats) {
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, P.ev_begin, P.ev_end));
        stats->elapsed_ms = ms;
        stats->rank2_ms = 0.0;
        stats->rank2_launches = (long long)passes;
        if (ev)
            for (size_t i = 0; i < passes; i++) {
                float m = 0.f;
                HIPC(hipEventElapsedTime(&m, ev[2 * i], ev[2 * i + 1]));
                stats->rank2_ms += m;
            }
        stats->corrections = corrections;
        stats->filter_steps = (long long)B * (t_end - t_begin);
        // algorithmic bytes of one covariance pass: every filter's Sigma read + written once
        const double per_pass = 2.0 * sizeof(double) * (double)P.pv.N * (double)P.pv.N;
        stats->rank2_bytes_per_launch =
            delayed ? per_pass * (double)B : (launches ? per_pass * (double)corrections / (double)launches : 0.0);
    }
    return EKF_OK;
}

// ---- batched unknown data association ---------------------------------------------------------

static ekf_status free_ulog(Pool& P) {
    HIPC(hipStreamSynchronize(P.stream));
    for (void* p : {(void*)P.ulog_twist, (void*)P.ulog_count, (void*)P.ulog_meas, (void*)P.ulog_assoc, (void*)P.ulog_truth})
        if (p) HIPC(hipFree(p));
    P.ulog_twist = nullptr; P.ulog_count = nullptr; P.ulog_meas = nullptr; P.ulog_assoc = nullptr; P.ulog_truth = nullptr;
    P.truth_is_unknown_log = 0;
    P.uT = 0; P.ujmax = 0; P.ulog_bytes = 0;
    P.ucount_host.clear();
    return EKF_OK;
}

ekf_status ekf_batch_upload_unknown_log(ekf_batch_handle hb, const ekf_unknown_log* log) {
    if (!hb || !log || !log->twist || !log->count || log->T <= 0 || log->jmax < 0 || (log->jmax > 0 && !log->meas_xy))
        return fail(EKF_ERR_INVALID, "ekf_batch_upload_unknown_log: bad argument");
    Pool& P = hb->pool;
    EKFC(P.use());
    const int B = P.pv.B, T = log->T, jmax = log->jmax;
    for (size_t i = 0; i < (size_t)T * B; i++)
        if (log->count[i] < 0 || log->count[i] > jmax)
            return fail(EKF_ERR_INVALID, "unknown log: count must lie in 0..jmax");
    EKFC(free_ulog(P));
    const size_t n_tw = (size_t)T * B * 2, n_ct = (size_t)T * B, n_me = n_ct * jmax * 2, n_as = n_ct * jmax;
    HIPC(hipMalloc((void**)&P.ulog_twist, sizeof(double) * n_tw));
    HIPC(hipMalloc((void**)&P.ulog_count, sizeof(int) * n_ct));
    HIPC(hipMalloc((void**)&P.ulog_meas, sizeof(double) * (n_me ? n_me : 1)));
    HIPC(hipMalloc((void**)&P.ulog_assoc, sizeof(int) * (n_as ? n_as : 1)));
    if (!P.corr_counter) HIPC(hipMalloc((void**)&P.corr_counter, sizeof(unsigned long long)));
    P.ulog_bytes = sizeof(double) * (n_tw + n_me) + sizeof(int) * (n_ct + n_as);
    HIPC(hipMemcpy(P.ulog_twist, log->twist, sizeof(double) * n_tw, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(P.ulog_count, log->count, sizeof(int) * n_ct, hipMemcpyHostToDevice));
    if (n_me) HIPC(hipMemcpy(P.ulog_meas, log->meas_xy, sizeof(double) * n_me, hipMemcpyHostToDevice));
    if (n_as) {
        std::vector<int> fill(n_as, -2);
        HIPC(hipMemcpy(P.ulog_assoc, fill.data(), sizeof(int) * n_as, hipMemcpyHostToDevice));
    }
    P.ucount_host.assign(log->count, log->count + n_ct);
    P.uT = T;
    P.ujmax = jmax;
    return EKF_OK;
}

void ekf_default_lidar_params(ekf_lidar_params* out) {
    if (!out) return;
    out->n_beams = 360;          // tube_world.cpp:452
    out->range_std = 0.005;      // noise_param.yaml
    out->range_max = 3.5;        // tube_world.cpp:476
    out->border_width = 2.0;     // tube_param.yaml
    out->tube_radius = 0.0762;   // tube_param.yaml
}

static ekf::SimParams to_sim(const ekf_sim_params* sp) {
    return ekf::SimParams{sp->seed, sp->first_filter_id, sp->v_cmd, sp->w_cmd, sp->vx_std, sp->the_std, sp->slip_min,
                          sp->slip_max, sp->sensor_std, sp->max_visible_dis, sp->wheel_base, sp->wheel_radius,
                          sp->ticks_per_step};
}

static bool lidar_ok(const ekf_lidar_params* lp) {
    return lp->n_beams >= 8 && lp->n_beams <= ekf::circles_max_beams() && lp->range_max > 0 && lp->border_width > 0 &&
           lp->tube_radius > 0 && lp->range_std >= 0;
}

ekf_status ekf_batch_simulate_unknown_log(ekf_batch_handle hb, const ekf_sim_params* sp, const ekf_lidar_params* lidar,
                                          const double* world_xy, int T, int jmax) {
    if (!hb || !sp || !world_xy || T <= 0 || jmax < 1 || jmax > 64 || sp->ticks_per_step < 1 || (lidar && !lidar_ok(lidar)))
        return fail(EKF_ERR_INVALID, "ekf_batch_simulate_unknown_log: bad argument (1 <= jmax <= 64, 8 <= n_beams <= 1024)");
    Pool& P = hb->pool;
    EKFC(P.use());
    EKFC(free_ulog(P));
    const int B = P.pv.B, n = P.pv.n;
    const size_t n_tw = (size_t)T * B * 2, n_ct = (size_t)T * B, n_me = n_ct * jmax * 2, n_as = n_ct * jmax, n_tr = n_ct * 3;
    HIPC(hipMalloc((void**)&P.ulog_twist, sizeof(double) * n_tw));
    HIPC(hipMalloc((void**)&P.ulog_count, sizeof(int) * n_ct));
    HIPC(hipMalloc((void**)&P.ulog_meas, sizeof(double) * n_me));
    HIPC(hipMalloc((void**)&P.ulog_assoc, sizeof(int) * n_as));
    HIPC(hipMalloc((void**)&P.ulog_truth, sizeof(double) * n_tr));
    if (!P.corr_counter) HIPC(hipMalloc((void**)&P.corr_counter, sizeof(unsigned long long)));
    P.ulog_bytes = sizeof(double) * (n_tw + n_me + n_tr) + sizeof(int) * (n_ct + n_as);
    double *d_world = nullptr, *d_ranges = nullptr, *d_radii = nullptr;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_world, sizeof(double) * 2 * (n > 0 ? n : 1)));
        if (n > 0) HIPC(hipMemcpyAsync(d_world, world_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, P.stream));
        HIPC(hipMemsetAsync(P.ulog_meas, 0, sizeof(double) * n_me, P.stream));
        HIPC(hipMemsetAsync(P.ulog_assoc, 0xFF, sizeof(int) * n_as, P.stream));  // -1; run_unknown overwrites
        const ekf::SimParams p = to_sim(sp);
        if (!lidar) {
            ekf::launch_sim_unknown(p, B, n, T, jmax, d_world, P.ulog_twist, P.ulog_truth, P.ulog_count, P.ulog_meas,
                                    true, P.stream);
        } else {
            ekf::launch_sim_unknown(p, B, n, T, jmax, d_world, P.ulog_twist, P.ulog_truth, nullptr, nullptr, true, P.stream);
            const ekf::LidarParams lp{lidar->n_beams, lidar->range_std, lidar->range_max, lidar->border_width,
                                      lidar->tube_radius};
            // scans are produced and consumed in chunks of whole steps (<= 256 MiB of ranges at a time)
            size_t steps_per_chunk = ((size_t)256 << 20) / (sizeof(double) * lp.n_beams * B);
            if (steps_per_chunk < 1) steps_per_chunk = 1;
            if (steps_per_chunk > (size_t)T) steps_per_chunk = T;
            HIPC(hipMalloc((void**)&d_ranges, sizeof(double) * steps_per_chunk * B * lp.n_beams));
            HIPC(hipMalloc((void**)&d_radii, sizeof(double) * steps_per_chunk * B * jmax));
            for (int t0 = 0; t0 < T; t0 += (int)steps_per_chunk) {
                const int tc = T - t0 < (int)steps_per_chunk ? T - t0 : (int)steps_per_chunk;
                const int S = tc * B;
                ekf::launch_sim_scans(p, lp, B, n, S, t0, d_world, P.ulog_truth + (size_t)t0 * B * 3, d_ranges, P.stream);
                ekf::launch_circles(d_ranges, S, lp.n_beams, jmax, P.ulog_meas + (size_t)t0 * B * jmax * 2, d_radii,
                                    P.ulog_count + (size_t)t0 * B, nullptr, nullptr, P.stream);
            }
        }
        HIPC(hipGetLastError());
        P.ucount_host.assign(n_ct, 0);
        HIPC(hipMemcpyAsync(P.ucount_host.data(), P.ulog_count, sizeof(int) * n_ct, hipMemcpyDeviceToHost, P.stream));
        HIPC(hipStreamSynchronize(P.stream));
        return EKF_OK;
    };
    const ekf_status st = body();
    for (void* q : {(void*)d_world, (void*)d_ranges, (void*)d_radii})
        if (q) (void)hipFree(q);
    if (st != EKF_OK) return st;
    // decisions start as "not run" (-2)
    {
        std::vector<int> fill(n_as, -2);
        HIPC(hipMemcpy(P.ulog_assoc, fill.data(), sizeof(int) * n_as, hipMemcpyHostToDevice));
    }
    P.uT = T;
    P.ujmax = jmax;
    P.truth_is_unknown_log = 1;
    return EKF_OK;
}

ekf_status ekf_batch_download_unknown_log(ekf_batch_handle hb, double* twist, int* count, double* meas_xy,
                                          double* true_pose) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.uT <= 0) return fail(EKF_ERR_STATE, "no unknown-association log on the device");
    if (true_pose && !P.ulog_truth) return fail(EKF_ERR_STATE, "the uploaded log carries no simulated truth");
    EKFC(P.use());
    HIPC(hipStreamSynchronize(P.stream));
    const size_t B = P.pv.B, T = P.uT, J = P.ujmax;
    if (twist) HIPC(hipMemcpy(twist, P.ulog_twist, sizeof(double) * T * B * 2, hipMemcpyDeviceToHost));
    if (count) HIPC(hipMemcpy(count, P.ulog_count, sizeof(int) * T * B, hipMemcpyDeviceToHost));
    if (meas_xy && J) HIPC(hipMemcpy(meas_xy, P.ulog_meas, sizeof(double) * T * B * J * 2, hipMemcpyDeviceToHost));
    if (true_pose) HIPC(hipMemcpy(true_pose, P.ulog_truth, sizeof(double) * T * B * 3, hipMemcpyDeviceToHost));
    return EKF_OK;
}

ekf_status ekf_simulate_scans(int device, const ekf_sim_params* sp, const ekf_lidar_params* lidar, const double* world_xy,
                              int n, const double* poses, int S, int step, double* ranges_out) {
    if (!sp || !lidar || !lidar_ok(lidar) || n < 0 || (n > 0 && !world_xy) || S < 0 || (S > 0 && (!poses || !ranges_out)) ||
        step < 0)
        return fail(EKF_ERR_INVALID, "ekf_simulate_scans: bad argument (8 <= n_beams <= 1024)");
    if (S == 0) return EKF_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= count) return fail(EKF_ERR_INVALID, "device index out of range");
    HIPC(hipSetDevice(device));
    double *d_world = nullptr, *d_poses = nullptr, *d_ranges = nullptr;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_world, sizeof(double) * 2 * (n > 0 ? n : 1)));
        HIPC(hipMalloc((void**)&d_poses, sizeof(double) * 3 * S));
        HIPC(hipMalloc((void**)&d_ranges, sizeof(double) * (size_t)S * lidar->n_beams));
        if (n > 0) HIPC(hipMemcpy(d_world, world_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(d_poses, poses, sizeof(double) * 3 * S, hipMemcpyHostToDevice));
        const ekf::LidarParams lp{lidar->n_beams, lidar->range_std, lidar->range_max, lidar->border_width,
                                  lidar->tube_radius};
        // B = S, t0 = step: scan s draws the noise stream of filter first_filter_id + s at that step
        ekf::launch_sim_scans(to_sim(sp), lp, S, n, S, step, d_world, d_poses, d_ranges, nullptr);
        HIPC(hipGetLastError());
        HIPC(hipMemcpy(ranges_out, d_ranges, sizeof(double) * (size_t)S * lidar->n_beams, hipMemcpyDeviceToHost));
        return EKF_OK;
    };
    const ekf_status st = body();
    for (void* q : {(void*)d_world, (void*)d_poses, (void*)d_ranges})
        if (q) (void)hipFree(q);
    return st;
}

ekf_status ekf_batch_run_unknown(ekf_batch_handle hb, int t_begin, int t_end, int time_kernels, ekf_run_stats* stats) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.uT <= 0) return fail(EKF_ERR_STATE, "ekf_batch_run_unknown: no unknown-association log uploaded");
    if (t_begin < 0 || t_end > P.uT || t_begin > t_end) return fail(EKF_ERR_INVALID, "step range outside the uploaded log");
    EKFC(P.use());
    EKFC(P.flush());
    P.alt_synced = false;
    const int B = P.pv.B, n = P.pv.n, jmax = P.ujmax;
    size_t launches = 0;
    for (int t = t_begin; t < t_end; t++) {
        const int* ct = P.ucount_host.data() + (size_t)t * B;
        int smax = 0;
        for (int b = 0; b < B; b++) if (ct[b] > smax) smax = ct[b];
        launches += smax;
    }
    hipEvent_t* ev = nullptr;
    if (time_kernels && launches) {
        ev = P.events(2 * launches);
        if (!ev) return fail(EKF_ERR_HIP, "hipEventCreate failed");
    }
    HIPC(hipMemsetAsync(P.corr_counter, 0, sizeof(unsigned long long), P.stream));
    HIPC(hipEventRecord(P.ev_begin, P.stream));
    ekf::CmdSrc src{};
    src.mode = ekf::SRC_ASSOC;
    src.assoc = P.pv.assoc;
    src.fresh_pose = 1;
    src.meas_stride = jmax * 2;
    ekf::PoolView pva = P.pv;
    // Host bound of every filter's known_count, slot by slot: landmarks are appended in discovery order
    // (ekf_slam.cpp:318-327), one per measurement at most, so known_count_b <= (its last known value) +
    // (measurements of b since).  It sizes the launches; each filter narrows its own correction to its real
    // prefix on the device (CorrRec.n_active).  The bound loosens by up to jmax per step, so the real counts are
    // read back now and then: every step for a big pool (a step is milliseconds of device work there), rarely
    // for a small one (where the read-back's stream sync would dominate).
    const int refresh_every = B >= 64 ? 1 : 16;
    std::vector<ekf::AssocRec> recs(B);
    std::vector<int> kc(B, 0);
    size_t k = 0;
    int kc_max = 0;
    // Small discovered prefixes: while every filter's 3 + 2*(known_count + readings of the step) fits the
    // LDS-resident path, ONE launch per step does all scoring, gating, initialisation and corrections of the step
    // for the whole pool (k_pool_associate) instead of four launches per measurement slot.
    const bool want_small = P.small_path && P.active_prefix && n > 0;
    for (int t = t_begin; t < t_end; t++) {
        bool fresh = false;
        auto refresh = [&]() -> ekf_status {
            EKFC(P.download(recs.data(), P.pv.assoc, sizeof(ekf::AssocRec) * B));
            for (int b = 0; b < B; b++) kc[b] = recs[b].known_count;
            fresh = true;
            return EKF_OK;
        };
        if ((t - t_begin) % refresh_every == 0) EKFC(refresh());
        const int* ct = P.ucount_host.data() + (size_t)t * B;
        int smax = 0;
        for (int b = 0; b < B; b++) if (ct[b] > smax) smax = ct[b];
        auto step_dim = [&]() {  // bound of every filter's active dimension after this step
            int m = P.touched_hwm;
            for (int b = 0; b < B; b++) if (kc[b] + ct[b] > m) m = kc[b] + ct[b];
            if (m > n) m = n;
            return 3 + 2 * m;
        };
        int Nstep = step_dim();
        if (want_small && smax > 0 && Nstep > ekf::small_max_dim() && !fresh) {  // is it only the bound that is loose?
            EKFC(refresh());
            Nstep = step_dim();
        }
        {   // prediction(), confined to the discovered prefix of the pool (exact: zeros map to zeros)
            ekf::PoolView pvp = P.pv;
            if (P.active_prefix) {
                int m = P.touched_hwm;
                for (int b = 0; b < B; b++) if (kc[b] > m) m = kc[b];
                if (m < n) pvp.N = 3 + 2 * m;
            }
            ekf::launch_predict(pvp, P.ulog_twist + (size_t)t * B * 2, 0.0, 0.0, P.pending(), P.stream);
        }
        if (want_small && smax > 0 && Nstep <= ekf::small_max_dim()) {
            pva.N = Nstep;
            if ((Nstep - 3) / 2 > kc_max) kc_max = (Nstep - 3) / 2;
            if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
            ekf::launch_pool_associate(pva, P.ulog_meas + (size_t)t * B * jmax * 2, P.ulog_count + (size_t)t * B, jmax,
                                       3 + 2 * P.touched_hwm, P.ulog_assoc + (size_t)t * B * jmax, P.corr_counter, P.stream);
            if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
            k++;
            smax = 0;  // the step is done
        }
        for (int j = 0; j < smax; j++) {  // ekf_slam.cpp:291: sequential, state-carrying
            int m_before = 0, m = 0;      // bounds of known_count before / after this slot's decision
            for (int b = 0; b < B; b++) {
                const int vb = kc[b] + (ct[b] > j ? j : ct[b]), va = kc[b] + (ct[b] > j ? j + 1 : ct[b]);
                if (vb > m_before) m_before = vb;
                if (va > m) m = va;
            }
            if (m > n) m = n;
            if (m_before > n) m_before = n;
            if (m > kc_max) kc_max = m;
            const ekf::MeasSrc ms{P.ulog_meas + ((size_t)t * B * jmax + j) * 2, jmax * 2, P.ulog_count + (size_t)t * B, j};
            if (P.active_prefix) {
                if (P.touched_hwm > m) m = P.touched_hwm;
                pva.N = 3 + 2 * m;  // launch bound over the pool; every filter narrows it to its own prefix
                src.min_active = 3 + 2 * P.touched_hwm;
            }
            ekf::launch_maha(P.pv, ms, P.scores, -1, m_before, P.stream);
            ekf::launch_assoc_decide(P.pv, ms, P.scores, P.ulog_assoc + (size_t)t * B * jmax, jmax, j, P.corr_counter,
                                     P.stream);
            src.meas = ms.xy;
            ekf::launch_gain(pva, src, P.stream);
            if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
            ekf::launch_rank2(pva, P.tuning, P.stream);
            if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
            k++;
        }
        for (int b = 0; b < B; b++) { kc[b] += ct[b]; if (kc[b] > n) kc[b] = n; }
    }
    HIPC(hipEventRecord(P.ev_end, P.stream));
    EKFC(checked_launch());
    unsigned long long corr = 0;
    EKFC(P.download(&corr, P.corr_counter, sizeof(corr)));
    EKFC(P.download(recs.data(), P.pv.assoc, sizeof(ekf::AssocRec) * B));
    kc_max = 0;  // the real high-water mark replaces the slot-by-slot bound
    for (int b = 0; b < B; b++) if (recs[b].known_count > kc_max) kc_max = recs[b].known_count;
    if (kc_max > n) kc_max = n;
    if (kc_max > P.touched_hwm) P.touched_hwm = kc_max;
    P.touched_bound = P.touched_bound + kc_max < n ? P.touched_bound + kc_max : n;
    P.touch_bound_base = P.touched_bound;
    if (stats) {
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, P.ev_begin, P.ev_end));
        stats->elapsed_ms = ms;
        stats->rank2_ms = 0.0;
        stats->rank2_launches = (long long)k;
        if (ev)
            for (size_t i = 0; i < k; i++) {
                float m = 0.f;
                HIPC(hipEventElapsedTime(&m, ev[2 * i], ev[2 * i + 1]));
                stats->rank2_ms += m;
            }
        stats->corrections = (long long)corr;
        stats->filter_steps = (long long)B * (t_end - t_begin);
        // a correction streams only the discovered prefix; the dense figure is the upper bound
        stats->rank2_bytes_per_launch = 0.0;
    }
    return EKF_OK;
}

ekf_status ekf_batch_get_known_counts(ekf_batch_handle hb, int* out) {
    if (!hb || !out) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    EKFC(P.use());
    std::vector<ekf::AssocRec> recs(P.pv.B);
    EKFC(P.download(recs.data(), P.pv.assoc, sizeof(ekf::AssocRec) * P.pv.B));
    for (int b = 0; b < P.pv.B; b++) out[b] = recs[b].known_count;
    return EKF_OK;
}

ekf_status ekf_batch_get_decisions(ekf_batch_handle hb, int* out) {
    if (!hb || !out) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    if (P.uT <= 0) return fail(EKF_ERR_STATE, "no unknown-association log on the device");
    EKFC(P.use());
    return P.download(out, P.ulog_assoc, sizeof(int) * (size_t)P.uT * P.pv.B * P.ujmax);
}

ekf_status ekf_batch_get_state(ekf_batch_handle hb, int b, double* out) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    return hb->pool.get_state(b, out);
}

ekf_status ekf_batch_get_cov(ekf_batch_handle hb, int b, double* out) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    return hb->pool.get_cov(b, out);
}

ekf_status ekf_batch_get_poses(ekf_batch_handle hb, double* out) {
    if (!hb || !out) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    EKFC(P.use());
    ekf::launch_gather_poses(P.pv, P.poses_dev, P.stream);
    EKFC(checked_launch());
    return P.download(out, P.poses_dev, sizeof(double) * 3 * P.pv.B);
}

ekf_status ekf_batch_checksum(ekf_batch_handle hb, double out[4]) {
    if (!hb || !out) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    EKFC(P.use());
    EKFC(P.flush());
    HIPC(hipMemsetAsync(P.digest_dev, 0, sizeof(double) * 4 * P.pv.B, P.stream));
    ekf::launch_checksum(P.pv, P.digest_dev, P.stream);
    EKFC(checked_launch());
    std::vector<double> h((size_t)4 * P.pv.B);
    EKFC(P.download(h.data(), P.digest_dev, sizeof(double) * h.size()));
    for (int k = 0; k < 4; k++) out[k] = 0.0;
    for (int b = 0; b < P.pv.B; b++)
        for (int k = 0; k < 4; k++) out[k] += h[(size_t)b * 4 + k];
    return EKF_OK;
}

// ---- laser-scan front end (f3) -------------------------------------------------------------------

ekf_status ekf_circle_fit_scans(int device, const double* ranges, int S, int n_beams, int max_out, double* centres,
                                double* radii, int* counts, double* all_clusters, int* n_clusters) {
    if (!ranges || !centres || !radii || !counts || S < 0 || n_beams < 1 || max_out < 1 ||
        n_beams > ekf::circles_max_beams())
        return fail(EKF_ERR_INVALID, "ekf_circle_fit_scans: bad argument (n_beams must be 1..1024)");
    if (S == 0) return EKF_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= count) return fail(EKF_ERR_INVALID, "device index out of range");
    HIPC(hipSetDevice(device));
    const int mc = ekf::circles_max_clusters();
    double *d_r = nullptr, *d_c = nullptr, *d_rad = nullptr, *d_all = nullptr;
    int *d_cnt = nullptr, *d_nc = nullptr;
    ekf_status st = EKF_OK;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_r, sizeof(double) * (size_t)S * n_beams));
        HIPC(hipMalloc((void**)&d_c, sizeof(double) * (size_t)S * max_out * 2));
        HIPC(hipMalloc((void**)&d_rad, sizeof(double) * (size_t)S * max_out));
        HIPC(hipMalloc((void**)&d_cnt, sizeof(int) * (size_t)S));
        HIPC(hipMalloc((void**)&d_nc, sizeof(int) * (size_t)S));
        if (all_clusters) HIPC(hipMalloc((void**)&d_all, sizeof(double) * (size_t)S * mc * 4));
        HIPC(hipMemcpy(d_r, ranges, sizeof(double) * (size_t)S * n_beams, hipMemcpyHostToDevice));
        HIPC(hipMemset(d_c, 0, sizeof(double) * (size_t)S * max_out * 2));
        HIPC(hipMemset(d_rad, 0, sizeof(double) * (size_t)S * max_out));
        if (d_all) HIPC(hipMemset(d_all, 0, sizeof(double) * (size_t)S * mc * 4));
        ekf::launch_circles(d_r, S, n_beams, max_out, d_c, d_rad, d_cnt, d_all, d_nc, nullptr);
        HIPC(hipGetLastError());
        HIPC(hipMemcpy(centres, d_c, sizeof(double) * (size_t)S * max_out * 2, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(radii, d_rad, sizeof(double) * (size_t)S * max_out, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(counts, d_cnt, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost));
        if (n_clusters) HIPC(hipMemcpy(n_clusters, d_nc, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost));
        if (all_clusters) HIPC(hipMemcpy(all_clusters, d_all, sizeof(double) * (size_t)S * mc * 4, hipMemcpyDeviceToHost));
        return EKF_OK;
    };
    st = body();
    for (void* p : {(void*)d_r, (void*)d_c, (void*)d_rad, (void*)d_cnt, (void*)d_nc, (void*)d_all})
        if (p) (void)hipFree(p);
    return st;
}

// ---- dense fp32 propagation (configs[3]) -----------------------------------------------------

ekf_status ekf_dense_create(int N, int device, ekf_dense_handle* out) {
    if (!out || N <= 0) return fail(EKF_ERR_INVALID, "ekf_dense_create: bad argument");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= count) return fail(EKF_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(EKF_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    ekf_dense_s* d = new (std::nothrow) ekf_dense_s();
    if (!d) return fail(EKF_ERR_NOMEM, "host allocation failed");
    d->device = device;
    d->N = N;
    d->ld = round_up(N, ekf::kDenseTile);
    const size_t bytes = sizeof(float) * (size_t)d->ld * d->ld;
    ekf_status st = EKF_OK;
    auto body = [&]() -> ekf_status {
        HIPC(hipSetDevice(device));
        HIPC(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        HIPC(hipStreamCreateWithFlags(&d->stream2, hipStreamNonBlocking));
        HIPC(hipEventCreateWithFlags(&d->j1, hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&d->j2, hipEventDisableTiming));
        HIPC(ekf::dense_gemm_prepare());
        for (float** p : {&d->F, &d->S, &d->T, &d->Q}) {
            HIPC(hipMalloc((void**)p, bytes));
            HIPC(hipMemsetAsync(*p, 0, bytes, d->stream));
        }
        HIPC(hipEventCreate(&d->e0));
        HIPC(hipEventCreate(&d->e1));
        HIPC(hipStreamSynchronize(d->stream));
        return EKF_OK;
    };
    st = body();
    if (st != EKF_OK) {
        ekf_dense_destroy(d);
        return st;
    }
    *out = d;
    return EKF_OK;
}

ekf_status ekf_dense_destroy(ekf_dense_handle d) {
    if (!d) return EKF_OK;
    if (d->device >= 0) (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    for (float* p : {d->F, d->S, d->T, d->Q})
        if (p) (void)hipFree(p);
    if (d->stream2) (void)hipStreamSynchronize(d->stream2);
    for (hipEvent_t e : {d->e0, d->e1, d->j1, d->j2})
        if (e) (void)hipEventDestroy(e);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->stream2) (void)hipStreamDestroy(d->stream2);
    delete d;
    return EKF_OK;
}

ekf_status ekf_dense_set(ekf_dense_handle d, const float* F, const float* Sigma, const float* Q) {
    if (!d) return fail(EKF_ERR_INVALID, "null handle");
    HIPC(hipSetDevice(d->device));
    const size_t w = sizeof(float) * d->N, pitch = sizeof(float) * d->ld;
    const float* src[3] = {F, Sigma, Q};
    float* dst[3] = {d->F, d->S, d->Q};
    for (int i = 0; i < 3; i++)
        if (src[i]) HIPC(hipMemcpy2DAsync(dst[i], pitch, src[i], w, w, d->N, hipMemcpyHostToDevice, d->stream));
    HIPC(hipStreamSynchronize(d->stream));
    return EKF_OK;
}

ekf_status ekf_dense_propagate(ekf_dense_handle d, int iterations, double* elapsed_ms) {
    if (!d || iterations < 0) return fail(EKF_ERR_INVALID, "ekf_dense_propagate: bad argument");
    HIPC(hipSetDevice(d->device));
    HIPC(hipEventRecord(d->e0, d->stream));
    // both streams meet before and after every product: the tail kernel on stream2 reads what the previous
    // product wrote on either stream, and the next product reads what both kernels of this one wrote
    auto join = [&]() -> ekf_status {
        HIPC(hipEventRecord(d->j1, d->stream));
        HIPC(hipStreamWaitEvent(d->stream2, d->j1, 0));
        HIPC(hipEventRecord(d->j2, d->stream2));
        HIPC(hipStreamWaitEvent(d->stream, d->j2, 0));
        return EKF_OK;
    };
    for (int it = 0; it < iterations; it++) {
        EKFC(join());
        ekf::launch_dense_gemm(d->F, d->S, d->T, nullptr, d->ld, false, d->stream, d->stream2);  // T = At*sigma (:102)
        EKFC(join());
        ekf::launch_dense_gemm(d->T, d->F, d->S, d->Q, d->ld, true, d->stream, d->stream2);  // sigma = T*At.t() + Q
    }
    EKFC(join());
    HIPC(hipEventRecord(d->e1, d->stream));
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(d->stream));
    if (elapsed_ms) {
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, d->e0, d->e1));
        *elapsed_ms = ms;
    }
    return EKF_OK;
}

ekf_status ekf_dense_get_sigma(ekf_dense_handle d, float* out) {
    if (!d || !out) return fail(EKF_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(d->device));
    const size_t w = sizeof(float) * d->N, pitch = sizeof(float) * d->ld;
    HIPC(hipMemcpy2DAsync(out, w, d->S, pitch, w, d->N, hipMemcpyDeviceToHost, d->stream));
    HIPC(hipStreamSynchronize(d->stream));
    return EKF_OK;
}

}  // extern "C"
