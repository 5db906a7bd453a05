// ekf_runtime.hpp -- host runtime shared by the C-ABI translation units (ekf_capi*.hip): error reporting,
// pinned staging, and Pool = the device state of B filters (covariance pool, scratch, logs, update-mode
// bookkeeping) with its stream-ordered helpers.  No CPU fallback exists: without a gfx950 device every
// entry point fails with EKF_ERR_NO_DEVICE.
#pragma once
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

namespace ekfrt {

inline thread_local std::string g_err;

inline ekf_status fail(ekf_status st, const std::string& msg) {
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
    ekf::Rank2Tuning tuning{0, -1, 0, 1, 1, 1};
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
    int dev_known_count = -1;  // single filter: AssocRec.known_count as last left on the device (-1 = unknown)
    int touched_hwm = 0;    // landmarks [0, touched_hwm) may carry non-constructor covariance (single filter)
    int small_path = 1;     // measurement() of a small map runs as one LDS-resident launch (ekf_small.hip)
    int active_set = 0;     // eager corrections stream only the rows of the touched set (opt-in)
    int touched_bound = 0;  // host-side upper bound of the device touch_count over the pool
    std::vector<unsigned char> host_touched;  // single filter: exact host copy of the touched flags
    std::vector<int> log_touch_bound;         // batch: bound after step t of the uploaded log
    int touch_bound_base = 0;                 // touched_bound when that log arrived (filters may not be fresh)
    unsigned char* visible_dev = nullptr;  // [n] (single filter)

    ekf::AssocRec* assoc_alt = nullptr;  // "next" association record of the fused data_association() step
    double* terms = nullptr;             // one launch per reading (ekf_assocfused.hip): [n][16] terms of the scored landmarks
    double* terms2 = nullptr;            // one launch per reading (ekf_assocfused.hip): terms / scores of the NEXT reading
    double* scores2 = nullptr;

    // Optional per-launch HIP-event timing of a single filter's kernels (bench.py's configs[1] / configs[2] legs):
    // class 0 = launches that stream the covariance (fused correction, rank-2, decision + correction),
    // class 1 = Mahalanobis scoring launches.  Off by default: the events sit on the stream between launches.
    int prof_on = 0;
    std::vector<hipEvent_t> prof_ev;   // pairs (begin, end)
    std::vector<int> prof_cls;
    size_t prof_used = 0;
    double prof_ms[2] = {0.0, 0.0};
    long long prof_launches[2] = {0, 0};
    static constexpr size_t kProfPairs = 2048;
    ekf_status prof_drain() {
        if (prof_used == 0) return EKF_OK;
        HIPC(hipStreamSynchronize(stream));
        for (size_t i = 0; i < prof_used; i++) {
            float m = 0.f;
            HIPC(hipEventElapsedTime(&m, prof_ev[2 * i], prof_ev[2 * i + 1]));
            prof_ms[prof_cls[i]] += m;
            prof_launches[prof_cls[i]]++;
        }
        prof_used = 0;
        return EKF_OK;
    }
    ekf_status prof_begin(int cls) {
        if (!prof_on) return EKF_OK;
        if (prof_used == kProfPairs) EKFC(prof_drain());
        while (prof_ev.size() < 2 * (prof_used + 1)) {
            hipEvent_t e;
            HIPC(hipEventCreate(&e));
            prof_ev.push_back(e);
        }
        if (prof_cls.size() <= prof_used) prof_cls.resize(prof_used + 1);
        prof_cls[prof_used] = cls;
        HIPC(hipEventRecord(prof_ev[2 * prof_used], stream));
        return EKF_OK;
    }
    ekf_status prof_end() {
        if (!prof_on) return EKF_OK;
        HIPC(hipEventRecord(prof_ev[2 * prof_used + 1], stream));
        prof_used++;
        return EKF_OK;
    }

    // measurement() as two launches per call: factor panels + ONE streaming pass over Sigma (ekf_callfused.hip).
    // Exact (bit-identical).  Default for single filters and pools beyond the small-map path (EKF_FORM_CALL_FUSED);
    // bench.py switches it off for the eager per-landmark stream its `value` / `roofline` are quoted on.
    int call_fused = 1;
    int step_fused = 1;            // pools, unknown association beyond the LDS-resident path: one launch per step
    double* cf_U = nullptr;        // [B][2 kCallV][ld] (+ slack)
    double* cf_V = nullptr;
    int* cf_cnt = nullptr;         // [B]
    double* cf_state = nullptr;    // [B][ld] out-of-place state of the factor kernel
    double* call_in = nullptr;     // single filter: [2n] sensor_reading | [1 + n] ints: V, visible landmarks
    double* cf_pred = nullptr;     // [B][2] (A10, A20) of a prediction folded into the call
    double* blk_cache = nullptr;   // [B][25][n] pools' step-fused association: every landmark's current 5 x 5 block
    // pools' delayed association: the old part of a step's gains for the guessed winners (ekf_kernels.hpp, launch_pool_step_spec)
    double* spec = nullptr;        // [B][spec_rows()][ld]
    int* specw = nullptr;          // [B][kCallV]
    int step_speculate = 1;        // EKF_FORM_STEP_SPECULATE
    ekf_status ensure_spec() {
        if (!spec) {
            EKFC(dalloc(&spec, (size_t)pv.B * ekf::spec_rows() * pv.ld));
            EKFC(dalloc(&specw, (size_t)pv.B * ekf::kCallV));
        }
        return EKF_OK;
    }
    ekf_status ensure_blk_cache() {
        if (!blk_cache) EKFC(dalloc(&blk_cache, (size_t)pv.B * 25 * (pv.n > 0 ? pv.n : 1)));
        return EKF_OK;
    }
    bool call_fused_ok() const { return call_fused && pend_cap == 0 && !active_set && pv.n > 0 && pv.N > ekf::small_max_dim(); }
    ekf_status ensure_callfused() {
        if (!cf_U) {
            const size_t cnt = (size_t)pv.B * 2 * ekf::kCallV * pv.ld + 64;
            EKFC(dalloc(&cf_U, cnt));
            EKFC(dalloc(&cf_V, cnt));
            EKFC(dalloc(&cf_cnt, (size_t)pv.B));
            EKFC(dalloc(&cf_state, (size_t)pv.B * pv.ld));
            EKFC(dalloc(&cf_pred, (size_t)pv.B * 2));
        }
        if (pv.B == 1 && !call_in) EKFC(dalloc(&call_in, (size_t)2 * pv.n + (size_t)(pv.n + 2 + 1) / 2));
        if (!assoc_alt) EKFC(dalloc(&assoc_alt, (size_t)pv.B));
        return EKF_OK;
    }
    // corrections [.., ..) of one call for every filter: factor panels, state, then the streaming pass
    ekf_status call_fused_pass(const ekf::CallSrc& src, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
        ekf::launch_call_factors(pv, src, cf_U, cf_V, cf_cnt, cf_state, stream);
        std::swap(pv.state, cf_state);
        if (ev0) HIPC(hipEventRecord(ev0, stream));
        EKFC(prof_begin(0));
        ekf::launch_rank2v(pv, cf_U, cf_V, cf_cnt, src.vcount, tuning, stream, src.has_twist ? cf_pred : nullptr);
        form_counts[3]++;
        EKFC(prof_end());
        if (ev1) HIPC(hipEventRecord(ev1, stream));
        return EKF_OK;
    }

    // execution forms (ekf_set_forms): which launch structures may be taken where they apply
    unsigned forms = EKF_FORMS_DEFAULT;
    long long form_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ekf_batch_form_counts
    long long* phase_trace = nullptr;                // [2][kTraceSlots], only while ekf_phase_trace is on
    ekf_status set_forms(unsigned f) {
        EKFC(use());  // (a prediction deferred under the old setting happens now)
        forms = f;
        small_path = (f & EKF_FORM_SMALL_MAP) ? 1 : 0;
        call_fused = (f & EKF_FORM_CALL_FUSED) ? 1 : 0;
        active_prefix = (f & EKF_FORM_ACTIVE_PREFIX) ? 1 : 0;
        step_fused = (f & EKF_FORM_STEP_FUSED) ? ((f & EKF_FORM_STEP_SPLIT_PASS) ? 1 : 2) : 0;
        delayed_pair = (f & EKF_FORM_DELAYED_PAIR) ? 1 : 0;
        column_panel = (f & EKF_FORM_COLUMN_PANEL) ? ((f & EKF_FORM_COLUMN_PANEL_ONE_SLOT) ? 2 : 1) : 0;
        step_speculate = (f & EKF_FORM_STEP_SPECULATE) ? 1 : 0;
        current_columns = (f & EKF_FORM_CURRENT_COLUMNS) ? 1 : 0;
        tuning.row_packing = (f & EKF_FORM_ROW_PACKING) ? 1 : 0;
        tuning.strip_flush = (f & EKF_FORM_STRIP_FLUSH_ALWAYS) ? 2 : (f & EKF_FORM_STRIP_FLUSH) ? 1 : 0;
        tuning.tile_queue = (f & EKF_FORM_TILE_QUEUE) ? 1 : 0;
        return EKF_OK;
    }
    void set_tuning(int rows_per_block, int nontemporal, int group_rows) {
        tuning.rows_per_block = rows_per_block > 0 ? rows_per_block : 0;
        tuning.nontemporal = nontemporal;
        tuning.group_rows = group_rows;
    }

    ekf::Pending pending() const {
        ekf::Pending p{Uf, Vf, pend_cap, pend_count, pend_symmetric};
        if (panel_active) {
            p.colp = colp; p.lmslot = lmslot; p.colp_rows = colp_rows(); p.uvc = uvc;
            if (current_columns && cur) {
                p.cur = cur; p.curv_in = curv[curv_sel]; p.curv_out = curv[curv_sel ^ 1];
                p.apred_in = apred[curv_sel]; p.apred_out = apred[curv_sel ^ 1];
            }
        }
        return p;
    }

    // Column panel of delayed known-association runs (ekf_kernels.hpp, Pending::colp; EKF_FORM_COLUMN_PANEL): buffers, and
    // whether the panel currently holds the columns of the materialised covariance (panel_valid: set by a run whose
    // closing flush wrote it, cleared by ANY later entry into the handle -- use() -- so a panel never outlives a change
    // of Sigma that did not go through it).  panel_active: inside ekf_batch_run_known's loop only.
    double* colp = nullptr;       // [B][3 + 2 slots][ld]
    short* lmslot = nullptr;      // [B][n]
    int* plan_list = nullptr;     // [B][slots]
    ekf::double2_t* uvc = nullptr;   // [B][3 + 2 slots][cap]: (U_j(c), V_j(c)) of the pending factors at the panel's indices
    int colp_slots = 0;
    int column_panel = 1;         // 0 off, 1 on, 2 on with a ONE-slot plan (test hook: the gather fallback beside the panel)
    // ... current rows / columns of the pose indices and the planned landmarks (Pending::cur; EKF_FORM_CURRENT_COLUMNS)
    double* cur = nullptr;        // [B][6 + 4 slots][ld]
    int* curv[2] = {nullptr, nullptr};   // [B][1 + slots] each: a paired gain launch reads one and writes the other
    double* apred[2] = {nullptr, nullptr};   // [B][2] each: the deferred prediction map of the kept pose vectors, same ping-pong
    int curv_sel = 0;
    int current_columns = 1;
    bool panel_valid = false, panel_active = false;
    int colp_rows() const { return 3 + 2 * colp_slots; }
    ekf_status free_panel() {
        if (colp) HIPC(hipFree(colp));
        if (lmslot) HIPC(hipFree(lmslot));
        if (plan_list) HIPC(hipFree(plan_list));
        if (uvc) HIPC(hipFree(uvc));
        if (cur) HIPC(hipFree(cur));
        for (int q = 0; q < 2; q++) if (curv[q]) HIPC(hipFree(curv[q]));
        for (int q = 0; q < 2; q++) if (apred[q]) HIPC(hipFree(apred[q]));
        apred[0] = apred[1] = nullptr;
        colp = nullptr; lmslot = nullptr; plan_list = nullptr; uvc = nullptr; cur = nullptr; curv[0] = curv[1] = nullptr;
        colp_slots = 0;
        panel_valid = panel_active = false;
        return EKF_OK;
    }
    ekf_status ensure_panel(bool* fresh) {   // *fresh: the buffers were (re)created -- whatever panel was valid is gone
        int want = column_panel == 2 ? 1 : pend_cap / 2;
        if (want > 64) want = 64;
        if (want < 1) want = 1;
        *fresh = false;
        if (colp && colp_slots == want) return EKF_OK;
        *fresh = true;
        HIPC(hipStreamSynchronize(stream));
        EKFC(free_panel());
        colp_slots = want;
        const size_t np = (size_t)pv.B * colp_rows() * pv.ld, nl = (size_t)pv.B * (pv.n > 0 ? pv.n : 1), nq = (size_t)pv.B * want;
        HIPC(hipMalloc((void**)&colp, np * sizeof(double)));
        HIPC(hipMalloc((void**)&lmslot, nl * sizeof(short)));
        HIPC(hipMalloc((void**)&plan_list, nq * sizeof(int)));
        HIPC(hipMalloc((void**)&uvc, (size_t)pv.B * colp_rows() * pend_cap * sizeof(ekf::double2_t)));
        const size_t nc = (size_t)pv.B * (6 + 4 * want) * pv.ld, nv = (size_t)pv.B * (1 + want);
        HIPC(hipMalloc((void**)&cur, nc * sizeof(double)));
        HIPC(hipMemsetAsync(cur, 0, nc * sizeof(double), stream));        // (pad entries of the rows stay 0)
        for (int q = 0; q < 2; q++) {
            HIPC(hipMalloc((void**)&curv[q], nv * sizeof(int)));
            HIPC(hipMemsetAsync(curv[q], 0xFF, nv * sizeof(int), stream));   // -1: nothing kept
            HIPC(hipMalloc((void**)&apred[q], (size_t)pv.B * 2 * sizeof(double)));
            HIPC(hipMemsetAsync(apred[q], 0, (size_t)pv.B * 2 * sizeof(double), stream));
        }
        curv_sel = 0;
        HIPC(hipMemsetAsync(colp, 0, np * sizeof(double), stream));       // (pad entries of the rows stay 0)
        HIPC(hipMemsetAsync(lmslot, 0xFF, nl * sizeof(short), stream));   // -1: no landmark has a slot
        HIPC(hipMemsetAsync(plan_list, 0xFF, nq * sizeof(int), stream));
        return EKF_OK;
    }

    ekf_status set_update_mode(int max_pending_corrections, int symmetric_gather) {
        EKFC(use());
        EKFC(flush());
        pend_symmetric = symmetric_gather ? 1 : 0;
        HIPC(hipStreamSynchronize(stream));
        for (double** p : {&Uf, &Vf, &state_alt})
            if (*p) { HIPC(hipFree(*p)); *p = nullptr; }
        EKFC(free_panel());
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
    ekf_status flush(const ekf::PanelIO& panel = ekf::PanelIO{nullptr, nullptr, nullptr, 0}) {
        if (pend_count > 0) {
            form_counts[ekf::launch_flush(pv, pending(), tuning, stream, panel)]++;
            HIPC(hipGetLastError());
            pend_count = 0;
        }
        return EKF_OK;
    }

    // one landmark correction, eager (gain + covariance stream) or delayed (gain only, factors appended).
    // active_N > 0: the correction is exactly confined to the leading active_N block (data_association()).
    // delayed mode, known-association log slots src.v and src.v + 1 in one launch (the pending factors are read once for
    // both corrections); the caller has made room for two pairs
    int delayed_pair = 1;
    ekf_status correct_pair(const ekf::CmdSrc& src) {
        ekf::launch_gain_delayed_pair(pv, src, pending(), state_alt, stream);
        form_counts[2]++;
        if (panel_active) form_counts[7]++;
        if (panel_active && current_columns && cur) curv_sel ^= 1;   // (the launch wrote the other count buffer)
        std::swap(pv.state, state_alt);
        pend_count += 4;
        return EKF_OK;
    }
    ekf_status correct(const ekf::CmdSrc& src, int active_N = 0) {
        if (pend_cap > 0 && (src.mode != ekf::SRC_ASSOC || pv.B == 1)) {
            // (a measurement that data_association() drops appends a zero pair: the state buffers swap either way)
            if (pend_count + 2 > pend_cap) EKFC(flush());
            ekf::launch_gain_delayed(pv, src, pending(), state_alt, stream);
            if (panel_active) form_counts[7]++;
            std::swap(pv.state, state_alt);
            pend_count += 2;
            return EKF_OK;
        }
        EKFC(flush());
        ekf::PoolView view = pv;
        if (active_N > 0 && active_N < pv.N) view.N = active_N;
        ekf::launch_gain(view, src, stream);
        EKFC(prof_begin(0));
        if (active_set && active_N == 0) ekf::launch_rank2_active(pv, tuning, touched_bound, stream);
        else ekf::launch_rank2(view, tuning, stream, active_N == 0 && src.mode != ekf::SRC_ASSOC);
        EKFC(prof_end());
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

    // Small single filters: prediction() is not launched by itself but handed to the LDS-resident launch of the
    // measurement() / data_association() call that follows it in the node loop (one launch per step instead of
    // two).  Anything else that looks at the filter first makes it happen: every entry point passes through use().
    bool pred_pending = false;
    double pred_dth = 0.0, pred_dx = 0.0;
    bool defer_predict_ok() const {
        return (pv.B == 1 && small_path && pend_cap == 0 && pv.n > 0 && pv.N <= ekf::small_max_dim()) ||
               (pv.B == 1 && call_fused_ok());
    }
    void launch_predict_now(double dth, double dx) {
        // Rows/columns of landmarks this object never corrected are exactly zero against the pose block
        // (constructor values), and At*0*At^T + 0 = 0: the propagation is confined to the touched prefix.
        ekf::PoolView view = pv;
        if (active_prefix && pend_cap == 0 && touched_hwm < pv.n) view.N = 3 + 2 * touched_hwm;
        ekf::launch_predict(view, nullptr, dth, dx, pending(), stream);
    }
    // Device error word (PoolView::err): sticky.  A kernel that gave up (a bounded in-kernel hand-off that never arrived)
    // has left results nobody may use: every later entry point of the handle fails with EKF_ERR_HIP.
    unsigned* err_host = nullptr;
    ekf_status check_device() const {
        if (err_host && *reinterpret_cast<volatile unsigned*>(err_host) != 0u)
            return fail(EKF_ERR_HIP, "a kernel reported a device-side error (word " + std::to_string(*err_host) +
                                         ": an in-kernel hand-off timed out); the handle's results are invalid");
        return EKF_OK;
    }
    ekf_status use(bool settle = true) {
        HIPC(hipSetDevice(device));
        EKFC(check_device());
        panel_valid = panel_active = false;   // (see colp: whoever enters the handle may change Sigma)
        if (settle && pred_pending) {
            pred_pending = false;
            launch_predict_now(pred_dth, pred_dx);
            HIPC(hipGetLastError());
        }
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
        pv.ld = ekf::pick_ld(pv.N);
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
        HIPC(hipHostMalloc((void**)&err_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *err_host = 0u;
        pv.err = err_host;   // (unified addressing: the mapped pointer is valid on the device)
        HIPC(hipMalloc((void**)&pv.queue, 64));   // (the resident streaming kernels' tile queue: one word, zeroed per launch)
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
        dev_known_count = 0;  // k_init resets the association record
        return EKF_OK;
    }

    void destroy() {
        if (device >= 0) (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (assoc_block) { pv.assoc = nullptr; assoc_alt = nullptr; assoc_out_dev = nullptr; }   // (parts of the block)
        void* ptrs[] = {pv.sigma, pv.state, pv.Kg, pv.Gh, pv.snap, pv.rec, pv.assoc, pv.touch_flag, pv.touch_list,
                        pv.touch_count, scores, meas_dev, assoc_block,
                        assoc_out_dev, sensor_dev, digest_dev, poses_dev, log_twist, log_lm, log_z, log_init,
                        Uf, Vf, state_alt, assoc_alt, terms, log_truth, ulog_twist, ulog_count, ulog_meas, ulog_assoc, ulog_truth, corr_counter,
                        phase_trace, terms2, scores2, blk_cache, cf_U, cf_V, cf_cnt, cf_state, call_in, cf_pred,
                        colp, lmslot, plan_list, uvc, spec, specw, cur, curv[0], curv[1], apred[0], apred[1], pv.queue};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        stage_in.release();
        stage_out.release();
        if (pub_host) (void)hipHostFree(pub_host);
        pub_host = nullptr;
        if (err_host) (void)hipHostFree(err_host);
        err_host = nullptr;
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e);
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }

    ekf_status sync() {
        EKFC(use());
        HIPC(hipStreamSynchronize(stream));
        return check_device();
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
        EKFC(check_device());
        std::memcpy(dst, stage_out.host, bytes);
        return EKF_OK;
    }

    // two device -> host pieces, ONE stream synchronisation (piece 2 may be empty)
    ekf_status download2(void* dst1, const void* src1, size_t bytes1, void* dst2, const void* src2, size_t bytes2) {
        const size_t off2 = (bytes1 + 63) / 64 * 64;
        EKFC(stage_out.reserve(off2 + bytes2));
        char* hostp = static_cast<char*>(stage_out.host);
        HIPC(hipMemcpyAsync(hostp, src1, bytes1, hipMemcpyDeviceToHost, stream));
        if (bytes2) HIPC(hipMemcpyAsync(hostp + off2, src2, bytes2, hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        EKFC(check_device());
        std::memcpy(dst1, hostp, bytes1);
        if (bytes2) std::memcpy(dst2, hostp + off2, bytes2);
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
        EKFC(check_device());
        std::memcpy(out, stage_out.host, w * pv.N);
        return EKF_OK;
    }

    ekf_status set_cov(int b, const double* in) {
        if (!in || b < 0 || b >= pv.B) return fail(EKF_ERR_INVALID, "set_cov: bad argument");
        EKFC(use());
        EKFC(flush());
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

    // Single filter: the two association records (they ping-pong between launches) and the decisions of a call live in
    // ONE block -- [record | record | decisions] -- so that data_association() ends with one device-to-host copy.
    char* assoc_block = nullptr;
    static constexpr size_t kAssocRecSlot = ekf::kAssocRecSlot, kAssocSeqOff = ekf::kAssocSeqOff, kAssocDecOff = ekf::kAssocDecOff;
    // ... and its mirror in mapped host memory ([record | sequence number | decisions]): k_publish_assoc writes it, the
    // host spins on the sequence number -- no copy engine, no stream synchronisation at the end of a call
    char* pub_host = nullptr;
    unsigned pub_seq = 0;
    bool pub_sent = false;
    // a deciding kernel publishes by itself (k_assoc_call / k_assoc_reading): the number its last launch will write
    unsigned claim_pub_seq() { pub_sent = true; return ++pub_seq; }
    ekf_status publish_assoc(int J) {
        if (!pub_host || pub_sent) return EKF_OK;
        ekf::launch_publish_assoc(pv.assoc, assoc_out_dev, J, pub_host, ++pub_seq, stream);
        pub_sent = true;
        return EKF_OK;
    }
    ekf_status ensure_meas_capacity(int J) {
        if (J <= jcap) return EKF_OK;
        HIPC(hipStreamSynchronize(stream));
        if (meas_dev) HIPC(hipFree(meas_dev));
        meas_dev = nullptr;
        const int cap = J < 64 ? 64 : round_up(J, 64);
        EKFC(dalloc(&meas_dev, (size_t)cap * 2));
        if (pv.B == 1) {
            char* blk = nullptr;
            EKFC(dalloc(&blk, kAssocDecOff + sizeof(int) * (size_t)cap));
            HIPC(hipMemsetAsync(blk, 0, kAssocDecOff, stream));
            // the records move in (the current one first); the old homes are released
            HIPC(hipMemcpyAsync(blk, pv.assoc, sizeof(ekf::AssocRec), hipMemcpyDeviceToDevice, stream));
            if (assoc_alt) HIPC(hipMemcpyAsync(blk + kAssocRecSlot, assoc_alt, sizeof(ekf::AssocRec), hipMemcpyDeviceToDevice, stream));
            HIPC(hipStreamSynchronize(stream));
            if (assoc_block) HIPC(hipFree(assoc_block));
            else {
                HIPC(hipFree(pv.assoc));
                if (assoc_alt) HIPC(hipFree(assoc_alt));
            }
            assoc_block = blk;
            if (pub_host) HIPC(hipHostFree(pub_host));
            pub_host = nullptr;
            HIPC(hipHostMalloc((void**)&pub_host, kAssocDecOff + sizeof(int) * (size_t)cap,
                               hipHostMallocMapped | hipHostMallocCoherent));   // (the host spins on it: fine-grained)
            std::memset(pub_host, 0, kAssocDecOff);
            pv.assoc = reinterpret_cast<ekf::AssocRec*>(blk);
            assoc_alt = reinterpret_cast<ekf::AssocRec*>(blk + kAssocRecSlot);
            assoc_out_dev = reinterpret_cast<int*>(blk + kAssocDecOff);
        } else {
            if (assoc_out_dev) HIPC(hipFree(assoc_out_dev));
            assoc_out_dev = nullptr;
            EKFC(dalloc(&assoc_out_dev, (size_t)cap));
        }
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

inline ekf_status checked_launch() {
    HIPC(hipGetLastError());
    return EKF_OK;
}

inline ekf_status free_log(Pool& P) {
    HIPC(hipStreamSynchronize(P.stream));
    for (void* p : {(void*)P.log_twist, (void*)P.log_lm, (void*)P.log_z, (void*)P.log_init, (void*)P.log_truth})
        if (p) HIPC(hipFree(p));
    P.log_twist = nullptr; P.log_lm = nullptr; P.log_z = nullptr; P.log_init = nullptr; P.log_truth = nullptr;
    P.T = 0; P.vmax = 0; P.log_bytes = 0;
    return EKF_OK;
}

inline ekf_status free_ulog(Pool& P) {
    HIPC(hipStreamSynchronize(P.stream));
    for (void* p : {(void*)P.ulog_twist, (void*)P.ulog_count, (void*)P.ulog_meas, (void*)P.ulog_assoc, (void*)P.ulog_truth})
        if (p) HIPC(hipFree(p));
    P.ulog_twist = nullptr; P.ulog_count = nullptr; P.ulog_meas = nullptr; P.ulog_assoc = nullptr; P.ulog_truth = nullptr;
    P.truth_is_unknown_log = 0;
    P.uT = 0; P.ujmax = 0; P.ulog_bytes = 0;
    P.ucount_host.clear();
    return EKF_OK;
}

}  // namespace ekfrt

struct ekf_filter_s { ekfrt::Pool pool; };
struct ekf_batch_s { ekfrt::Pool pool; };
