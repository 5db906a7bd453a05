// ekf_capi_batch.hip -- C ABI of include/ekfslam.h, pools of B independent filters: device-resident known- and
// unknown-association logs, the run loops, read-back and digests.
#include "ekf_runtime.hpp"

using namespace ekfrt;

extern "C" {

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

ekf_status ekf_batch_set_forms(ekf_batch_handle hb, unsigned forms) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    return hb->pool.set_forms(forms);
}

ekf_status ekf_batch_get_forms(ekf_batch_handle hb, unsigned* forms) {
    if (!hb || !forms) return fail(EKF_ERR_INVALID, "null argument");
    *forms = hb->pool.forms;
    return EKF_OK;
}

ekf_status ekf_batch_form_counts(ekf_batch_handle hb, long long counts[8]) {
    if (!hb || !counts) return fail(EKF_ERR_INVALID, "null argument");
    for (int i = 0; i < 8; i++) counts[i] = hb->pool.form_counts[i];
    return EKF_OK;
}

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
    hb->pool.set_tuning(rows_per_block, nontemporal, group_rows);
    return EKF_OK;
}

ekf_status ekf_batch_rank2_variant(ekf_batch_handle hb, int* group_rows, int* nontemporal, int* threads, int* rows_per_block) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    ekf::rank2_variant(hb->pool.pv, hb->pool.tuning, group_rows, nontemporal, threads, rows_per_block);
    return EKF_OK;
}

ekf_status ekf_batch_rank2_resident(ekf_batch_handle hb, int* resident) {
    if (!hb || !resident) return fail(EKF_ERR_INVALID, "null argument");
    EKFC(hb->pool.use());
    *resident = ekf::rank2_resident(hb->pool.pv, hb->pool.tuning) ? 1 : 0;
    return EKF_OK;
}

ekf_status ekf_batch_set_update_mode(ekf_batch_handle hb, int max_pending_corrections, int symmetric_gather) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    return hb->pool.set_update_mode(max_pending_corrections, symmetric_gather);
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
    const bool panel_valid_in = P.panel_valid;   // (use() clears it: see Pool::colp)
    EKFC(P.use());
    const int B = P.pv.B, vmax = P.vmax;
    P.touched_hwm = P.pv.n;  // a known log corrects arbitrary indices: no discovered-prefix structure afterwards
    size_t launches = 0;
    long long corrections = 0;
    for (int t = t_begin; t < t_end; t++)
        for (int v = 0; v < vmax; v++)
            if (P.slot_active[(size_t)t * vmax + v] > 0) {
                launches++;
                corrections += P.slot_active[(size_t)t * vmax + v];
            }
    const bool delayed = P.pend_cap > 0;
    const bool callf = !delayed && P.call_fused_ok();
    if (callf) EKFC(P.ensure_callfused());
    // worst-case number of covariance passes (rank-2 launches, or flushes in delayed mode) for the events
    // (paired slots flush when FOUR rows no longer fit: up to one flush per two corrections for tiny k)
    const size_t max_passes = delayed ? launches / (size_t)(P.pend_cap >= 4 ? (P.pend_cap / 2 > 2 ? P.pend_cap / 2 - 1 : 1) : 1) + 2 : launches;
    hipEvent_t* ev = nullptr;
    if (time_kernels && launches) {
        ev = P.events(2 * max_passes);
        if (!ev) return fail(EKF_ERR_HIP, "hipEventCreate failed");
    }
    HIPC(hipEventRecord(P.ev_begin, P.stream));
    size_t k = 0;
    bool cols_stale = false;   // (see upper_run below)
    // Column panel (EKF_FORM_COLUMN_PANEL; Pending::colp): every flush of the run writes the columns 0..2 and the columns
    // of the landmarks the NEXT corrections touch (the log is on the device) as contiguous panel rows; while a panel is
    // on, the gain kernels read those columns coalesced and prediction() keeps columns 0..2 current in the panel instead
    // of in the matrix.  A run whose closing flush wrote a panel hands it to the next run (panel_valid).
    const bool panel_capable = P.pend_cap >= 4 && !P.pend_symmetric && P.column_panel && !P.active_set && P.pv.n > 0 &&
                               P.pv.N >= 256 && t_end > t_begin;
    bool panel_fresh = false;
    if (panel_capable) EKFC(P.ensure_panel(&panel_fresh));
    bool panel_on = panel_capable && panel_valid_in && !panel_fresh && P.pend_count == 0;
    bool panel_cols_stale = false;   // predictions since the last flush while the panel was on: matrix columns 1, 2 are behind
    P.panel_active = panel_on;
    // t_next: the first log step with corrections behind this flush (its slots and those of the following steps, up to
    // the capacity of the factor store, are what the panel is planned for)
    auto timed_flush = [&](int t_next) -> ekf_status {
        if (P.pend_count == 0) return EKF_OK;
        ekf::PanelIO pio{nullptr, nullptr, nullptr, 0};
        if (panel_capable) {
            pio.rows = P.colp_rows();
            if (panel_on) pio.in = P.colp;
            if (t_next < P.T) {
                int h = 0, vecs = 0;
                for (int tt = t_next; tt < P.T; tt++) {
                    int a = 0;
                    for (int v = 0; v < vmax; v++) a += P.slot_active[(size_t)tt * vmax + v] > 0;
                    if (h > 0 && vecs + 2 * a > P.pend_cap) break;
                    vecs += 2 * a;
                    h++;
                }
                ekf::launch_panel_plan(P.pv, P.log_lm + (size_t)t_next * B * vmax, h, vmax, P.colp_slots, P.lmslot,
                                       P.plan_list, P.stream, P.curv[P.curv_sel], P.apred[P.curv_sel]);
                pio.out = P.colp;
                pio.lmslot = P.lmslot;
            }
        }
        P.panel_active = false;   // (the flush itself takes the panel through pio)
        if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
        EKFC(P.flush(pio));
        if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
        k++;
        cols_stale = false;    // (a mirroring flush rewrites everything below the diagonal squares)
        panel_on = pio.out != nullptr;
        panel_cols_stale = false;
        P.panel_active = panel_on;
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
    // symmetric option with mirroring flushes: between flushes the tiles on and above the diagonal are the covariance
    // (Pending::symmetric == 2: the prediction leaves the strided column entries below the first square alone)
    const bool upper_run = delayed && ekf::sym_flush_applies(P.pv, P.pending(), P.tuning);
    for (int t = small_run ? t_end : t_begin; t < t_end; t++) {
        ekf::Pending pdp = P.pending();
        if (upper_run) { pdp.symmetric = 2; cols_stale = true; }
        if (panel_on) panel_cols_stale = true;
        ekf::launch_predict(P.pv, P.log_twist + (size_t)t * B * 2, 0.0, 0.0, pdp, P.stream);  // prediction()
        ekf::launch_measure_begin(P.pv, P.log_init, !P.init_flag, P.stream);               // measurement() top
        P.init_flag = 1;
        src.lm_idx = P.log_lm + (size_t)t * B * vmax;
        src.z_xy = P.log_z + (size_t)t * B * vmax * 2;
        if ((size_t)t < P.log_touch_bound.size()) {
            int cand = P.touch_bound_base + P.log_touch_bound[t];
            if (cand > P.pv.n) cand = P.pv.n;
            if (cand > P.touched_bound) P.touched_bound = cand;
        }
        if (callf) {   // the whole call of every filter: factor panels + ONE streaming pass per kCallV log slots
            int vtop = 0;
            for (int v = 0; v < vmax; v++) if (P.slot_active[(size_t)t * vmax + v] > 0) vtop = v + 1;
            ekf::CallSrc cs{};
            cs.mode = ekf::SRC_COMPACT_LOG;
            cs.lm_idx = src.lm_idx;
            cs.z_xy = src.z_xy;
            cs.vmax = vmax;
            for (int v0 = 0; v0 < vtop; v0 += ekf::kCallV) {
                cs.v0 = v0;
                cs.vcount = vtop - v0 < ekf::kCallV ? vtop - v0 : ekf::kCallV;
                cs.fresh_pose = 0;   // k_measure_begin has recorded the pose of the call
                EKFC(P.call_fused_pass(cs, ev ? ev[2 * k] : nullptr, ev ? ev[2 * k + 1] : nullptr));
                k++;
            }
            continue;
        }
        for (int v = 0; v < vmax; v++) {
            if (P.slot_active[(size_t)t * vmax + v] == 0) continue;
            src.v = v;
            if (delayed) {
                // two consecutive slots in one launch: the pending factor rows are read once for both corrections
                if (P.delayed_pair && P.pend_cap >= 4 && v + 1 < vmax && P.slot_active[(size_t)t * vmax + v + 1] > 0) {
                    if (P.pend_count + 4 > P.pend_cap) EKFC(timed_flush(t));
                    EKFC(P.correct_pair(src));
                    v++;
                    continue;
                }
                if (P.pend_count + 2 > P.pend_cap) EKFC(timed_flush(t));
                EKFC(P.correct(src));
            } else {
                ekf::launch_gain(P.pv, src, P.stream);
                if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
                if (P.active_set) ekf::launch_rank2_active(P.pv, P.tuning, P.touched_bound, P.stream);
                else ekf::launch_rank2(P.pv, P.tuning, P.stream, true);
                P.form_counts[4]++;
                if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
                k++;
            }
        }
    }
    if (delayed) EKFC(timed_flush(t_end));  // every run leaves Sigma materialised
    if (cols_stale) ekf::launch_sym_repair(P.pv, P.stream);   // (predictions after the last flush)
    if (panel_on && panel_cols_stale) ekf::launch_panel_repair(P.pv, P.colp, P.colp_rows(), P.stream);   // (the same, panel form)
    P.panel_active = false;
    P.panel_valid = panel_on;   // the panel holds the columns of the covariance as this run leaves it
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
            (delayed || callf) ? per_pass * (double)B : (launches ? per_pass * (double)corrections / (double)launches : 0.0);
    }
    return EKF_OK;
}

// ---- batched unknown data association ---------------------------------------------------------

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

ekf_status ekf_batch_run_unknown(ekf_batch_handle hb, int t_begin, int t_end, int time_kernels, ekf_run_stats* stats) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.uT <= 0) return fail(EKF_ERR_STATE, "ekf_batch_run_unknown: no unknown-association log uploaded");
    if (t_begin < 0 || t_end > P.uT || t_begin > t_end) return fail(EKF_ERR_INVALID, "step range outside the uploaded log");
    EKFC(P.use());
    const int B = P.pv.B, n = P.pv.n, jmax = P.ujmax;
    // Delayed mode (ekf_batch_set_update_mode(k > 0)): the step-fused form keeps the pairs of a step pending ACROSS steps in
    // the pool's factor store -- jmax pairs per step and filter, Sigma rewritten once per floor(k / jmax) steps instead of
    // once per step.  Everything else (the LDS-resident small-map steps, four launches per slot) works on the materialised
    // covariance and flushes first.
    const bool delayed = P.pend_cap > 0 && P.step_fused && jmax > 0 && jmax <= ekf::kCallV && 2 * jmax <= P.pend_cap &&
                         P.pend_cap / 2 <= ekf::step_pending_pairs_max();
    if (!delayed) EKFC(P.flush());
    P.dev_known_count = -1;
    size_t launches = 0;
    for (int t = t_begin; t < t_end; t++) {
        const int* ct = P.ucount_host.data() + (size_t)t * B;
        int smax = 0;
        for (int b = 0; b < B; b++) if (ct[b] > smax) smax = ct[b];
        launches += smax;
    }
    hipEvent_t* ev = nullptr;
    if (time_kernels && launches) {
        ev = P.events(2 * (launches + (size_t)(t_end - t_begin) + 2));   // (+ flushes of the delayed mode)
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
    auto timed_flush = [&]() -> ekf_status {
        if (P.pend_count == 0) return EKF_OK;
        if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
        EKFC(P.flush());
        if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
        k++;
        return EKF_OK;
    };
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
            if (delayed) EKFC(P.ensure_callfused());
            ekf::launch_predict(pvp, P.ulog_twist + (size_t)t * B * 2, 0.0, 0.0, P.pending(), P.stream,
                                delayed ? P.cf_pred : nullptr);
        }
        // A delayed step in which NO filter has a reading launches no step kernel, so nothing would carry this step's
        // prediction into the block cache (the kernel applies its own step's At / Q to the cached 5 x 5 blocks): fold the
        // pending pairs now -- the next step with readings then rebuilds every block from Sigma (ekf_slam.cpp:55-106,300-309).
        if (delayed && smax == 0) EKFC(timed_flush());
        if (want_small && smax > 0 && Nstep <= ekf::small_max_dim()) {
            if (delayed) EKFC(timed_flush());   // (the LDS-resident step works on the materialised covariance)
            pva.N = Nstep;
            if ((Nstep - 3) / 2 > kc_max) kc_max = (Nstep - 3) / 2;
            if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
            ekf::launch_pool_associate(pva, P.ulog_meas + (size_t)t * B * jmax * 2, P.ulog_count + (size_t)t * B, jmax,
                                       3 + 2 * P.touched_hwm, P.ulog_assoc + (size_t)t * B * jmax, P.corr_counter, P.stream);
            if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
            k++;
            smax = 0;  // the step is done
        }
        if (smax > 0 && delayed) {
            // the step's pairs join the pending store (ekf_stepfused.hip, DELAYED): no pass over Sigma in this step
            EKFC(P.ensure_callfused());
            EKFC(P.ensure_blk_cache());
            if (Nstep > 3 + 2 * kc_max) kc_max = (Nstep - 3) / 2;
            if (P.pend_count + 2 * jmax > P.pend_cap) EKFC(timed_flush());
            // the part of what this step's gains need that does not depend on the step's own readings -- stored entries minus the
            // pairs of earlier steps -- is rebuilt ONCE, for the landmarks the readings are guessed to match (EKF_FORM_STEP_SPECULATE; ekf_stepfused.hip, k_pool_step_spec)
            const bool speculate = P.step_speculate != 0;
            if (speculate) {
                EKFC(P.ensure_spec());
                ekf::launch_pool_step_spec(P.pv, P.ulog_meas + (size_t)t * B * jmax * 2, P.ulog_count + (size_t)t * B, jmax,
                                           P.pending(), P.spec, P.specw, P.stream);
            }
            ekf::launch_pool_step_unknown_delayed(P.pv, P.ulog_meas + (size_t)t * B * jmax * 2, P.ulog_count + (size_t)t * B, jmax,
                                                  P.active_prefix ? 3 + 2 * P.touched_hwm : P.pv.N,
                                                  P.ulog_assoc + (size_t)t * B * jmax, P.pending(), P.corr_counter, P.cf_cnt,
                                                  P.blk_cache, P.cf_pred, P.stream, speculate ? P.spec : nullptr,
                                                  speculate ? P.specw : nullptr);
            P.pend_count += 2 * jmax;
            P.form_counts[5]++;
            smax = 0;  // the step is done
        }
        if (smax > 0 && P.step_fused && jmax <= ekf::kCallV && P.pend_cap == 0) {
            // any prefix size: the whole step of every filter in ONE launch, its covariance streamed once per step
            // (ekf_stepfused.hip); bit-identical to the four launches per measurement slot below
            EKFC(P.ensure_callfused());
            EKFC(P.ensure_blk_cache());
            if (Nstep > 3 + 2 * kc_max) kc_max = (Nstep - 3) / 2;
            // Big prefixes: the step kernel stops at the factor pairs and k_rank2v streams every covariance spread over
            // the whole chip (one workgroup per filter streams its 32 MB at 4.6 TB/s pool-wide, k_rank2v at 6.4).  Chosen
            // when the known counts are fresh (exact) and most of the launch bound is real work: k_rank2v covers the
            // pool-wide bound Nstep for every filter, the one-launch form each filter's own prefix.
            bool split = false;
            if (fresh && Nstep >= 603 && P.step_fused == 1) {
                double real = 0.0;
                for (int b = 0; b < B; b++) {
                    if (ct[b] <= 0) continue;
                    const double nb = 3.0 + 2.0 * (kc[b] + ct[b] < n ? kc[b] + ct[b] : n);
                    real += nb * nb;
                }
                const double span = P.active_prefix ? (double)Nstep : (double)P.pv.N;
                split = real >= 0.7 * B * span * span;
            }
            if (split) {
                pva.N = P.active_prefix ? Nstep : P.pv.N;
                ekf::launch_pool_step_unknown(P.pv, P.ulog_meas + (size_t)t * B * jmax * 2, P.ulog_count + (size_t)t * B, jmax,
                                              P.active_prefix ? 3 + 2 * P.touched_hwm : P.pv.N, P.ulog_assoc + (size_t)t * B * jmax,
                                              P.cf_U, P.cf_V, P.corr_counter, P.stream, P.cf_cnt, ekf::rank2v_round_count(smax),
                                              P.blk_cache);
                if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
                ekf::launch_rank2v(pva, P.cf_U, P.cf_V, P.cf_cnt, smax, P.tuning, P.stream);
                P.form_counts[5]++;
            } else {
                if (ev) HIPC(hipEventRecord(ev[2 * k], P.stream));
                ekf::launch_pool_step_unknown(P.pv, P.ulog_meas + (size_t)t * B * jmax * 2, P.ulog_count + (size_t)t * B, jmax,
                                              P.active_prefix ? 3 + 2 * P.touched_hwm : P.pv.N, P.ulog_assoc + (size_t)t * B * jmax,
                                              P.cf_U, P.cf_V, P.corr_counter, P.stream, nullptr, 0, P.blk_cache);
            }
            if (ev) HIPC(hipEventRecord(ev[2 * k + 1], P.stream));
            k++;
            smax = 0;  // the step is done
        }
        if (smax > 0) EKFC(P.flush());   // (four launches per slot: on the materialised covariance)
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
    if (delayed) EKFC(timed_flush());   // every run leaves Sigma materialised
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

ekf_status ekf_batch_set_known_counts(ekf_batch_handle hb, const int* counts) {
    if (!hb || !counts) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    EKFC(P.use());
    for (int b = 0; b < P.pv.B; b++)
        if (counts[b] < 0 || counts[b] > P.pv.n) return fail(EKF_ERR_INVALID, "known count must lie in 0..n");
    int* dev = nullptr;
    HIPC(hipMalloc((void**)&dev, sizeof(int) * P.pv.B));
    ekf_status st = P.upload(dev, counts, sizeof(int) * P.pv.B);
    if (st == EKF_OK) {
        ekf::launch_assoc_begin(P.pv, dev, 0, P.stream);
        st = checked_launch();
    }
    hipError_t e = hipStreamSynchronize(P.stream);
    (void)hipFree(dev);
    if (st == EKF_OK && e != hipSuccess) st = fail(EKF_ERR_HIP, hipGetErrorString(e));
    if (st != EKF_OK) return st;
    // landmarks below a caller-declared count may carry any covariance: no discovered-prefix structure below it
    int m = 0;
    for (int b = 0; b < P.pv.B; b++) if (counts[b] > m) m = counts[b];
    if (m > P.touched_hwm) P.touched_hwm = m;
    P.dev_known_count = -1;
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

}  // extern "C"
