// ekf_capi.hip -- C ABI of include/ekfslam.h, single-filter entry points: the rigid2d::EKF_SLAM call surface
// (create / clone / prediction / measurement / data_association / getters) and its mode switches.
#include <atomic>

#include "ekf_runtime.hpp"

using namespace ekfrt;

extern "C" {

const char* ekf_last_error(void) { return g_err.c_str(); }

int ekf_leading_dimension(int n_landmarks) { return n_landmarks < 0 ? 0 : ekf::pick_ld(3 + 2 * n_landmarks); }

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
    EKFC(c.set_forms(a.forms));
    c.tuning = a.tuning;
    c.active_set = a.active_set;
    c.pv.active_set = a.active_set;
    c.touched_bound = a.touched_bound;
    c.host_touched = a.host_touched;
    HIPC(hipMemcpyAsync(c.pv.touch_flag, a.pv.touch_flag, (size_t)(a.pv.n > 0 ? a.pv.n : 1), hipMemcpyDeviceToDevice, c.stream));
    HIPC(hipMemcpyAsync(c.pv.touch_list, a.pv.touch_list, sizeof(int) * (size_t)(a.pv.n > 0 ? a.pv.n : 1), hipMemcpyDeviceToDevice, c.stream));
    HIPC(hipMemcpyAsync(c.pv.touch_count, a.pv.touch_count, sizeof(int), hipMemcpyDeviceToDevice, c.stream));
    return c.sync();
}

ekf_status ekf_predict(ekf_handle h, double dtheta, double dx) {
    if (!h) return fail(EKF_ERR_INVALID, "ekf_predict: null handle");
    Pool& P = h->pool;
    EKFC(P.use());  // (a prediction still pending from the previous call happens now)
    if (P.defer_predict_ok()) {  // small map: rides along with the next measurement() / data_association() launch
        P.pred_pending = true;
        P.pred_dth = dtheta;
        P.pred_dx = dx;
        return EKF_OK;
    }
    P.launch_predict_now(dtheta, dx);
    return checked_launch();
}

ekf_status ekf_measure_known(ekf_handle h, const double* sensor_xy, const uint8_t* visible) {
    if (!h || !sensor_xy || !visible) return fail(EKF_ERR_INVALID, "ekf_measure_known: null argument");
    Pool& P = h->pool;
    EKFC(P.use(false));
    const int n = P.pv.n;
    static_assert(3 + 2 * ekf::kSmallInlineN >= 103, "every map of the small path (odd N <= 104) fits the by-value argument");
    if (P.small_path && P.pend_cap == 0 && P.pv.N <= ekf::small_max_dim() && n > 0) {
        // small map (the reference runs n = 20): the whole call -- and the prediction() before it -- in one
        // LDS-resident launch whose inputs travel by value in the kernel arguments (no staging, no copy)
        ekf::SmallInline in;
        std::memcpy(in.sensor, sensor_xy, sizeof(double) * 2 * n);
        std::memcpy(in.visible, visible, (size_t)n);
        ekf::launch_small_measure_inline(P.pv, in, !P.init_flag, P.pred_pending, P.pred_dth, P.pred_dx, P.stream);
        P.pred_pending = false;
        P.init_flag = 1;
        for (int i = n - 1; i >= 0; i--)
            if (visible[i]) { if (i + 1 > P.touched_hwm) P.touched_hwm = i + 1; break; }
        for (int i = 0; i < n; i++)
            if (visible[i]) P.note_touched(i);
        return checked_launch();
    }
    if (P.call_fused_ok()) {
        // beyond the small-map path: the call is two launches whatever the number of visible landmarks -- the factor
        // panels of all its corrections, then ONE read-modify-write pass over Sigma (ekf_callfused.hip); bit-identical
        EKFC(P.ensure_callfused());
        std::vector<int> vl(1, 0);
        for (int i = 0; i < n; i++)
            if (visible[i]) {
                vl.push_back(i);
                if (i + 1 > P.touched_hwm) P.touched_hwm = i + 1;
                P.note_touched(i);
            }
        const int V = (int)vl.size() - 1;
        vl[0] = V;
        ekf::CallSrc cs{};
        cs.trace = P.phase_trace;
        const bool by_value = P.init_flag && V <= ekf::kCallV;   // the first call needs the whole sensor vector (:113-128)
        if (by_value) {
            cs.mode = ekf::SRC_INLINE;
            for (int v = 0; v < ekf::kCallV; v++) {
                cs.inl_lm[v] = v < V ? vl[1 + v] : -1;
                cs.inl_xy[v][0] = v < V ? sensor_xy[2 * vl[1 + v]] : 0.0;
                cs.inl_xy[v][1] = v < V ? sensor_xy[2 * vl[1 + v] + 1] : 0.0;
            }
        } else {
            EKFC(P.upload2(P.call_in, sensor_xy, sizeof(double) * 2 * n, vl.data(), sizeof(int) * vl.size()));
            cs.mode = ekf::SRC_SENSOR_VECTOR;
            cs.sensor = P.call_in;
            cs.vlist = reinterpret_cast<const int*>(P.call_in + 2 * (size_t)n);
        }
        // A prediction() deferred by ekf_predict is folded into the first pass of the call (the factor kernel applies it
        // to its panels on the fly, the streaming pass to every other element): no launch of its own.  The first call
        // (landmark initialisation from the predicted pose) and calls without a visible landmark settle it the plain way.
        const bool fold = P.pred_pending && P.init_flag && V > 0;
        EKFC(P.use(!fold));
        if (!P.init_flag) ekf::launch_measure_begin(P.pv, P.call_in, 1, P.stream);   // first call: every landmark (:113-128)
        P.init_flag = 1;
        for (int v0 = 0; v0 < V; v0 += ekf::kCallV) {
            cs.v0 = v0;
            cs.vcount = V - v0 < ekf::kCallV ? V - v0 : ekf::kCallV;
            cs.fresh_pose = v0 == 0;
            cs.has_twist = fold && v0 == 0;
            cs.dtheta = P.pred_dth;
            cs.dx = P.pred_dx;
            cs.pred_out = P.cf_pred;
            EKFC(P.call_fused_pass(cs));
        }
        if (fold) P.pred_pending = false;
        return checked_launch();
    }
    EKFC(P.upload2(P.sensor_dev, sensor_xy, sizeof(double) * 2 * n, visible, (size_t)n));
    EKFC(P.use());  // settles a deferred prediction()
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

// Tail of every data_association() form: one synchronising read-back of the association record and the
// decisions, then the caller's known_list (:323) and the host-side bounds are brought up to date.
static ekf_status associate_finish(Pool& P, int known_count, int J, uint8_t* known, int* assoc_out) {
    const int n = P.pv.n;
    ekf::AssocRec rec;
    if (P.pub_host) {   // record and decisions through mapped host memory: the host waits for the sequence number only
        EKFC(P.publish_assoc(J));   // (callers with a covariance pass behind the last decision have published in front of it)
        P.pub_sent = false;
        volatile unsigned* seq = reinterpret_cast<volatile unsigned*>(P.pub_host + Pool::kAssocSeqOff);
        for (unsigned spins = 1; *seq != P.pub_seq; spins++) {
            __builtin_ia32_pause();   // (a polite spin: the sibling hyperthread keeps its issue slots; a call is microseconds)
            if ((spins & 0x3FFF) == 0 && hipStreamQuery(P.stream) != hipErrorNotReady) {
                // the stream has drained (or failed): the number must be there now, or never will be
                HIPC(hipStreamSynchronize(P.stream));
                if (*seq != P.pub_seq) return fail(EKF_ERR_HIP, "data_association: the result was not published");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        EKFC(P.check_device());
        std::memcpy(&rec, P.pub_host, sizeof(rec));
        if (assoc_out) std::memcpy(assoc_out, P.pub_host + Pool::kAssocDecOff, sizeof(int) * (size_t)J);
    } else if (P.assoc_block) {   // records and decisions in one block: one copy, one synchronisation
        const size_t bytes = Pool::kAssocDecOff + sizeof(int) * (size_t)J;
        EKFC(P.stage_out.reserve(bytes));
        HIPC(hipMemcpyAsync(P.stage_out.host, P.assoc_block, bytes, hipMemcpyDeviceToHost, P.stream));
        HIPC(hipStreamSynchronize(P.stream));
        const char* hb = static_cast<const char*>(P.stage_out.host);
        const size_t off = reinterpret_cast<const char*>(P.pv.assoc) == P.assoc_block ? 0 : Pool::kAssocRecSlot;
        std::memcpy(&rec, hb + off, sizeof(rec));
        if (assoc_out) std::memcpy(assoc_out, hb + Pool::kAssocDecOff, sizeof(int) * (size_t)J);
    } else {
        EKFC(P.download2(&rec, P.pv.assoc, sizeof(rec), assoc_out, P.assoc_out_dev, assoc_out ? sizeof(int) * J : 0));
    }
    for (int i = known_count; i < rec.known_count && i < n; i++) known[i] = 1;  // :323
    P.dev_known_count = rec.known_count;
    if (rec.known_count > P.touched_hwm) P.touched_hwm = rec.known_count < n ? rec.known_count : n;
    // for the host-side bound every landmark below the new known_count counts as touched (a superset
    // of what this call's decisions actually corrected)
    for (int i = 0; i < rec.known_count && i < n; i++) P.note_touched(i);
    return EKF_OK;
}

ekf_status ekf_associate(ekf_handle h, const double* meas_xy, int J, uint8_t* known, int* assoc_out) {
    if (!h || !known || J < 0 || (J > 0 && !meas_xy)) return fail(EKF_ERR_INVALID, "ekf_associate: bad argument");
    Pool& P = h->pool;
    EKFC(P.use(false));
    const int n = P.pv.n;
    int known_count = 0;  // ekf_slam.cpp:281-288: leading run of true
    for (int i = 0; i < n; i++) {
        if (known[i]) known_count++; else break;
    }
    if (J == 0) return EKF_OK;
    const bool delayed = P.pend_cap > 0;   // delayed mode: scores and gains against Sigma_base minus the pending pairs
    EKFC(P.ensure_meas_capacity(J));
    P.pub_sent = false;
    if (!delayed && P.small_path && P.pv.N <= ekf::small_max_dim() && n > 0 && J <= ekf::kSmallInlineJ) {
        // small map: the whole call (and the prediction() before it) in one LDS-resident launch, measurements by value
        ekf::SmallInlineMeas in;
        std::memcpy(in.xy, meas_xy, sizeof(double) * 2 * J);
        ekf::launch_small_associate_inline(P.pv, in, J, known_count, P.assoc_out_dev, P.pred_pending, P.pred_dth, P.pred_dx,
                                           P.stream);
        P.pred_pending = false;
        EKFC(checked_launch());
        return associate_finish(P, known_count, J, known, assoc_out);
    }
    // The call-fused forms take their readings BY VALUE in the kernel arguments; every other path reads them from the device:
    // staged on first need (one host-to-device copy per call, ~5 us).
    bool staged = false;
    auto need_dev = [&]() -> ekf_status {
        if (!staged) EKFC(P.upload(P.meas_dev, meas_xy, sizeof(double) * 2 * J));
        staged = true;
        return EKF_OK;
    };
    if (!delayed && P.small_path && P.pv.N <= ekf::small_max_dim() && n > 0 && n <= 128) {
        // small map: scores, decisions and corrections of all J measurements in one LDS-resident launch
        EKFC(need_dev());
        ekf::launch_small_associate(P.pv, P.meas_dev, J, known_count, P.assoc_out_dev, P.pred_pending, P.pred_dth,
                                    P.pred_dx, P.stream);
        P.pred_pending = false;
        EKFC(checked_launch());
        return associate_finish(P, known_count, J, known, assoc_out);
    }
    EKFC(P.use());  // settles a deferred prediction()
    // known_count lives on the device between calls (the node only ever passes back the list this function
    // returned); it is rewritten only when the caller's list says something else
    if (known_count != P.dev_known_count) ekf::launch_assoc_begin(P.pv, nullptr, known_count, P.stream);
    {   // A big map whose DISCOVERED part is still small: the whole call as one LDS-resident launch on the prefix
        // (everything beyond 3 + 2*(known_count + J), and beyond what this object ever corrected, still holds
        // constructor values).
        int m = known_count + J < n ? known_count + J : n;
        if (P.touched_hwm > m) m = P.touched_hwm;
        const int Nb = 3 + 2 * m;
        if (!delayed && P.small_path && P.active_prefix && n > 0 && Nb <= ekf::small_max_dim()) {
            EKFC(need_dev());
            ekf::PoolView pva = P.pv;
            pva.N = Nb;
            ekf::launch_pool_associate(pva, P.meas_dev, nullptr, J, 3 + 2 * P.touched_hwm, P.assoc_out_dev, nullptr, P.stream);
            EKFC(checked_launch());
            return associate_finish(P, known_count, J, known, assoc_out);
        }
    }
    ekf::CmdSrc src{};
    src.mode = ekf::SRC_ASSOC;
    src.assoc = P.pv.assoc;
    src.fresh_pose = 1;
    // Landmarks are appended in discovery order (:318-327), so rows/columns beyond 3 + 2*(known_count + j + 1)
    // -- and beyond every landmark this object has ever corrected (touched_hwm: known_list is caller-owned
    // and may have holes) -- still hold their constructor values, and every correction of this call is
    // exactly confined to that leading block: K and H*Sigma are exact zeros outside it.  (Non-finite
    // states void this; they are already garbage in the reference.)
    auto active_dim = [&](int j) {   // leading block that reading j of this call can touch (exact, see above)
        if (!P.active_prefix) return P.pv.N;
        int m = known_count + j + 1 < n ? known_count + j + 1 : n;
        if (P.touched_hwm > m) m = P.touched_hwm;
        return 3 + 2 * m;
    };
    // The discovered map fits one workgroup (known_count + J and every landmark ever corrected <= 448): the whole call is
    // ONE launch -- a thread per landmark keeps its 5 x 5 block of Sigma current in registers from the first reading to the
    // last (ekf_assocfused.hip: k_assoc_call) -- plus one pass over the prefix.  configs[2]'s discovery run: two launches
    // per call where the per-reading forms take two per reading.
    {
        int carried = known_count + J < n ? known_count + J : n;
        if (P.touched_hwm > carried) carried = P.touched_hwm;
        if (!delayed && P.pv.B == 1 && P.call_fused_ok() && carried <= ekf::assoc_call_capacity()) {
            EKFC(P.ensure_callfused());
            for (int j0 = 0; j0 < J; j0 += ekf::kCallV) {
                const int jc = J - j0 < ekf::kCallV ? J - j0 : ekf::kCallV;
                ekf::AssocCallArgs ca{};
                for (int jj = 0; jj < jc; jj++) { ca.xy[jj][0] = meas_xy[2 * (j0 + jj)]; ca.xy[jj][1] = meas_xy[2 * (j0 + jj) + 1]; }
                ca.J = jc;
                ca.known_count = known_count + j0 < n ? known_count + j0 : n;
                ca.touched_hwm = P.touched_hwm;
                ca.active_prefix = P.active_prefix;
                EKFC(P.prof_begin(1));
                // (the launch writes its decisions to the host's mapped copy; the call's last launch also the record and the
                // sequence number, behind its last decision: the host waits neither for the last gain nor for the pass)
                const bool last = j0 + jc >= J;
                ekf::launch_assoc_call(P.pv, ca, carried, P.assoc_out_dev + j0, P.cf_U, P.cf_V, P.cf_cnt, ekf::rank2v_round_count(jc),
                                       P.stream, P.phase_trace, P.pub_host, j0, last && P.pub_host ? P.claim_pub_seq() : 0u);
                EKFC(P.prof_end());
                ekf::PoolView view = P.pv;
                view.N = active_dim(j0 + jc - 1);
                EKFC(P.prof_begin(0));
                ekf::launch_rank2v(view, P.cf_U, P.cf_V, P.cf_cnt, jc, P.tuning, P.stream);
                EKFC(P.prof_end());
            }
            EKFC(checked_launch());
            return associate_finish(P, known_count, J, known, assoc_out);
        }
    }
    // Discovered maps beyond one workgroup: ONE launch per reading -- decision + gain against the stored covariance minus the
    // call's pending pairs, and the next reading's scores -- and ONE pass over Sigma per call (per 8 readings);
    // ekf_assocfused.hip, bit-identical.  (Round 3 took this form from an active dimension of 1400 on and an out-of-place
    // correction launch per reading below it; measured in round 4, tools/forms_ab.py: 89-92 us per 8-reading call at
    // n = 460 ... 690 against 122-138 us -- the per-reading form wins wherever the one-workgroup form ends, and the
    // out-of-place form with its second N x N buffer is gone.)
    if (!delayed && P.pv.B == 1 && P.call_fused_ok()) {
        EKFC(P.ensure_callfused());
        if (!P.terms) EKFC(P.dalloc(&P.terms, (size_t)(n > 0 ? n : 1) * 16));
        if (!P.terms2) {
            EKFC(P.dalloc(&P.terms2, (size_t)(n > 0 ? n : 1) * 16));
            EKFC(P.dalloc(&P.scores2, (size_t)(n > 0 ? n : 1)));
        }
        EKFC(P.ensure_blk_cache());   // every landmark's 5 x 5 block, kept current from reading to reading
        auto m_bound = [&](int j) { return known_count + j < n ? known_count + j : n; };   // known count in front of reading j
        for (int j0 = 0; j0 < J; j0 += ekf::kCallV) {
            const int jc = J - j0 < ekf::kCallV ? J - j0 : ekf::kCallV;
            // the first reading of a pass is scored by a launch of its own; every other one by its predecessor's launch
            double *sc_in = P.scores, *tm_in = P.terms, *sc_out = P.scores2, *tm_out = P.terms2;
            EKFC(P.prof_begin(1));
            ekf::launch_assoc_score(P.pv, meas_xy[2 * j0], meas_xy[2 * j0 + 1], P.pv.assoc, P.cf_U, P.cf_V, 0, m_bound(j0), sc_in,
                                    tm_in, P.stream, P.blk_cache);
            EKFC(P.prof_end());
            for (int jj = 0; jj < jc; jj++) {
                const int j = j0 + jj;
                const int has_next = jj + 1 < jc;
                EKFC(P.prof_begin(1));
                ekf::launch_assoc_reading(P.pv, meas_xy[2 * j], meas_xy[2 * j + 1], has_next, has_next ? meas_xy[2 * j + 2] : 0.0,
                                          has_next ? meas_xy[2 * j + 3] : 0.0, P.pv.assoc, P.assoc_alt, P.assoc_out_dev + j,
                                          P.cf_state, P.cf_U, P.cf_V, P.cf_cnt, jj, active_dim(j),
                                          jj == jc - 1 ? ekf::rank2v_round_count(jc) : 0, m_bound(j), sc_in, tm_in, sc_out, tm_out,
                                          P.stream, P.phase_trace, P.blk_cache, P.pub_host, j,
                                          j == J - 1 && P.pub_host ? P.claim_pub_seq() : 0u);
                EKFC(P.prof_end());
                std::swap(P.pv.state, P.cf_state);
                std::swap(P.pv.assoc, P.assoc_alt);
                std::swap(sc_in, sc_out);
                std::swap(tm_in, tm_out);
            }
            ekf::PoolView view = P.pv;
            view.N = active_dim(j0 + jc - 1);
            EKFC(P.prof_begin(0));
            ekf::launch_rank2v(view, P.cf_U, P.cf_V, P.cf_cnt, jc, P.tuning, P.stream);
            EKFC(P.prof_end());
        }
        EKFC(checked_launch());
        return associate_finish(P, known_count, J, known, assoc_out);
    }
    for (int j = 0; j < J; j++) {  // ekf_slam.cpp:291: sequential, state-carrying
        const double* mj = P.meas_dev + 2 * (size_t)j;
        int active_N = 0;
        if (P.active_prefix) {
            int m = known_count + j + 1 < n ? known_count + j + 1 : n;
            if (P.touched_hwm > m) m = P.touched_hwm;
            active_N = 3 + 2 * m;
        }
        EKFC(need_dev());
        const ekf::MeasSrc ms{mj, 2, nullptr, 0};
        EKFC(P.prof_begin(1));
        const ekf::Pending pend = P.pending();
        ekf::launch_maha(P.pv, ms, P.scores, -1, known_count + j < n ? known_count + j : n, P.stream,
                         delayed ? &pend : nullptr);  // :300-309
        EKFC(P.prof_end());
        ekf::launch_assoc_decide(P.pv, ms, P.scores, P.assoc_out_dev, 0, j, nullptr, P.stream);    // :293-330
        src.meas = mj;
        src.meas_stride = 2;
        EKFC(P.correct(src, active_N));                                                   // :331-390
    }
    EKFC(checked_launch());
    return associate_finish(P, known_count, J, known, assoc_out);
}

ekf_status ekf_maha_scores(ekf_handle h, double meas_x, double meas_y, int M, double* scores_out) {
    if (!h || !scores_out || M < 0) return fail(EKF_ERR_INVALID, "ekf_maha_scores: bad argument");
    Pool& P = h->pool;
    if (M > P.pv.n) return fail(EKF_ERR_INVALID, "ekf_maha_scores: M exceeds the number of landmarks");
    if (M == 0) return EKF_OK;
    EKFC(P.use());
    EKFC(P.flush());   // (the parity hook reads the materialised covariance)
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

ekf_status ekf_set_forms(ekf_handle h, unsigned forms) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.set_forms(forms);
}

ekf_status ekf_get_forms(ekf_handle h, unsigned* forms) {
    if (!h || !forms) return fail(EKF_ERR_INVALID, "null argument");
    *forms = h->pool.forms;
    return EKF_OK;
}

ekf_status ekf_phase_trace(ekf_handle h, int enable, long long* out) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = h->pool;
    EKFC(P.use());
    const size_t cnt = (size_t)2 * ekf::kTraceSlots;
    if (out && P.phase_trace) EKFC(P.download(out, P.phase_trace, sizeof(long long) * cnt));
    if (enable && !P.phase_trace) {
        EKFC(P.dalloc(&P.phase_trace, cnt));
        HIPC(hipMemsetAsync(P.phase_trace, 0, sizeof(long long) * cnt, P.stream));
    } else if (!enable && P.phase_trace) {
        HIPC(hipStreamSynchronize(P.stream));
        HIPC(hipFree(P.phase_trace));
        P.phase_trace = nullptr;
    }
    return EKF_OK;
}

ekf_status ekf_test_raise_device_error(ekf_handle h) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = h->pool;
    EKFC(P.use());
    ekf::launch_raise_device_error(P.pv, P.stream);
    EKFC(checked_launch());
    HIPC(hipStreamSynchronize(P.stream));   // (returns EKF_OK: the word is looked at by the NEXT entry)
    return EKF_OK;
}

ekf_status ekf_set_profiling(ekf_handle h, int enable) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = h->pool;
    EKFC(P.use());
    EKFC(P.prof_drain());
    P.prof_on = enable ? 1 : 0;
    P.prof_ms[0] = P.prof_ms[1] = 0.0;
    P.prof_launches[0] = P.prof_launches[1] = 0;
    return EKF_OK;
}

ekf_status ekf_get_profile(ekf_handle h, double ms[2], long long launches[2]) {
    if (!h || !ms || !launches) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = h->pool;
    EKFC(P.use());
    EKFC(P.prof_drain());
    for (int c = 0; c < 2; c++) { ms[c] = P.prof_ms[c]; launches[c] = P.prof_launches[c]; }
    return EKF_OK;
}

ekf_status ekf_sync(ekf_handle h) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.sync();
}

ekf_status ekf_set_tuning(ekf_handle h, int rows_per_block, int nontemporal, int group_rows) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    h->pool.set_tuning(rows_per_block, nontemporal, group_rows);
    return EKF_OK;
}

// ---- batch ---------------------------------------------------------------------------------

ekf_status ekf_set_update_mode(ekf_handle h, int max_pending_corrections, int symmetric_gather) {
    if (!h) return fail(EKF_ERR_INVALID, "null handle");
    return h->pool.set_update_mode(max_pending_corrections, symmetric_gather);
}

}  // extern "C"
