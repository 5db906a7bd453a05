/*
 * ekfslam.h -- C ABI of libekfslam_hip.so, the MI355X (gfx950) EKF-SLAM filter core.
 *
 * Drop-in boundary for the hot path of tonylitianyu/EKF-SLAM-ML: the public class
 * rigid2d::EKF_SLAM (rigid2d/include/rigid2d/ekf_slam.hpp:19-57, implemented in
 * rigid2d/src/ekf_slam.cpp).  The reference has no FFI layer; the binding a maintainer
 * adds is a replacement for src/ekf_slam.cpp that forwards each method to one entry
 * point below (INTEGRATION.md shows it; host/ekf_slam.hpp is the header-only C++ mirror).
 *
 * Conventions
 *   - plain pointers and sizes only; caller-owned HOST buffers unless a name says _dev;
 *   - every function returns an ekf_status (0 = EKF_OK); no exceptions cross the boundary;
 *     ekf_last_error() gives the text of the calling thread's last failure;
 *   - one handle = one filter (or one batch of filters) bound to one HIP device and one HIP
 *     stream; a handle is NOT thread-safe (the reference is driven from ROS1's single-threaded
 *     spinner, nuslam/src/slam.cpp:525);
 *   - state order [theta, x, y, m1x, m1y, ...] (ekf_slam.cpp:15-21,72-74); N = 3 + 2n;
 *   - covariance crosses the boundary ROW-major N x N fp64; std::vector<bool> arguments of the
 *     reference cross as uint8_t[n] (0 / non-zero).
 *   - all arithmetic is fp64 like the reference (arma::mat = Mat<double>).
 */
#ifndef EKFSLAM_H
#define EKFSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    EKF_OK = 0,
    EKF_ERR_INVALID = 1,   /* bad argument (null pointer, n < 0, index out of range ...) */
    EKF_ERR_NO_DEVICE = 2, /* no HIP device / wrong architecture */
    EKF_ERR_HIP = 3,       /* a HIP runtime call failed, see ekf_last_error() */
    EKF_ERR_NOMEM = 4,     /* host or device allocation failed */
    EKF_ERR_STATE = 5      /* call sequence error (e.g. run before a log was uploaded) */
} ekf_status;

/* The reference hard-codes these; defaults reproduce it exactly. */
typedef struct {
    double sigma0_landmark; /* 100     ekf_slam.cpp:32      initial landmark variance       */
    double q_pose;          /* 1e-4    ekf_slam.cpp:40-43   process noise on theta, x, y    */
    double r_meas;          /* 0.01    ekf_slam.cpp:172-175 range / bearing noise           */
    double gate_new;        /* 10.0    ekf_slam.cpp:293     Mahalanobis gate: new landmark  */
    double gate_update;     /* 1.0     ekf_slam.cpp:330     Mahalanobis gate: apply update  */
    double straight_eps;    /* 1e-6    ekf_slam.cpp:79      |dtheta| below -> straight line */
} ekf_params;

typedef struct ekf_filter_s* ekf_handle;      /* one filter  == one rigid2d::EKF_SLAM object */
typedef struct ekf_batch_s* ekf_batch_handle; /* B independent filters (Monte-Carlo batch)   */

const char* ekf_last_error(void);
void ekf_default_params(ekf_params* out);
/* Doubles between the starts of two rows of a filter's covariance (and the length of its state / factor vectors) for a map of
 * n landmarks: N = 3 + 2 n rounded up to 16, or to 256 -- rows on 2-KB boundaries -- where that adds at most 1/32 of a row
 * (n = 1000: 2048).  A filter takes 8 N ld bytes of device memory for its covariance.  Needs no device. */
int ekf_leading_dimension(int n_landmarks);
/* Number of visible HIP devices (0 if none); does not initialise a device context. */
int ekf_device_count(void);

/* ---- single filter: the rigid2d::EKF_SLAM call surface ------------------------------------ */

/* EKF_SLAM::EKF_SLAM(int n_measurements)            ekf_slam.hpp:27, ekf_slam.cpp:27-53.
 * params may be NULL (reference constants).  device < 0 selects the current HIP device. */
ekf_status ekf_create(int n, const ekf_params* params, int device, ekf_handle* out);
ekf_status ekf_destroy(ekf_handle h);
/* Copy construction / copy assignment of the by-value member (nuslam/src/slam.cpp:213,428). */
ekf_status ekf_clone(ekf_handle h, ekf_handle* out);

/* void prediction(const Twist2D&)                   ekf_slam.hpp:31, ekf_slam.cpp:55-106.
 * dtheta = twist.angular(), dx = twist.linearX(); linearY is ignored by the reference (:70). */
ekf_status ekf_predict(ekf_handle h, double dtheta, double dx);

/* void measurement(mat sensor_reading, vector<bool> visible_list, vector<bool> known_list)
 *                                                   ekf_slam.hpp:36, ekf_slam.cpp:108-197.
 * sensor_xy: 2n doubles (robot-frame x,y per landmark = sensor_reading.memptr());
 * visible: n bytes.  known_list is unused by the reference and therefore not passed. */
ekf_status ekf_measure_known(ekf_handle h, const double* sensor_xy, const uint8_t* visible);

/* void data_association(vector<Vector2D> measures, vector<bool>& known_list)
 *                                                   ekf_slam.hpp:41, ekf_slam.cpp:278-402.
 * meas_xy: 2J doubles (Vector2D is {double x, y}, rigid2d.hpp:68-72, so &measures[0].x);
 * known: n bytes, IN/OUT (entries are set as landmarks are initialised, :323);
 * assoc_out: optional J ints, the landmark each measurement updated (-1 = dropped).
 * Synchronous in its RESULTS only: the call returns when the decisions and known_list are final; the gain of the last
 * reading and the call's pass over the covariance may still be running on the handle's stream.  An execution error of
 * that tail (or a device-side error word, EKF_ERR_HIP) therefore surfaces at the NEXT call that synchronises with the
 * stream (ekf_sync, any getter, the next ekf_associate); stream order keeps every later call behind it.
 * Which launch structure ("form") runs is chosen by map size; state, covariance, decisions and known_list are
 * bit-identical across forms.  Internal bookkeeping is not: the diagnostic winning distance of the association record
 * is only kept by the per-reading forms, and the touched set (ekf_set_active_set) is "every landmark below the known
 * count" on the call-fused forms and "landmarks actually corrected" on the others -- both supersets of what the
 * exact sparsity needs. */
ekf_status ekf_associate(ekf_handle h, const double* meas_xy, int J, uint8_t* known, int* assoc_out);

/* double calculate_maha_dis(Vector2D, int) for i in [0, M)   ekf_slam.hpp:86, ekf_slam.cpp:217-276.
 * Private in the reference; exposed as the parity hook of the one-landmark-per-wavefront
 * scoring kernel.  scores_out: M doubles. */
ekf_status ekf_maha_scores(ekf_handle h, double meas_x, double meas_y, int M, double* scores_out);

/* double rigid2d::normalize_angle(double)            rigid2d/src/rigid2d.cpp:336-345 (range (-pi, pi]).
 * Parity hook of the device helper every kernel uses: out[i] = normalize_angle(in[i]), count values. */
ekf_status ekf_normalize_angles(int device, const double* in, int count, double* out);

/* getStateTheta / getStateX / getStateY             ekf_slam.cpp:404-414 -> out = {theta, x, y} */
ekf_status ekf_get_pose(ekf_handle h, double out[3]);
/* mat getStateLandmark()                            ekf_slam.cpp:416-418 -> 2n doubles */
ekf_status ekf_get_landmarks(ekf_handle h, double* out);

/* snapshot / restore and test access to the private members state, sigma (ekf_slam.hpp:61-65) */
ekf_status ekf_dim(ekf_handle h, int* n, int* N);
ekf_status ekf_get_state(ekf_handle h, double* out /* N */);
ekf_status ekf_set_state(ekf_handle h, const double* in /* N */);
ekf_status ekf_get_cov(ekf_handle h, double* out /* N*N row-major */);
ekf_status ekf_set_cov(ekf_handle h, const double* in /* N*N row-major */);
ekf_status ekf_get_init_flag(ekf_handle h, int* flag);  /* landmark_init_flag, ekf_slam.hpp:65 */
ekf_status ekf_set_init_flag(ekf_handle h, int flag);
/* ---- execution forms (test / measurement hook; a caller never needs it) -----------------------------------------
 * Every entry point computes ONE function -- the reference's -- but the library holds several launch structures
 * ("forms") for it and picks among them by map size and pool size (DESIGN.md section 4, table "path selection").
 * All exact forms are bit-identical to each other; the tests switch them off one by one and compare bit for bit.
 * A bit set = the form may be taken where it applies (default: EKF_FORMS_DEFAULT = all but the ALWAYS hook). */
typedef enum {
    /* N = 3 + 2n <= 104 (the reference's own n = 20): a whole measurement() / data_association() call, a whole step of
     * a pool (while 3 + 2*(known_count + readings) <= 104) or a whole run of a known-association log is ONE launch with
     * Sigma resident in LDS (ekf_small.hip).  Off: the multi-kernel chain. */
    EKF_FORM_SMALL_MAP = 1u << 0,
    /* (1u << 1: retired in round 4.  It was EKF_FORM_FUSED_CORRECTION -- gain + state + covariance of a correction in one
     * launch that wrote Sigma - K(H Sigma) out of place into a second N x N buffer; tools/forms_ab.py measured it 2.5-4.7x
     * behind the call-fused forms on measurement() and 1.3-1.5x behind them on data_association() at every map size, so
     * the kernels and the second buffer were deleted.  The bit is ignored.) */
    /* beyond the small-map path: measurement() as TWO launches per call whatever the number of visible landmarks --
     * the gains K_v and rows H_v Sigma of all the call's corrections from two thin panels of Sigma, then ONE
     * read-modify-write pass in which every element takes its V rank-2 corrections in order: 16 N^2 bytes per CALL
     * (per 8 corrections) instead of per landmark (ekf_callfused.hip); a single filter's data_association() as one launch
     * per reading + one pass per call (ekf_assocfused.hip).  Off: one covariance stream per landmark -- the eager
     * contract stream bench.py quotes `value` / `roofline` on. */
    EKF_FORM_CALL_FUSED = 1u << 2,
    /* data_association() appends landmarks in discovery order, so its corrections are exactly confined to the leading
     * 3 + 2*known_count block of the state: stream only that block.  Off: full width. */
    EKF_FORM_ACTIVE_PREFIX = 1u << 3,
    /* pools, ekf_batch_run_unknown beyond the LDS-resident path, <= 8 reading slots per step: one launch per STEP (a
     * workgroup per filter scores, decides and builds the gains against the stored covariance minus the step's pending
     * pairs; ekf_stepfused.hip).  Off: four launches per reading slot. */
    EKF_FORM_STEP_FUSED = 1u << 4,
    /* ... and when the discovered prefixes are big, the step's covariance pass as a second launch spread over the chip. */
    EKF_FORM_STEP_SPLIT_PASS = 1u << 5,
    /* delayed mode, ekf_batch_run_known: the two corrections of a step in ONE gain launch (pending factors read once). */
    EKF_FORM_DELAYED_PAIR = 1u << 6,
    /* narrow maps: P consecutive rows streamed as one virtual row that fills the 256-lane strips (k_rank2_packed). */
    EKF_FORM_ROW_PACKING = 1u << 7,
    /* delayed mode: the strip-form flush (V strip in LDS) beyond 40 pending vectors on pools that fill the chip. */
    EKF_FORM_STRIP_FLUSH = 1u << 8,
    /* test hook: the strip-form flush on pools of any size, for any count <= 80 vectors. */
    EKF_FORM_STRIP_FLUSH_ALWAYS = 1u << 9,
    /* delayed mode, ekf_batch_run_known: the flush also writes the COLUMNS of Sigma that the next corrections' K = Sigma H^T
     * S^-1 (ekf_slam.cpp:178) will read -- columns 0..2 and the two columns of every landmark in the next corrections of
     * the device-resident log -- as contiguous rows of a column panel, the gain kernels read them coalesced instead of as
     * 16-KB-strided sectors, and prediction() keeps columns 0..2 current in the panel (coalesced) instead of in the matrix.
     * Same values from another address: bit-identical.  Off: columns are gathered from the matrix. */
    EKF_FORM_COLUMN_PANEL = 1u << 10,
    /* test hook: the panel plans ONE landmark per flush period, so panel rows and matrix gathers are mixed in one launch. */
    EKF_FORM_COLUMN_PANEL_ONE_SLOT = 1u << 11,
    /* pools' delayed data_association() (ekf_batch_run_unknown in delayed mode): a launch in front of every step guesses each
     * reading's winner (the landmark nearest to where the reading lands) and rebuilds "stored covariance minus the pairs of
     * earlier steps" for the guessed landmarks' rows / columns ONCE per step; the step kernel continues from it where its own
     * decision agrees and rebuilds from scratch where it does not -- the pending store is read once per step instead of once
     * per reading.  Same operations in the same order: bit-identical.  Off: every reading rebuilds. */
    EKF_FORM_STEP_SPECULATE = 1u << 12,
    /* with the column panel (delayed known-association runs, paired gain launches): the rows and columns of Sigma at the pose
     * indices and at every planned landmark are KEPT as they stand after each launch that rebuilt them, and the next launch
     * that meets them folds only the pending vectors appended since -- a robot corrects the same few landmarks step after
     * step, so a gain launch reads 4 factor vectors instead of all pending ones.  prediction() maps the kept vectors like
     * the rest of Sigma.  The same quantities in another association of the sums: equal to the run without it to rounding
     * (1e-10 in the tests, where the delayed mode itself sits 1e-10 from the eager run), not bit for bit.  Off: every launch rebuilds from the stored entries. */
    EKF_FORM_CURRENT_COLUMNS = 1u << 13,
    /* pools with many more rank-2 tiles than CUs stream the eager correction as RESIDENT workgroups that take their tiles from
     * one queue (an atomicAdd per tile) instead of a grid of short-lived ones: the dispatcher deals a fixed eighth of a grid to
     * each XCD, and XCDs / CUs do not stream at the same rate (tools/micro/strip_walk.hip).  Speed only: same arithmetic. */
    EKF_FORM_TILE_QUEUE = 1u << 14,
    EKF_FORMS_DEFAULT = ((1u << 9) - 1) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14)
} ekf_form;
ekf_status ekf_set_forms(ekf_handle h, unsigned forms);
ekf_status ekf_get_forms(ekf_handle h, unsigned* forms);
ekf_status ekf_batch_set_forms(ekf_batch_handle hb, unsigned forms);
ekf_status ekf_batch_get_forms(ekf_batch_handle hb, unsigned* forms);
/* Test hook: covariance passes of each form this pool has launched since it was created --
 * counts[0] plain flush, [1] strip-form flush, [2] paired delayed gain launches, [3] call-fused passes,
 * [4] per-landmark rank-2 streams, [5] step-fused launches with a separate pass, [6] mirrored flushes (symmetric
 * option of ekf_set_update_mode), [7] delayed gain launches that read the column panel (EKF_FORM_COLUMN_PANEL). */
ekf_status ekf_batch_form_counts(ekf_batch_handle hb, long long counts[8]);

/* Active-set covariance update (opt-in, default 0; reported separately from the dense contract path):
 * the eager correction streams only the rows of the TOUCHED set -- the pose rows and the rows of landmarks
 * that have ever been corrected.  Every other row has K(r,:) = 0 exactly (its landmark still carries the
 * constructor covariance and is decoupled from everything), so the result is bit-identical for finite
 * states while the traffic drops from 2*8*N^2 to 2*8*N*(3 + 2*touched) bytes per correction. */
ekf_status ekf_set_active_set(ekf_handle h, int enable);
ekf_status ekf_batch_set_active_set(ekf_batch_handle hb, int enable);
/* Size of every filter's touched set (landmarks corrected at least once): counts_out[B]. */
ekf_status ekf_batch_get_touched(ekf_batch_handle hb, int* counts_out);
/* Diagnostics of the two-launch measurement() call: while enabled, lane 0 of the control wave (row 0) and of the first
 * slice wave (row 1) of workgroup 0 of k_call_factors stamp the 100 MHz wall clock at their phase boundaries.  out (nullable)
 * receives the stamps of the LAST call, [2][64] (slots: 0 start, 1 gathers issued, 2 prediction folded, then per
 * correction t at 3 + 5t: barrier passed, terms / panel update done, second barrier passed, gains done; 60 loop done). */
ekf_status ekf_phase_trace(ekf_handle h, int enable, long long* out);
/* Test hook of the device error word: a one-thread kernel raises it the way a kernel that gave up would (a bounded
 * in-kernel hand-off that never arrived).  The call itself returns EKF_OK; every LATER entry point of the handle must
 * fail with EKF_ERR_HIP (the word is sticky: results behind it are not to be used). */
ekf_status ekf_test_raise_device_error(ekf_handle h);
/* Blocks until every kernel queued on the handle's stream has finished. */
ekf_status ekf_sync(ekf_handle h);
/* Measurement hook (off by default): brackets every covariance-streaming launch (class 0: fused correction, rank-2
 * stream, decision + correction) and every Mahalanobis scoring launch (class 1) of this filter with HIP events on
 * its stream.  ekf_get_profile synchronises and returns the summed kernel milliseconds and launch counts per class
 * since profiling was switched on. */
ekf_status ekf_set_profiling(ekf_handle h, int enable);
ekf_status ekf_get_profile(ekf_handle h, double ms[2], long long launches[2]);

/* ---- batch of independent filters (BASELINE.json configs[4]) -------
 * Each filter is exactly one EKF_SLAM object; they share nothing.  Inputs are a compact
 * known-association log (visible readings only, ascending landmark index -- the order of the
 * loop at ekf_slam.cpp:132), uploaded once so the timed region starts with inputs in HBM. */

typedef struct {
    int T;               /* steps; step t = prediction(twist[t]) + measurement(readings[t])   */
    int vmax;            /* reading slots per step                                             */
    const double* twist; /* [T][B][2]  (dtheta, dx)                                            */
    const int* lm_idx;   /* [T][B][vmax] landmark index per slot, ascending, -1 ends the list  */
    const double* z_xy;  /* [T][B][vmax][2] robot-frame (x, y) readings                        */
    const double* init_xy; /* [B][2n] sensor_reading of the FIRST measurement() call (:113-128) */
} ekf_known_log;

typedef struct {
    double elapsed_ms;        /* HIP-event time of the whole run on the batch's stream         */
    double rank2_ms;          /* sum of the durations of the covariance passes (time_kernels): */
                              /* eager: the rank-2 kernel; delayed mode: the flush kernel      */
    long long rank2_launches; /* covariance passes in the run (eager: one per correction slot) */
    long long corrections;    /* landmark corrections applied over all filters                 */
    long long filter_steps;   /* (prediction + measurement) pairs over all filters             */
    double rank2_bytes_per_launch; /* algorithmic bytes of one pass: (filters touched) * 2*8*N^2 */
} ekf_run_stats;

ekf_status ekf_batch_create(int B, int n, const ekf_params* params, int device, ekf_batch_handle* out);
ekf_status ekf_batch_destroy(ekf_batch_handle hb);
/* Re-initialise every filter to the constructor state (ekf_slam.cpp:27-53). */
ekf_status ekf_batch_reset(ekf_batch_handle hb);
/* Device bytes the batch holds (covariance pool + state + scratch + uploaded log). */
ekf_status ekf_batch_device_bytes(ekf_batch_handle hb, size_t* bytes);
ekf_status ekf_batch_upload_known_log(ekf_batch_handle hb, const ekf_known_log* log);
/* Runs steps [t_begin, t_end) of the uploaded log for all filters on the batch's stream and
 * waits for completion.  time_kernels != 0 brackets every rank-2 launch with HIP events. */
ekf_status ekf_batch_run_known(ekf_batch_handle hb, int t_begin, int t_end, int time_kernels,
                               ekf_run_stats* stats);

/* Unknown-association log: per step one twist and up to jmax robot-frame circle centres per filter --
 * the node loop of nuslam/src/unknown_data_assoc.cpp:300-323 (prediction(twist), then
 * data_association(measures, known_list)) for B independent robots. */
typedef struct ekf_unknown_log {
    int T;                 /* steps */
    int jmax;              /* measurement slots per (step, filter) */
    const double* twist;   /* [T][B][2]  (angular, linearX) */
    const int* count;      /* [T][B]     measurements of filter b in step t, 0..jmax */
    const double* meas_xy; /* [T][B][jmax][2] robot-frame (x, y); slots >= count are ignored */
} ekf_unknown_log;
ekf_status ekf_batch_upload_unknown_log(ekf_batch_handle hb, const ekf_unknown_log* log);
/* Runs steps [t_begin, t_end) of the uploaded unknown-association log: per step prediction()
 * (ekf_slam.cpp:55-106), then per measurement slot the Mahalanobis scores (:217-276), the
 * gate decision / landmark initialisation (:293-330) and the correction (:331-390) of every filter
 * that has a measurement in the slot.  Each filter's known_count (the leading run of its
 * known_list, :281-288) lives on the device and carries over between runs until ekf_batch_reset.
 * In delayed mode (ekf_batch_set_update_mode(k > 0)) with at most 8 reading slots per step, the pairs of a step stay
 * pending across steps -- every reading is scored and corrected against the stored covariance minus ALL pending pairs,
 * Sigma is rewritten once per floor(k / jmax) steps -- with the mode's tolerance (1e-9; decisions identical in every
 * test); otherwise pending delayed corrections are flushed first and the run is eager.  stats->corrections counts the
 * corrections actually applied (decided on the device). */
ekf_status ekf_batch_run_unknown(ekf_batch_handle hb, int t_begin, int t_end, int time_kernels,
                                 ekf_run_stats* stats);
/* The batch twin of the caller-owned known_list argument of data_association(): sets every filter's known_count
 * (the leading run of its known_list, ekf_slam.cpp:281-288) to counts[b] in 0..n -- e.g. to continue with unknown
 * association on a map built through the known-association path, whose first call initialises all n landmarks
 * (:113-128).  Landmarks below the count are treated as possibly corrected (no discovered-prefix shortcut). */
ekf_status ekf_batch_set_known_counts(ekf_batch_handle hb, const int* counts);
/* known_count of every filter: out[B]. */
ekf_status ekf_batch_get_known_counts(ekf_batch_handle hb, int* out);
/* Decisions of the uploaded unknown log's steps run so far: out = [T][B][jmax], landmark index
 * corrected, -1 = measurement dropped (:330), -2 = no measurement in the slot / step not run. */
ekf_status ekf_batch_get_decisions(ekf_batch_handle hb, int* out);

ekf_status ekf_batch_get_state(ekf_batch_handle hb, int b, double* out /* N */);
ekf_status ekf_batch_get_cov(ekf_batch_handle hb, int b, double* out /* N*N row-major */);
/* All poses at once: out = [B][3] (theta, x, y) -- the Monte-Carlo read-back. */
ekf_status ekf_batch_get_poses(ekf_batch_handle hb, double* out);
/* order-independent digest of every filter's state and covariance (device-side reduction):
 * out[0] = sum state, out[1] = sum |state|, out[2] = sum sigma, out[3] = sum |sigma| */
ekf_status ekf_batch_checksum(ekf_batch_handle hb, double out[4]);

/* Tuning knobs of the covariance rank-2 kernel: rows per workgroup, non-temporal access (0/1),
 * rows per load/store group (2, 4, 8 or 16).  0 (nontemporal: < 0) restores the automatic choice.
 * Results do not depend on them, bit for bit. */
ekf_status ekf_batch_set_tuning(ekf_batch_handle hb, int rows_per_block, int nontemporal, int group_rows);
/* Report hook: the instantiation ekf::k_rank2<group_rows, nontemporal, threads> and the rows per workgroup that a
 * full-width eager correction of this pool launches (bench.py ties its PMC traffic record to this name). */
ekf_status ekf_batch_rank2_variant(ekf_batch_handle hb, int* group_rows, int* nontemporal, int* threads,
                                   int* rows_per_block);
/* ... and whether that launch runs as resident workgroups on one tile queue (ekf::k_rank2_queue<...>, EKF_FORM_TILE_QUEUE). */
ekf_status ekf_batch_rank2_resident(ekf_batch_handle hb, int* resident);
ekf_status ekf_set_tuning(ekf_handle h, int rows_per_block, int nontemporal, int group_rows);

/* ---- on-device Monte-Carlo inputs and consistency statistics --------
 * Generates the known-association log of every filter ON THE DEVICE from the noise model of the
 * reference's simulator (nurtlesim/src/tube_world.cpp:191-227,369-414; constants
 * nurtlesim/config/noise_param.yaml:2-11) and the caller's odometry marshalling
 * (nuslam/src/slam.cpp:173-176), instead of uploading it.  Every random number is a pure function of
 * (seed, first_filter_id + b, step, kind, k), so shards of one job are reproducible independently. */
typedef struct {
    unsigned long long seed;
    long long first_filter_id;   /* global id of filter 0 of this batch (multi-GPU sharding)          */
    double v_cmd, w_cmd;         /* commanded twist (m/s, rad/s)                                       */
    double vx_std, the_std;      /* noise_param.yaml:3,5   noise on the commanded twist                */
    double slip_min, slip_max;   /* noise_param.yaml:6-7   wheel slip ~ U(slip_min, slip_max)          */
    double sensor_std;           /* noise_param.yaml:8-9   per-axis reading noise                      */
    double max_visible_dis;      /* noise_param.yaml:10    visibility radius                           */
    double wheel_base, wheel_radius; /* rigid2d/config/fake_turtle_param.yaml:6-7                     */
    int ticks_per_step;          /* 100 Hz simulator ticks per 10 Hz filter step (10)                  */
} ekf_sim_params;
void ekf_default_sim_params(ekf_sim_params* out);
/* world_xy: [n][2] landmark positions (host).  Replaces any uploaded log; T steps, vmax reading slots. */
ekf_status ekf_batch_simulate_known_log(ekf_batch_handle hb, const ekf_sim_params* sp, const double* world_xy,
                                        int T, int vmax);
/* Copies the device-resident log back (any pointer may be NULL): shapes as in ekf_known_log, plus
 * true_pose [T][B][3] = simulated ground truth (theta, x, y) AFTER step t (simulated logs only). */
ekf_status ekf_batch_download_log(ekf_batch_handle hb, double* twist, int* lm_idx, double* z_xy, double* init_xy,
                                  double* true_pose);

/* 2-D lidar of the simulator (publishScan, nurtlesim/src/tube_world.cpp:451-577). */
typedef struct ekf_lidar_params {
    int n_beams;          /* 360, tube_world.cpp:452                                      */
    double range_std;     /* nurtlesim/config/noise_param.yaml: range_std 0.005           */
    double range_max;     /* 3.5, tube_world.cpp:476                                      */
    double border_width;  /* tube_param.yaml: world_border_width 2.0 (square wall)        */
    double tube_radius;   /* tube_param.yaml: tube_radius 0.0762                          */
    /* 0 (default): clean ray geometry -- nearest of the wall hit, range_max and the first intersection of the beam with
     * every tube.  1: publishScan's own procedure, step for step (tube_world.cpp:496-570): a tube is looked at only by the
     * beams inside a bearing window of 2 atan2(radius, range_min) around it (:503,533-562), and the hit is the nearer of
     * the two intersections of the LINE through the robot and the beam's end point with the tube's circle
     * (getLineCircleIntersection, :420-450).  The two agree to rounding wherever no tube is closer than
     * radius / sin(atan2(radius, range_min)) = 0.142 m to the lidar (tests/test_host.py); closer tubes the window clips. */
    int model;
    double range_min;     /* 0.12, tube_world.cpp:475 (model 1: the bearing window)      */
} ekf_lidar_params;
void ekf_default_lidar_params(ekf_lidar_params* out);
/* Unknown-association inputs generated ON THE DEVICE for every filter of the batch (replaces any
 * uploaded unknown log): the simulated trajectory / odometry twists of ekf_batch_simulate_known_log, and
 * per (step, filter) up to jmax (<= 64) robot-frame measurements from
 *   lidar == NULL : the fake sensor -- noisy positions of the landmarks within max_visible_dis, shuffled
 *                   (the scan_measures vector of nuslam/src/unknown_data_assoc.cpp:309-320);
 *   lidar != NULL : a simulated laser scan per (step, filter) pushed through the batched circle
 *                   fitting (nuslam/src/landmarks.cpp:141 -> rigid2d::CircleFitting), scans never
 *                   leaving the device. */
ekf_status ekf_batch_simulate_unknown_log(ekf_batch_handle hb, const ekf_sim_params* sp,
                                          const ekf_lidar_params* lidar, const double* world_xy, int T, int jmax);
/* Copies the device-resident unknown log back (any pointer may be NULL): shapes as in ekf_unknown_log,
 * true_pose [T][B][3] for simulated logs. */
ekf_status ekf_batch_download_unknown_log(ekf_batch_handle hb, double* twist, int* count, double* meas_xy,
                                          double* true_pose);
/* Stand-alone scan simulator: scan s is taken from poses[s] = (theta, x, y) with the noise stream of
 * filter sp->first_filter_id + s at step `step`; ranges_out = [S][n_beams]. */
ekf_status ekf_simulate_scans(int device, const ekf_sim_params* sp, const ekf_lidar_params* lidar,
                              const double* world_xy, int n, const double* poses, int S, int step,
                              double* ranges_out);

/* Consistency of the batch against the simulated truth of step t:
 * out = {mean NEES (3 dof), max NEES, RMSE position, RMSE heading, mean trace of the pose covariance,
 *        fraction of filters with NEES < 7.815 (95 % chi-square bound, 3 dof)}. */
ekf_status ekf_batch_mc_stats(ekf_batch_handle hb, int t, double out[6]);

/* Covariance update mode.  0 (default) = EAGER: every landmark correction streams Sigma once
 * (16 N^2 bytes) -- the contract path the roofline is quoted on.  k > 0 = DELAYED rank-2k update
 * (SURVEY.md section 8(f) f2): up to k corrections are kept as low-rank factors
 * (Sigma = Sigma_base - sum K_j (H Sigma)_j), the rows/columns a correction needs are rebuilt on the
 * fly, prediction() maps the factors, and Sigma is rewritten once per k corrections (and before any
 * call that reads it: get_cov, checksum, clone, maha_scores; batch pools also before unknown-association runs).
 * A single filter's data_association() stays delayed too: the Mahalanobis scores are taken against Sigma_base minus
 * the pending pairs, the winner's correction is appended like any other.  Same results to rounding (tested at 1e-9);
 * k is capped at 64.
 * symmetric_gather != 0 (delayed mode only; opt-in, reported separately): the SYMMETRIC option.  The reference never
 * symmetrises Sigma, but (I - KH)Sigma keeps it symmetric to rounding (measured asymmetry 1e-18 relative, SURVEY.md
 * App. A2), and this option uses that:
 *   - the gain step takes Sigma H^T as (H Sigma)^T: only the rows Sigma(c, .) are rebuilt, from coalesced base rows and
 *     the V half of the pending store (no 16-KB-strided column gathers, half of the factor read);
 *   - for N >= 256 the flush forms the tiles on and above the diagonal only and writes each tile above it twice, in place
 *     and mirrored (4 N^2 bytes read + 8 N^2 written instead of 8 + 8; half of the multiply-adds): Sigma_base is then
 *     symmetric to the bit outside the 32 x 32 diagonal squares;
 *   - inside a known-association pool run the tiles on and above the diagonal ARE the covariance between flushes
 *     (prediction() keeps up the rows 1, 2 and leaves their strided column images to the next mirroring flush; every run
 *     returns with the full matrix restored).
 * Not the reference's operands -- do not use it on a covariance that was set deliberately asymmetric -- but within the
 * mode's 1e-9 (tests/test_gpu_delayed.py: 3.7e-13 on the bench configuration). */
ekf_status ekf_set_update_mode(ekf_handle h, int max_pending_corrections, int symmetric_gather);
ekf_status ekf_batch_set_update_mode(ekf_batch_handle hb, int max_pending_corrections, int symmetric_gather);
/* ---- dense general-F covariance propagation, fp32 on the matrix cores (BASELINE.json configs[3]) ----
 * Sigma <- F * Sigma * F^T + Q for an ARBITRARY dense F: the reference's expression
 * `sigma = At*sigma*At.t() + Q` (ekf_slam.cpp:101-102) as Armadillo executes it (two dense N^3
 * products).  The filter's own At = I + A never needs this (ekf_predict is O(N)); the entry point
 * serves motion models with a dense Jacobian and is checked against fp64 (tolerance 1e-4 per block).
 * All matrices are row-major N x N fp32 host buffers. */
typedef struct ekf_dense_s* ekf_dense_handle;
ekf_status ekf_dense_create(int N, int device, ekf_dense_handle* out);
ekf_status ekf_dense_destroy(ekf_dense_handle h);
/* Any of F, Sigma, Q may be NULL to keep the current device contents (initially all zero). */
ekf_status ekf_dense_set(ekf_dense_handle h, const float* F, const float* Sigma, const float* Q);
/* Applies the propagation `iterations` times; elapsed_ms (nullable) = HIP-event time of the launches. */
ekf_status ekf_dense_propagate(ekf_dense_handle h, int iterations, double* elapsed_ms);
ekf_status ekf_dense_get_sigma(ekf_dense_handle h, float* out);
/* Test / report hook: how one product of this handle is launched -- ld (N rounded up to 128), tiles = ld / 128 per
 * side, n_big = 256 x 128 tiles run by the main kernel (whole rounds of resident workgroups), n_tail = 128 x 128 tiles
 * cut into 64 x 64 quarters for the tail kernel behind it (what is left of the big-tile list, and the bottom strip of an
 * ld that is an odd multiple of 128).  Any pointer may be NULL. */
ekf_status ekf_dense_launch_info(ekf_dense_handle h, int* ld, int* tiles, int* n_big, int* n_tail);
/* ... and which kernel computes which 128 x 128 block of the result: map[tiles * tiles], row-major over blocks,
 * 0 = main kernel, 1 = tail kernel (the tests sample rows inside tail tiles with it). */
ekf_status ekf_dense_tile_map(ekf_dense_handle h, unsigned char* map);

/* ---- laser-scan front end: rigid2d::CircleFitting, batched ----------
 * std::vector<Vector2D> approxCirclePositions(std::vector<double> ranges)
 *                                          circle_fitting.hpp:27, circle_fitting.cpp:298-304
 * = clusteringRanges (:11-90) + circleRegression (:104-232) + classifyCircle (:234-296): the producer
 * of the `measures` argument of data_association (nuslam/src/landmarks.cpp:141 ->
 * unknown_data_assoc.cpp:309-320).  S scans of n_beams ranges each (beam i at angle 2*pi*i/n_beams):
 *   centres [S][max_out][2], radii [S][max_out], counts [S] (circles kept per scan, <= max_out);
 *   all_clusters (nullable) [S][128][4] = {x, y, r, is_circle} of EVERY cluster, n_clusters (nullable) [S].
 * n_beams <= 1024.  A scan without any cluster (undefined behaviour in the reference, :54) gives 0. */
ekf_status ekf_circle_fit_scans(int device, const double* ranges, int S, int n_beams, int max_out,
                                double* centres, double* radii, int* counts, double* all_clusters,
                                int* n_clusters);

#ifdef __cplusplus
}
#endif
#endif /* EKFSLAM_H */
