// ekf_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the EKF-SLAM filter core.
//
// Every kernel serves a POOL of B independent filters (B = 1 for a single rigid2d::EKF_SLAM
// object); blockIdx.z (or .y) selects the filter.  HBM layout per filter b:
//   sigma  [N][ld]   fp64 row-major, ld = N rounded up to 16 doubles (rows start on 128-B lines,
//                    so every lane moves aligned 16-B double2) or, where that costs at most 1/32 of
//                    a row, to 256 (rows on 2-KB boundaries: pick_ld below); pad columns stay 0
//   state  [ld]      [theta, x, y, m1x, m1y, ...]                  (ekf_slam.cpp:15-21,72-74)
//   Kg     [ld][2]   Kalman gain rows (K(r,0), K(r,1))   (scratch between gain and rank-2 kernel)
//   Gh     [2][ld]   rows of H*Sigma                      (pad entries 0)
//   rec              per-filter correction record (innovation, active flag)
//   snap   [4]       pose captured at the top of measurement()      (ekf_slam.cpp:109-111)
//
// Arithmetic is fp64 and is compiled with -ffp-contract=off so that the operation order is
// exactly that of oracle/ekf_oracle.c mode 1 ("structured"); only sin/cos/atan2 (OCML vs glibc)
// can differ in the last bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ekf {

constexpr double kPI = 3.14159265358979323846;  // rigid2d/include/rigid2d/rigid2d.hpp:13
constexpr int kWave = 64;

typedef double double2_t __attribute__((ext_vector_type(2)));
// Two doubles at an 8-byte-aligned address, fetched by ONE 16-byte access: the columns {3 + 2i, 4 + 2i} of a row (3 + 2i is
// odd), a thread's indices {3 + 2t, 4 + 2t}.  A column gather costs the memory pipe one request per lane and instruction
// whatever its width, so the five entries Sigma(r, c5(i)) are three loads ({0, 1}, {2}, {3 + 2i, 4 + 2i}), not five.
struct __attribute__((packed, aligned(8))) D2u { double x, y; };
__device__ __forceinline__ void gather_row5(const double* __restrict__ row, int lm, double (&p)[5]) {
    const D2u c01 = *reinterpret_cast<const D2u*>(row);
    const D2u cl = *reinterpret_cast<const D2u*>(row + 3 + 2 * (size_t)lm);
    p[0] = c01.x; p[1] = c01.y; p[2] = row[2]; p[3] = cl.x; p[4] = cl.y;
}

// the five non-zero columns of Hj for landmark lm: {0, 1, 2, 3+2lm, 4+2lm} (ekf_slam.cpp:164-170)
__host__ __device__ __forceinline__ int idx5(int k, int lm) { return k < 3 ? k : 3 + 2 * lm + (k - 3); }

// Delayed ("rank-2k") covariance update: Sigma = Sigma_base - sum_{j < count} U[j] V[j]^T, where every
// landmark correction appends the pair (K(:,0), (H Sigma)(0,:)), (K(:,1), (H Sigma)(1,:)) instead of
// streaming Sigma; k_flush folds all pending pairs into Sigma_base in ONE pass (16 N^2 bytes per FLUSH
// instead of per correction).  U and V are [B][cap][ld]; count is the same for every filter of a pool
// (a filter with nothing to correct in a slot appends a zero pair).
struct Pending {
    double* U;
    double* V;
    int cap;
    int count;
    int symmetric;  // 1: the gain step takes Sigma H^T as (H Sigma)^T (see ekf_set_update_mode); 2: and, inside a delayed
                    // known-association run whose flushes mirror (sym_flush_applies), the tiles on and above the diagonal
                    // ARE the covariance between flushes: k_predict leaves columns 1, 2 below the first square alone
    // Column panel (delayed known-association runs of a pool, ekf_batch_run_known; nullptr = off).  Sigma H^T reads
    // COLUMNS of Sigma (ekf_slam.cpp:178): entries 16 KB apart, one 64-byte sector fetched per 8 or 16 useful bytes.
    // The flush, which holds every new entry of Sigma_base in registers anyway, also writes the columns the NEXT
    // corrections will read as contiguous rows colp[b][row][ld]: rows 0..2 = columns 0, 1, 2; rows 3 + 2s, 4 + 2s =
    // the two columns of the landmark planned into slot s (lmslot[b][landmark] = s, or -1: that landmark's columns are
    // gathered from the matrix as before -- the plan is a hint, never a condition of correctness).  While a panel is
    // on, prediction() keeps the matrix's ROWS 1, 2 and the panel's rows 0..2 current and leaves the matrix's columns
    // 1, 2 (one sector per row, most of k_predict's traffic) to the next flush, which takes them from the panel.
    // Same values from another address: bit-identical to the run without a panel.
    double* colp;
    const short* lmslot;
    int colp_rows;   // rows of a filter's panel (3 + 2 x slots)
    // ... and the pending factors' own entries at the panel's indices, transposed: uvc[b][panel row of index c][j] =
    // (U_j(c), V_j(c)) for every pair j appended while this panel has been on.  A correction needs U_j, V_j at its 5 (7)
    // core indices for all pending j: contiguous in j here, one 64-byte sector per 8-byte entry in the factor store
    // (PMC, profiles/r04: 7 % of the gain kernel's fetched bytes).  Written by the lane that owns index c when it appends
    // a pair, kept up (entries of the indices 1, 2) by prediction().
    double2_t* uvc;
    // ... and CURRENT rows / columns, maintained lazily (nullptr = off).  Rebuilding Sigma(r, c) and Sigma(c, r) of a
    // correction's indices as "stored minus ALL pending pairs" reads the whole pending store (count x 32 bytes per index);
    // but a robot corrects the same few landmarks step after step, and the pose indices in every step.  cur[b][row][ld]
    // keeps, for the pose indices and every planned landmark, the column and the row AS THEY STOOD after `curv` pending
    // vectors; a gain launch starts from there, folds only the vectors appended since, and stores the result back.
    // prediction() maps the kept vectors like the rest of Sigma (their entries 1, 2; the pose vectors wholesale).
    // Rows: 0..2 columns 0, 1, 2; 3..5 rows 0, 1, 2; 6 + 4s + {0, 1} the columns of slot s, + {2, 3} its rows.
    // curv [B][1 + slots]: [0] the pose vectors, [1 + s] slot s; -1 = not kept (start from the stored entries).  A launch
    // reads curv_in and writes curv_out (the filter's other workgroups are still reading curv_in).
    double* cur;
    const int* curv_in;
    int* curv_out;
    // The kept POSE vectors (columns / rows 0..2 over the indices >= 3) are not rewritten by prediction(): it would cost ten
    // vectors of traffic per step for a map Sigma(k, 1) += Sigma(k, 0) a10 (...) that the next gain launch can apply to the
    // values it loads anyway.  k_predict adds its (a10, a20) to apred_in[b] (the map composes by addition: the two
    // off-diagonal entries of At multiply into zero), the gain launch applies the sum on load and writes 0 to apred_out[b].
    // The 3 x 3 pose block of the kept vectors and the landmarks' entries 1, 2 are mapped by k_predict itself, exactly.
    double* apred_in;
    double* apred_out;
};
// flush <-> panel: `in` (nullable) = the panel that has been on since the last flush (its rows 1, 2 ARE the matrix's
// columns 1, 2); `out` (nullable) = the panel to write for the landmarks of `lmslot`
struct PanelIO {
    const double* in;
    double* out;
    const short* lmslot;
    int rows;
};
constexpr int kSymSquare = 32;   // side of the mirrored flush's diagonal squares (its tile rows)

struct Params {
    double sigma0_landmark, q_pose, r_meas, gate_new, gate_update, straight_eps;
};

// One per filter: what the gain kernel decided / computed for the rank-2 kernel.
struct CorrRec {
    double nu0, nu1;  // innovation, bearing wrapped (ekf_slam.cpp:182-183)
    int active;       // 0 -> the rank-2 kernel skips this filter
    int lm;           // landmark index being corrected
    int n_active;     // > 0: this filter's own active dimension (<= the launch's N); 0: the launch's N
    int pad;
};

// Per-filter association state (data_association(), ekf_slam.cpp:278-402).
struct AssocRec {
    int known_count;  // leading run of known_list (:281-288), grows as landmarks are initialised
    int lm;           // decision for the current measurement (-1 = dropped)
    int active;
    int pad;
    double best;      // winning Mahalanobis distance (diagnostic)
};

// Where a correction's (landmark, reading) comes from.
enum : int { SRC_SENSOR_VECTOR = 0, SRC_COMPACT_LOG = 1, SRC_ASSOC = 2 };
struct CmdSrc {
    int mode;
    int lm_imm;             // SRC_SENSOR_VECTOR: landmark index (host loop over visible_list)
    const double* sensor;   // SRC_SENSOR_VECTOR: [B][2n] sensor_reading
    const int* lm_idx;      // SRC_COMPACT_LOG: slot table for this step, [B][vmax]
    const double* z_xy;     // SRC_COMPACT_LOG: [B][vmax][2]
    int vmax, v;
    const AssocRec* assoc;  // SRC_ASSOC: per-filter decision
    const double* meas;     // SRC_ASSOC: current measurement of filter b at meas[b * meas_stride + {0,1}]
    int meas_stride;        //            (2 for a single filter; jmax*2 inside a batch log step)
    int min_active;         // SRC_ASSOC, > 0: per-filter discovered prefix -- filter b is corrected within its
                            // leading max(min_active, 3 + 2*known_count_b) block only (exact, see ekf_associate)
    int fresh_pose;         // 1: read (theta,x,y) from state (:331-333); 0: from snap (:109-111)
    int write_snap;         // 1: this is the first correction of a measurement() call -- its fresh pose IS
                            //    the pose captured at :109-111; record it in snap for the corrections after it
};

// The current measurement of every filter for the association kernels.
struct MeasSrc {
    const double* xy;   // filter b reads xy[b * stride + {0, 1}]
    int stride;
    const int* count;   // nullable: filter b takes part in slot j iff j < count[b]
    int j;
};

struct PoolView {
    double* sigma;
    double* state;
    double* Kg;
    double* Gh;
    double* snap;
    CorrRec* rec;
    AssocRec* assoc;
    // Touched set: landmarks that have ever been corrected, in first-touch order.  Rows/columns of every
    // other landmark still hold their constructor values and are decoupled from everything (exact zeros
    // off the diagonal), so K(r,:) and (H Sigma)(:,c) are exact zeros there: a correction is a no-op on them.
    unsigned char* touch_flag;  // [B][n]
    int* touch_list;            // [B][n]
    int* touch_count;           // [B]
    int active_set;             // 1: kernels may skip the rows/columns of untouched landmarks (exact)
    int n, N, ld, B;
    size_t sigma_stride;  // doubles between consecutive filters' covariances = N * ld
    Params p;
    // Device error word in mapped host memory (one per pool; never null once the pool exists): a kernel that cannot go on
    // -- an in-kernel hand-off that never arrives -- sets a bit here instead of continuing with stale operands; the host
    // runtime turns a non-zero word into EKF_ERR_HIP at its next entry or synchronisation point (Pool::check_device).
    unsigned* err;
    // one word of device scratch: the tile queue of the resident streaming kernels (k_rank2_queue); nullptr = none
    unsigned* queue;
};
enum : unsigned { kErrHandoffTimeout = 1u };
__device__ __forceinline__ void report_device_error(const PoolView& pv, unsigned bit) {
    // a plain system-scope store (one bit is defined so far, so nothing can be lost): a read-modify-write on host memory
    // would need PCIe atomics
    __hip_atomic_store(pv.err, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Layout of a single filter's association block -- [record | record | decisions] on the device, and its mirror in mapped
// host memory [record | sequence number | decisions] that the deciding kernels publish to (ekf_runtime.hpp, Pool).
constexpr size_t kAssocRecSlot = 32, kAssocSeqOff = 32, kAssocDecOff = 64;
static_assert(sizeof(AssocRec) <= kAssocRecSlot, "AssocRec must fit its slot: the sequence number sits right behind it");

// called by ONE lane per filter and correction
__device__ __forceinline__ void touch_landmark(const PoolView& pv, int b, int lm) {
    unsigned char* tf = pv.touch_flag + (size_t)b * pv.n;
    if (!tf[lm]) {
        tf[lm] = 1;
        const int c = pv.touch_count[b];
        pv.touch_list[(size_t)b * pv.n + c] = lm;
        pv.touch_count[b] = c + 1;
    }
}

// rigid2d/src/rigid2d.cpp:336-345 -- double-fmod form, range (-pi, pi].
// fmod is exact, so for |rad| < 2*pi (every angle the filter produces in normal operation) the two calls reduce to
// identities and one exact subtraction: fmod(rad, 2pi) = rad, and s = rad + 2pi lies in [0, 4pi], where
// fmod(s, 2pi) is s or s - 2pi (exact by Sterbenz's lemma; s = 4pi gives 2pi here and 0 after the final fold,
// as the reference's 0).  Bit-identical to the two-fmod form for every input (tests/test_gpu_parity.py sweeps it
// against the reference build); the general form handles |rad| >= 2pi, NaN and infinities.
__device__ __forceinline__ double normalize_angle(double rad) {
    const double two_pi = (2 * kPI);
    double ang;
    if (fabs(rad) < two_pi) {
        ang = rad + two_pi;
        if (ang >= two_pi) ang = ang - two_pi;
    } else {
        double reduced_ang = fmod(rad, two_pi);
        ang = fmod((reduced_ang + two_pi), two_pi);
    }
    if (ang > kPI) ang = ang - two_pi;
    return ang;
}

struct MeasTerms {
    double z0, z1;      // (r, phi) of the reading               ekf_slam.cpp:142-146
    double zh0, zh1;    // predicted (r, phi), bearing wrapped   ekf_slam.cpp:152-155
    double H[2][5];     // non-zero columns {0,1,2,3+2i,4+2i}    ekf_slam.cpp:158-166
};

// the predicted half of measurement_terms (everything that depends on the state); m.z0, m.z1 are left alone
__device__ __forceinline__ void predicted_terms(double tx, double ty, double theta, double x, double y, MeasTerms& m);

__device__ __forceinline__ void measurement_terms(double tx, double ty, double sx, double sy, double theta,
                                                  double x, double y, MeasTerms& m) {
    m.z0 = sqrt(sx * sx + sy * sy);
    m.z1 = atan2(sy, sx);
    predicted_terms(tx, ty, theta, x, y, m);
}

__device__ __forceinline__ void predicted_terms(double tx, double ty, double theta, double x, double y, MeasTerms& m) {
    double delta_x = tx - x, delta_y = ty - y;
    double d = delta_x * delta_x + delta_y * delta_y;
    m.zh0 = sqrt(d);
    m.zh1 = normalize_angle(atan2(delta_y, delta_x) - theta);
    double sd = sqrt(d);
    // eight quotients, four divisions: (-a) / b == -(a / b) bit for bit in IEEE arithmetic (the sign is an exclusive or)
    const double xs = delta_x / sd, ys = delta_y / sd, yd = delta_y / d, xd = delta_x / d;
    m.H[0][0] = 0;  m.H[0][1] = -xs; m.H[0][2] = -ys;
    m.H[1][0] = -1; m.H[1][1] = yd;  m.H[1][2] = -xd;
    m.H[0][3] = xs;  m.H[0][4] = ys;
    m.H[1][3] = -yd; m.H[1][4] = xd;
}

// S = H Sigma H^T + R on the 5x5 sub-block, same summation order as the CPU restatement.
__device__ __forceinline__ void innovation_cov(const double S55[5][5], const double H[2][5], double r_meas,
                                               double S[2][2]) {
    double HS5[2][5];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int l = 0; l < 5; l++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) s += H[a][k] * S55[k][l];
            HS5[a][l] = s;
        }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < 5; l++) s += HS5[a][l] * H[b][l];
            S[a][b] = s;
        }
    S[0][0] += r_meas;
    S[1][1] += r_meas;
}

__device__ __forceinline__ void inv2(const double S[2][2], double Si[2][2]) {
    double det = S[0][0] * S[1][1] - S[0][1] * S[1][0];
    Si[0][0] = S[1][1] / det;  Si[0][1] = -S[0][1] / det;
    Si[1][0] = -S[1][0] / det; Si[1][1] = S[0][0] / det;
}

// Lane-parallel form of measurement_terms + innovation_cov + inv2 for ONE converged wavefront (all 64 lanes call it
// with the same arguments).  Every output is produced by exactly the operation sequence of the scalar functions above
// (bit-identical), but independent outputs are evaluated in different lanes: the dependent chain is one atan2, one sqrt,
// two divisions and two 5-term dot products instead of 2 atan2 + 12 divisions + 14 dot products on a single lane --
// the difference between ~0.4 us and ~1.3 us on the critical path of every correction of a single filter.
//   s55(k, l): Sigma(c5[k], c5[l]) (called by lanes < 10 with l = lane % 5; other lanes may read anything valid)
//   outH[10] = H[0][0..4], H[1][0..4]; outSi[4] = S^-1 row-major; outNu[2] = innovation (bearing wrapped iff wrap_nu)
// broadcast of a double from a compile-time lane to the whole wavefront through the scalar file (v_readlane_b32 x 2:
// no LDS round trip, unlike __shfl's ds_bpermute)
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src_lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src_lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <class S55Fn>
__device__ __forceinline__ void wave_terms(int lane, double tx, double ty, double sx, double sy, double theta, double x,
                                           double y, double r_meas, S55Fn s55, bool wrap_nu, double* outH, double* outSi,
                                           double* outNu) {
    // the five covariance entries of this lane's column are fetched first: they fly under the trigonometry
    const int ha = (lane / 5) & 1, hl = lane % 5;
    double s5[5];
#pragma unroll
    for (int k = 0; k < 5; k++) s5[k] = s55(k, hl);
    const double delta_x = tx - x, delta_y = ty - y;
    const double d = delta_x * delta_x + delta_y * delta_y;
    // odd lanes: the predicted reading (sqrt(d), atan2(delta_y, delta_x)); even lanes: the reading itself (:142-146)
    const bool pred = lane & 1;
    const double px = pred ? delta_x : sx, py = pred ? delta_y : sy;
    const double sq = sqrt(px * px + py * py);
    const double at = atan2(py, px);
    const double z0 = lane_bcast(sq, 0), z1 = lane_bcast(at, 0);
    const double sd = lane_bcast(sq, 1), atd = lane_bcast(at, 1);
    const double zh0 = sd, zh1 = normalize_angle(atd - theta);                       // :152-155
    // the eight quotients of H (:158-166), one per lane: q = 0..7 -> H01 H02 H11 H12 H03 H04 H13 H14
    const int q = lane & 7;
    const double num = (q == 0 || q == 3) ? -delta_x : (q == 1 || q == 6) ? -delta_y : (q == 2 || q == 5) ? delta_y : delta_x;
    const double den = (q == 0 || q == 1 || q == 4 || q == 5) ? sd : d;
    const double hq = num / den;
    const double H0[5] = {0.0, lane_bcast(hq, 0), lane_bcast(hq, 1), lane_bcast(hq, 4), lane_bcast(hq, 5)};
    const double H1[5] = {-1.0, lane_bcast(hq, 2), lane_bcast(hq, 3), lane_bcast(hq, 6), lane_bcast(hq, 7)};
    // (H Sigma55)(a, l) in lane 5a + l
    double hs = 0.0;
#pragma unroll
    for (int k = 0; k < 5; k++) hs += (ha ? H1[k] : H0[k]) * s5[k];
    // S(a, b) in lane 2a + b
    const int sa = (lane >> 1) & 1, sb = lane & 1;
    double sv = 0.0;
#pragma unroll
    for (int l = 0; l < 5; l++) {
        const double h0l = lane_bcast(hs, l), h1l = lane_bcast(hs, 5 + l);
        sv += (sa ? h1l : h0l) * (sb ? H1[l] : H0[l]);
    }
    if (sa == sb) sv += r_meas;
    const double S00 = lane_bcast(sv, 0), S01 = lane_bcast(sv, 1), S10 = lane_bcast(sv, 2), S11 = lane_bcast(sv, 3);
    const double det = S00 * S11 - S01 * S10;
    const int sq4 = lane & 3;
    const double si = (sq4 == 0 ? S11 : sq4 == 1 ? -S01 : sq4 == 2 ? -S10 : S00) / det;
    if (lane < 8) outH[q < 2 ? 1 + q : q < 4 ? 4 + q : q < 6 ? q - 1 : 2 + q] = hq;
    if (lane == 8) outH[0] = 0.0;
    if (lane == 9) outH[5] = -1.0;
    if (lane < 4) outSi[lane] = si;
    if (lane == 0) {
        outNu[0] = z0 - zh0;                                           // :182
        outNu[1] = wrap_nu ? normalize_angle(z1 - zh1) : z1 - zh1;     // :183 (the Mahalanobis score keeps it unwrapped, :269)
    }
}

// The same in two halves, for a pipeline in which the half that needs only the STATE (reading, prediction, H, nu: the
// trigonometry and the divisions) runs while another wavefront still updates the covariance block, and the half that
// needs the covariance (S, S^-1) follows.  Same operations as wave_terms, result for result.
__device__ __forceinline__ void wave_terms_h(int lane, double tx, double ty, double sx, double sy, double theta, double x,
                                             double y, bool wrap_nu, double* outH, double* outNu) {
    const double delta_x = tx - x, delta_y = ty - y;
    const double d = delta_x * delta_x + delta_y * delta_y;
    const bool pred = lane & 1;
    const double px = pred ? delta_x : sx, py = pred ? delta_y : sy;
    const double sq = sqrt(px * px + py * py);
    const double at = atan2(py, px);
    const double z0 = lane_bcast(sq, 0), z1 = lane_bcast(at, 0);
    const double sd = lane_bcast(sq, 1), atd = lane_bcast(at, 1);
    const double zh0 = sd, zh1 = normalize_angle(atd - theta);                       // :152-155
    const int q = lane & 7;
    const double num = (q == 0 || q == 3) ? -delta_x : (q == 1 || q == 6) ? -delta_y : (q == 2 || q == 5) ? delta_y : delta_x;
    const double den = (q == 0 || q == 1 || q == 4 || q == 5) ? sd : d;
    const double hq = num / den;
    if (lane < 8) outH[q < 2 ? 1 + q : q < 4 ? 4 + q : q < 6 ? q - 1 : 2 + q] = hq;
    if (lane == 8) outH[0] = 0.0;
    if (lane == 9) outH[5] = -1.0;
    if (lane == 0) {
        outNu[0] = z0 - zh0;                                           // :182
        outNu[1] = wrap_nu ? normalize_angle(z1 - zh1) : z1 - zh1;     // :183
    }
}

// wave_terms_h in two parts for two wavefronts that run side by side: the GEOMETRY part (ranges, the quotients of H,
// nu0: one sqrt and one division deep) and the ANGLE part (the two atan2, the wraps, nu1) share no intermediate result
// beyond delta, so the per-correction chain of a single filter is max(geometry, angle) instead of their sum.
// Same operations as wave_terms_h, result for result.
__device__ __forceinline__ void wave_terms_geo(int lane, double tx, double ty, double sx, double sy, double x, double y,
                                               double* outH, double* outNu0) {
    const double delta_x = tx - x, delta_y = ty - y;
    const double d = delta_x * delta_x + delta_y * delta_y;
    const bool pred = lane & 1;
    const double px = pred ? delta_x : sx, py = pred ? delta_y : sy;
    const double sq = sqrt(px * px + py * py);
    const double z0 = lane_bcast(sq, 0);
    const double sd = lane_bcast(sq, 1);
    const int q = lane & 7;
    const double num = (q == 0 || q == 3) ? -delta_x : (q == 1 || q == 6) ? -delta_y : (q == 2 || q == 5) ? delta_y : delta_x;
    const double den = (q == 0 || q == 1 || q == 4 || q == 5) ? sd : d;
    const double hq = num / den;
    if (lane < 8) outH[q < 2 ? 1 + q : q < 4 ? 4 + q : q < 6 ? q - 1 : 2 + q] = hq;
    if (lane == 8) outH[0] = 0.0;
    if (lane == 9) outH[5] = -1.0;
    if (lane == 0) *outNu0 = z0 - sd;                                  // :182
}
__device__ __forceinline__ void wave_terms_ang(int lane, double tx, double ty, double sx, double sy, double theta, double x,
                                               double y, bool wrap_nu, double* outNu1) {
    const double delta_x = tx - x, delta_y = ty - y;
    const bool pred = lane & 1;
    const double px = pred ? delta_x : sx, py = pred ? delta_y : sy;
    const double at = atan2(py, px);
    const double z1 = lane_bcast(at, 0);
    const double atd = lane_bcast(at, 1);
    const double zh1 = normalize_angle(atd - theta);                   // :152-155
    if (lane == 0) *outNu1 = wrap_nu ? normalize_angle(z1 - zh1) : z1 - zh1;   // :183
}

template <class S55Fn>
__device__ __forceinline__ void wave_terms_s(int lane, const double* H, double r_meas, S55Fn s55, double* outSi) {
    const int ha = (lane / 5) & 1, hl = lane % 5;
    double s5[5], H0[5], H1[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { s5[k] = s55(k, hl); H0[k] = H[k]; H1[k] = H[5 + k]; }
    double hs = 0.0;
#pragma unroll
    for (int k = 0; k < 5; k++) hs += (ha ? H1[k] : H0[k]) * s5[k];
    const int sa = (lane >> 1) & 1, sb = lane & 1;
    double sv = 0.0;
#pragma unroll
    for (int l = 0; l < 5; l++) {
        const double h0l = lane_bcast(hs, l), h1l = lane_bcast(hs, 5 + l);
        sv += (sa ? h1l : h0l) * (sb ? H1[l] : H0[l]);
    }
    if (sa == sb) sv += r_meas;
    const double S00 = lane_bcast(sv, 0), S01 = lane_bcast(sv, 1), S10 = lane_bcast(sv, 2), S11 = lane_bcast(sv, 3);
    const double det = S00 * S11 - S01 * S10;
    const int sq4 = lane & 3;
    const double si = (sq4 == 0 ? S11 : sq4 == 1 ? -S01 : sq4 == 2 ? -S10 : S00) / det;
    if (lane < 4) outSi[lane] = si;
}

// Leading dimension of Sigma, the factor vectors and the state: N rounded up to 16 doubles (rows on 128-byte lines), and
// to 256 doubles -- rows on 2-KB boundaries -- where that costs at most 1/32 of a row (n = 1000: 2003 -> 2048, n = 5000:
// 10003 -> 10240).  The strip-form flush reads a row as 2-KB pieces, one per workgroup: on a 2-KB boundary a piece is one
// DRAM page visit instead of two halves (48.3 -> 47.0 ms at 64 pending vectors, 46.8 -> 43.9 at 2; k_rank2 40.09 -> 39.90:
// profiles/r04/ld_alignment_ab.txt).
inline int pick_ld(int N) {
    const int wide = (N + 255) / 256 * 256;
    return (wide - N) * 32 <= N ? wide : (N + 15) / 16 * 16;
}

// ---- host-side launchers (ekf_kernels.hip) ---------------------------------------------------
struct Rank2Tuning {
    int rows_per_block;  // <= 0: automatic
    int nontemporal;     // < 0: automatic
    int group_rows;      // rows per load/store group U in {2,4,8}; other values: automatic
    int row_packing;     // 1: narrow full-width views may take the row-packed kernel (EKF_FORM_ROW_PACKING)
    int strip_flush;     // delayed mode: 0 never, 1 automatic, 2 always the strip-form flush (EKF_FORM_STRIP_FLUSH*)
    int tile_queue;      // 1: big pools stream the rank-2 update as resident workgroups on one tile queue (EKF_FORM_TILE_QUEUE)
};

void launch_init(const PoolView& pv, hipStream_t s);
// data_association() of a single filter ends here: the association record and the call's J decisions go to mapped host
// memory, then the sequence number (host layout: [record 32 B | seq 4 B | .. | decisions from byte 64])
void launch_publish_assoc(const AssocRec* rec, const int* decisions, int J, char* host, unsigned seq, hipStream_t s);
// prediction(): twist = imm (dtheta, dx) when twist_dev == nullptr, else twist_dev[b*2 + {0,1}]
void launch_raise_device_error(const PoolView& pv, hipStream_t s);   // test hook (ekf_test_raise_device_error)
void launch_predict(const PoolView& pv, const double* twist_dev, double dtheta, double dx, const Pending& pend,
                    hipStream_t s, double* pred_out = nullptr /* [B][2]: (a10, a20) of At, for the block cache */);
// Delayed mode: one kernel per correction (no covariance stream); reads state from pv.state, writes the
// corrected state to state_out (ping-pong) and appends the factor pair at rows pend.count, pend.count+1.
void launch_gain_delayed(const PoolView& pv, const CmdSrc& src, const Pending& pend, double* state_out,
                         hipStream_t s);
// Known-association log slots src.v and src.v + 1 (SRC_COMPACT_LOG) of every filter in ONE launch: the pending factor
// rows are read once for both corrections; appends TWO pairs (rows pend.count .. pend.count + 3).
void launch_gain_delayed_pair(const PoolView& pv, const CmdSrc& src, const Pending& pend, double* state_out,
                              hipStream_t s);
// Sigma_base -= sum_j U[j] V[j]^T for j < pend.count (count even); the caller then resets count to 0.
// Returns the form taken: 0 plain (k_flush), 1 strip (k_flush_strip).
int launch_flush(const PoolView& pv, const Pending& pend, const Rank2Tuning& t, hipStream_t s,
                 const PanelIO& panel = PanelIO{nullptr, nullptr, nullptr, 0});
// Column panel of a delayed known-association run (Pending::colp).  plan: which landmarks the next corrections of every
// filter touch -- the log slots lm_idx[t][b][v] of `nsteps` steps from `lm_idx` on, in order, the first `slots` distinct
// ones; plan_list [B][slots] remembers them so that the next plan can clear lmslot [B][n] again.
void launch_panel_plan(const PoolView& pv, const int* lm_idx, int nsteps, int vmax, int slots, short* lmslot, int* plan_list,
                       hipStream_t s, int* curv_reset = nullptr /* [B][1 + slots]: set to -1 (a flush follows) */,
                       double* apred_reset = nullptr /* [B][2]: set to 0 */);
// matrix columns 1, 2 <- panel rows 1, 2 (a run that ends on predictions with nothing pending)
void launch_panel_repair(const PoolView& pv, const double* colp, int colp_rows, hipStream_t s);
bool sym_flush_applies(const PoolView& pv, const Pending& pend, const Rank2Tuning& t);   // launch_flush would mirror
void launch_sym_repair(const PoolView& pv, hipStream_t s);
// top of measurement(): pose snapshot (+ first-call landmark initialisation from init_xy [B][2n])
void launch_measure_begin(const PoolView& pv, const double* init_xy, int do_init, hipStream_t s);
void launch_gain(const PoolView& pv, const CmdSrc& src, hipStream_t s);
// full_width: the caller guarantees valid K / G scratch over the whole row and column range (known-association paths),
// which lets narrow maps take the row-packed kernel
void launch_rank2(const PoolView& pv, const Rank2Tuning& t, hipStream_t s, bool full_width = false);
int rank2_packing(const PoolView& pv, const Rank2Tuning& t);
// the k_rank2<U, NT, TPB> instantiation and rows per workgroup launch_rank2 takes for this view (report hook)
void rank2_variant(const PoolView& pv, const Rank2Tuning& t, int* u, int* nontemporal, int* tpb, int* rows);
bool rank2_resident(const PoolView& pv, const Rank2Tuning& t);   // the launch runs as k_rank2_queue (EKF_FORM_TILE_QUEUE)
// Same update restricted to the rows of the touched set (exact: every other row has K = 0).
// max_touched: host-side upper bound of touch_count over the pool (sizes the grid).
void launch_rank2_active(const PoolView& pv, const Rank2Tuning& t, int max_touched, hipStream_t s);
void launch_touch_all(const PoolView& pv, hipStream_t s);  // marks every landmark touched (after set_cov)
// data_association(): scores for landmarks [0, known_count) of every filter, one landmark per wavefront
// parity hook of normalize_angle (a8): out[i] = normalize_angle(in[i])
void launch_normalize_angles(const double* in, int count, double* out, hipStream_t s);
// m_bound >= 0: host-side upper bound of every filter's known_count (sizes the grid)
// pend (nullable): delayed mode -- the scores are taken against Sigma_base minus the pending factor pairs
void launch_maha(const PoolView& pv, const MeasSrc& ms, double* scores /*[B][n]*/, int m_override, int m_bound,
                 hipStream_t s, const Pending* pend = nullptr);
void launch_assoc_begin(const PoolView& pv, const int* known_count_dev, int known_count_imm, hipStream_t s);
// decision for measurement j of every filter; assoc_out (nullable) gets [b*out_stride + j] = landmark or -1
// corr_counter (nullable): += number of filters whose decision leads to a correction
void launch_assoc_decide(const PoolView& pv, const MeasSrc& ms, const double* scores, int* assoc_out,
                         int out_stride, int j, unsigned long long* corr_counter, hipStream_t s);
// out[b][4] += {sum state, sum |state|, sum sigma, sum |sigma|}; caller zeroes out first
void launch_checksum(const PoolView& pv, double* out, hipStream_t s);
// batched rigid2d::CircleFitting::approxCirclePositions (ekf_circles.hip): S scans of nb beams ->
// centres [S][max_out][2], radii [S][max_out], counts [S]; all_out (nullable) [S][circles_max_clusters()][4]
void launch_circles(const double* ranges, int S, int nb, int max_out, double* centres, double* radii, int* counts,
                    double* all_out, int* n_clusters, hipStream_t s);
int circles_max_beams();
int circles_max_clusters();
// whole measurement() call of a SMALL map (one filter) in one single-workgroup, LDS-resident launch (ekf_small.hip),
// the inputs passed BY VALUE in the kernel-argument segment (n <= kSmallInlineN covers every N <= small_max_dim()):
// no staging buffer, no host-to-device copy, no copy -> kernel dependency in the stream.
// (has_twist: a prediction(dtheta, dx) deferred by the host runs first, on the LDS image)
constexpr int kSmallInlineN = 50;
struct SmallInline {
    double sensor[2 * kSmallInlineN];
    unsigned char visible[kSmallInlineN + 6];
};
void launch_small_measure_inline(const PoolView& pv, const SmallInline& in, int do_init, int has_twist, double dtheta,
                                 double dx, hipStream_t s);
constexpr int kSmallInlineJ = 32;
struct SmallInlineMeas { double xy[2 * kSmallInlineJ]; };  // the measures vector of data_association(), by value
void launch_small_associate_inline(const PoolView& pv, const SmallInlineMeas& in, int J, int known_count, int* assoc_out,
                                   int has_twist, double dtheta, double dx, hipStream_t s);
void launch_small_associate(const PoolView& pv, const double* meas, int J, int known_count, int* assoc_out,
                            int has_twist, double dtheta, double dx, hipStream_t s);
// one step of an unknown-association log for a pool whose every discovered prefix fits the small path:
// pv.N = the pool-wide bound of 3 + 2*(known_count + count) (<= small_max_dim()); meas [B][jmax][2], count [B]
void launch_pool_associate(const PoolView& pv, const double* meas, const int* count, int jmax, int min_active,
                           int* assoc_out /*[B][jmax]*/, unsigned long long* corr_counter, hipStream_t s);
// steps [t0, t1) of a compact known-association log for a pool of small maps (pv.N <= small_max_dim(),
// vmax <= 64) in one launch, Sigma resident in LDS throughout (log layout as ekf_known_log)
void launch_pool_run_known(const PoolView& pv, const double* twist, const int* lm_idx, const double* z_xy,
                           const double* init_xy, int vmax, int t0, int t1, int do_init, hipStream_t s);
int small_max_dim();          // largest N = 3 + 2n the small path accepts
hipError_t small_prepare();   // raises the kernel's dynamic-LDS limit (87 KB > 64 KB default)
// ---- a whole measurement() call as two launches: factor panels + one streaming pass (ekf_callfused.hip) ----
constexpr int kCallV = 8;          // corrections per pass (longer calls take several passes)
struct CallSrc {
    int mode;                      // SRC_SENSOR_VECTOR: vlist + sensor; SRC_COMPACT_LOG: lm_idx + z_xy of one log step
    const double* sensor;          // [B][2n]
    const int* vlist;              // [1 + V]: V, then the visible landmarks in ascending order (single filter)
    const int* lm_idx;             // [B][vmax], ascending, -1 padded
    const double* z_xy;            // [B][vmax][2]
    int vmax;
    int v0, vcount;                // this pass takes corrections [v0, v0 + vcount) of the call, vcount <= kCallV
    int fresh_pose;                // 1: first pass of a call -- the pose is read from the state and recorded in snap
    // SRC_INLINE (single filter, calls of <= kCallV visible landmarks): the landmarks and their readings travel BY VALUE
    // in the kernel-argument segment -- no staging buffer, no host-to-device copy, no copy -> kernel dependency
    int inl_lm[kCallV];
    double inl_xy[kCallV][2];
    long long* trace;              // nullable diagnostics: clock stamps of workgroup 0, [2][kTraceSlots] (ekf_phase_trace)
    // has_twist != 0 (first pass of a single filter's call): the prediction() that precedes the call (ekf_slam.cpp:55-106)
    // is folded in -- k_call_factors applies At Sigma At^T + Q to its panels and the core on the fly and records
    // (A10, A20) in pred_out[b][2]; k_rank2v applies it to every other element before the corrections.  Nothing else
    // writes Sigma in between, so there is no separate prediction launch.
    int has_twist;
    double dtheta, dx;
    double* pred_out;              // [B][2]
};
enum : int { SRC_INLINE = 3 };
constexpr int kTraceSlots = 64;
// U, V: [B][2 * kCallV][ld] factor rows (K_v(:,0), K_v(:,1) / G_v(0,:), G_v(1,:)); cnt [B]: corrections of this pass
void launch_call_factors(const PoolView& pv, const CallSrc& src, double* U, double* V, int* cnt, double* state_out,
                         hipStream_t s);
// vcount: corrections of the pass (the pool-wide maximum; filters with fewer have zero factor rows beyond theirs)
// pred (nullable): [B][2] = (A10, A20) of a prediction to apply to every element first (single filter, see CallSrc)
void launch_rank2v(const PoolView& pv, const double* U, const double* V, const int* cnt, int vcount, const Rank2Tuning& t,
                   hipStream_t s, const double* pred = nullptr);

// data_association() of a single filter with Sigma streamed once per call (ekf_assocfused.hip; readings travel by value).
// launch_assoc_score: scores + correction terms of the FIRST reading of a pass against the stored covariance minus the pc
// pending pairs.  launch_assoc_reading: reading (mx, my) -- decision from `scores` / `terms`, gain -> pair pc, state out of
// place -- and, when has_next, the scores / terms of the next reading (mxn, myn) into scores_out / terms_out (terms: [16][n]).
// m_bound: host bound of the known count in front of the reading.  Nb: active dimension of the reading (discovered prefix).  The caller
// ends the pass with launch_rank2v.
void launch_assoc_score(const PoolView& pv, double mx, double my, const AssocRec* assoc_in, const double* U, const double* V,
                        int pc, int m_bound, double* scores, double* terms, hipStream_t s, double* blk = nullptr);
void launch_assoc_reading(const PoolView& pv, double mx, double my, int has_next, double mxn, double myn,
                          const AssocRec* assoc_in, AssocRec* assoc_next, int* assoc_out_j, double* state_out, double* U,
                          double* V, int* cnt_out, int pc, int Nb, int zero_upto, int m_bound, const double* scores,
                          const double* terms, double* scores_out, double* terms_out, hipStream_t s,
                          long long* trace = nullptr, double* blk = nullptr, char* pub_host = nullptr, int pub_j = 0,
                          unsigned pub_seq = 0);
// A whole data_association() call (<= kCallV readings) of a single filter in ONE launch while the discovered part of the
// map fits one workgroup (`carried` >= max(known_count + J, touched_hwm) landmarks, <= assoc_call_capacity()): a thread per
// landmark keeps its block of Sigma current in registers.  In place on pv.state / pv.assoc; the pairs go to U / V, the caller
// ends the call with launch_rank2v.
struct AssocCallArgs {
    double xy[kCallV][2];   // the readings, by value
    int J;
    int known_count;        // host's known count in front of the call (= the device's)
    int touched_hwm;        // landmarks below it may carry non-constructor covariance
    int active_prefix;      // 1: corrections confined to the discovered prefix
};
int assoc_call_capacity();
void launch_assoc_call(const PoolView& pv, const AssocCallArgs& a, int carried, int* assoc_out, double* U, double* V,
                       int* cnt_out, int zero_upto, hipStream_t s, long long* trace = nullptr, char* pub_host = nullptr,
                       int pub_j0 = 0, unsigned pub_seq = 0);
int rank2v_round_count(int vcount);   // corrections per pass, rounded up to an instantiated count of k_rank2v

// one step of an unknown-association log for every filter of a pool in ONE launch, any prefix size (ekf_stepfused.hip):
// jmax <= kCallV readings per filter; U, V: [B][2 kCallV][ld] scratch for the step's factor pairs.  cnt_out != nullptr:
// the kernel leaves the covariance pass to the caller (launch_rank2v over the same U, V with cnt_out as its counts,
// zero_upto = rank2v_round_count(jmax)): big prefixes stream faster spread over the chip than one workgroup per filter.
void launch_pool_step_unknown(const PoolView& pv, const double* meas, const int* count, int jmax, int min_active,
                              int* assoc_out, double* U, double* V, unsigned long long* corr_counter, hipStream_t s,
                              int* cnt_out, int zero_upto, double* blocks /* [B][25][n] scratch: the landmarks' current blocks */);
// Delayed mode: the step's pairs (exactly jmax per filter, zero pairs beyond a filter's own readings) are appended to the
// pool's pending store behind the pend.count / 2 pairs of earlier steps, which the step's readings see subtracted; nothing
// is applied to Sigma (the caller flushes every few steps).  cnt_scratch: [B] ints.
void launch_pool_step_unknown_delayed(const PoolView& pv, const double* meas, const int* count, int jmax, int min_active,
                                      int* assoc_out, const Pending& pend, unsigned long long* corr_counter, int* cnt_scratch,
                                      double* blocks, const double* pred /* [B][2] (a10, a20) of the step's prediction */,
                                      hipStream_t s, const double* spec = nullptr, const int* specw = nullptr);
int step_pending_pairs_max();   // pairs a filter can carry between flushes in that mode
// Delayed mode, in front of launch_pool_step_unknown_delayed when pairs of earlier steps are pending: the "old part" of
// what the step's gains will need.  A reading's gain needs Sigma(r, c5) and Sigma(c5, r) of its winner as they stand now =
// stored entries minus ALL pending pairs, i.e. a pass over the whole pending store per reading.  The winner is decided
// reading by reading, but the old pairs' part of those rows / columns does not depend on this step's earlier readings:
// this launch GUESSES each reading's winner (the landmark nearest to where the reading lands from the step's pose),
// rebuilds "stored minus the old pairs" once for the <= kCallV guessed landmarks and for the pose indices, in the order the
// step kernel would take, and leaves it in spec [B][spec_rows()][ld] with the guesses in specw [B][kCallV].  The step kernel
// uses a guess that matches its decision and rebuilds from scratch otherwise: bit-identical either way.
int spec_rows();
void launch_pool_step_spec(const PoolView& pv, const double* meas, const int* count, int jmax, const Pending& pend,
                           double* spec, int* specw, hipStream_t s);

int max_pending();  // capacity limit of the delayed-update factor store (rows of U / V per filter)
void launch_gather_poses(const PoolView& pv, double* out, hipStream_t s);

}  // namespace ekf
