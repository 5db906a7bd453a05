// ekf_delayed.hip -- delayed rank-2k covariance update.
//
// The eager correction streams Sigma once per landmark (16 N^2 bytes, ekf_slam.cpp:191-192).  Here the
// rank-2 terms are kept as factors, Sigma = Sigma_base - sum_j U[j] V[j]^T, every quantity a correction
// needs from Sigma (5 rows, 5 columns, the 5x5 block) is reconstructed on the fly in O(k N), and
// Sigma_base is rewritten once per FLUSH.  Declared algorithmic bytes:
//   per correction   2*8*N*(k_pending) (read U, V)  + 10*8*N (base gathers) + 4*8*N (append)
//   per flush        2*8*N^2 + 2*8*N*k
// The terms are subtracted pair by pair in correction order.  The inner loops use fused multiply-adds
// (the flush is compute-bound in fp64 beyond ~16 pending corrections; the rest of the library is built
// -ffp-contract=off), so the delayed path matches the eager path to rounding, not bit for bit.
#include "ekf_kernels.hpp"

#include <cstdlib>
#include <mutex>

namespace ekf {

constexpr int kMaxPending = 128;

// 1-D grid of B x parts workgroups, XCD-aware (speed only): workgroup ids are dealt round-robin to the 8 XCDs, each with its
// own L2.  The `parts` workgroups of one filter all gather the same pending-factor entries at the correction's core
// indices (7 columns x count x 2 values, one sector each): a filter is given to ONE XCD so that three of four find them in
// its L2 (PMC, profiles/r04: dealt over four XCDs these gathers were 14 % of the gain kernel's fetched bytes).
__device__ __forceinline__ void xcd_decode(int id, int parts, int B, int& b, int& part) {
    const int full = (B / 8) * 8 * parts;
    if (id < full) {
        const int xcd = id & 7, slot = id >> 3;
        b = (slot / parts) * 8 + xcd;
        part = slot % parts;
    } else {   // the last B % 8 filters: plain order
        const int rest = id - full;
        b = (B / 8) * 8 + rest / parts;
        part = rest % parts;
    }
}

// Column panel (see Pending::uvc): the lane that owns the indices r, r + 1 files the entries of the pairs it appends
// (vectors [j0, j0 + NV), values k[q] / g[q] = U / V at (r, r + 1)) under the panel rows of those indices, if they have one
template <int NV>
__device__ __forceinline__ void uvc_append(const Pending& pend, int b, int n, int N, int r, int j0, const double2_t (&k)[NV],
                                           const double2_t (&g)[NV]) {
    if (!pend.uvc) return;
    const short* lms = pend.lmslot + (size_t)b * n;
    double2_t* uc = pend.uvc + (size_t)b * pend.colp_rows * pend.cap;
    int pr0 = -1, pr1 = -1;
    if (r < 3) pr0 = r; else if (r < N) { const int sl = lms[(r - 3) >> 1]; if (sl >= 0) pr0 = 3 + 2 * sl + ((r - 3) & 1); }
    if (r + 1 < 3) pr1 = r + 1; else if (r + 1 < N) { const int sl = lms[(r - 2) >> 1]; if (sl >= 0) pr1 = 3 + 2 * sl + ((r - 2) & 1); }
    if (pr0 >= 0) {
#pragma unroll
        for (int q = 0; q < NV; q++) uc[(size_t)pr0 * pend.cap + j0 + q] = double2_t{k[q].x, g[q].x};
    }
    if (pr1 >= 0) {
#pragma unroll
        for (int q = 0; q < NV; q++) uc[(size_t)pr1 * pend.cap + j0 + q] = double2_t{k[q].y, g[q].y};
    }
}

// ---------------------------------------------------------------------------------------------
// One landmark correction in delayed mode: ekf_slam.cpp:137-187 without the covariance stream.
// grid (ceil(ld/512), B).  Reads: state (in), Sigma_base, U/V rows [0, count); writes: U/V rows
// count, count+1, state_out, rec.  No buffer is both read and written -> race-free across workgroups.
// ---------------------------------------------------------------------------------------------
// SYM (ekf_set_update_mode's symmetric option): Sigma H^T is taken as (H Sigma)^T -- only the rows Sigma(c5, .) are
// rebuilt, from coalesced base rows and the V half of the pending store; the column gathers and the U half of the factor
// read (half of this kernel's traffic) are not made.
template <bool SYM>
__global__ __launch_bounds__(256) void k_gain_delayed(PoolView pv, CmdSrc src, Pending pend,
                                                      double* __restrict__ state_out) {
    int b, part;
    xcd_decode(blockIdx.x, (pv.ld / 2 + 255) / 256, pv.B, b, part);
    const int tid = threadIdx.x;
    const int N = pv.N, ld = pv.ld;
    const int rc = pend.count;
    __shared__ double sh_U5[5 * kMaxPending];
    __shared__ double sh_V5[5 * kMaxPending];
    __shared__ double sh_S55[25];
    __shared__ double sh_H[10];
    __shared__ double sh_Si[4];
    __shared__ double sh_nu[2];

    // a lane owns the two consecutive indices r, r+1 (16-B accesses); a workgroup covers 512 indices
    const int r = 2 * (part * 256 + tid);
    const double* st = pv.state + (size_t)b * ld;
    double* so = state_out + (size_t)b * ld;
    double* Ub = pend.U + (size_t)b * pend.cap * ld;
    double* Vb = pend.V + (size_t)b * pend.cap * ld;
    const double2_t zero2 = {0.0, 0.0};

    int lm = -1;
    double sx = 0.0, sy = 0.0;
    if (src.mode == SRC_SENSOR_VECTOR) {
        lm = src.lm_imm;
        sx = src.sensor[(size_t)b * 2 * pv.n + 2 * lm];
        sy = src.sensor[(size_t)b * 2 * pv.n + 2 * lm + 1];
    } else if (src.mode == SRC_COMPACT_LOG) {
        const size_t slot = (size_t)b * src.vmax + src.v;
        lm = src.lm_idx[slot];
        if (lm >= 0) {
            sx = src.z_xy[slot * 2];
            sy = src.z_xy[slot * 2 + 1];
        }
    } else {  // data_association(): the decision of k_assoc_decide, :330-390
        const AssocRec a = src.assoc[b];
        lm = a.active ? a.lm : -1;
        sx = src.meas[(size_t)b * src.meas_stride];
        sy = src.meas[(size_t)b * src.meas_stride + 1];
    }
    if (lm < 0 || lm >= pv.n) {  // nothing to correct: carry the state over, append a zero pair
        if (r < ld) {
            *reinterpret_cast<double2_t*>(so + r) = *reinterpret_cast<const double2_t*>(st + r);
            *reinterpret_cast<double2_t*>(Ub + (size_t)rc * ld + r) = zero2;
            *reinterpret_cast<double2_t*>(Ub + (size_t)(rc + 1) * ld + r) = zero2;
            *reinterpret_cast<double2_t*>(Vb + (size_t)rc * ld + r) = zero2;
            *reinterpret_cast<double2_t*>(Vb + (size_t)(rc + 1) * ld + r) = zero2;
            const double2_t z2[2] = {zero2, zero2};
            if (!SYM) uvc_append<2>(pend, b, pv.n, N, r, rc, z2, z2);
        }
        if (part == 0 && tid == 0) pv.rec[b].active = 0;
        return;
    }

    const double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    // column panel (see Pending): columns 0..2 of Sigma_base, and the columns of planned landmarks, as contiguous rows; the
    // matrix's own columns 1, 2 are stale below the pose block while it is on
    const double* __restrict__ cp = (!SYM && pend.colp) ? pend.colp + (size_t)b * pend.colp_rows * ld : nullptr;
    const int slot = cp ? pend.lmslot[(size_t)b * pv.n + lm] : -1;
    const double2_t* __restrict__ uc = (cp && pend.uvc) ? pend.uvc + (size_t)b * pend.colp_rows * pend.cap : nullptr;
    for (int idx = tid; idx < 5 * rc; idx += 256) {
        const int k = idx / rc, j = idx - k * rc;
        const int c = idx5(k, lm);
        const int pr = k < 3 ? k : (slot >= 0 ? 3 + 2 * slot + (k - 3) : -1);
        if (uc && pr >= 0) {   // the transposed copy: contiguous in j
            const double2_t uv = uc[(size_t)pr * pend.cap + j];
            sh_U5[k * kMaxPending + j] = uv.x;
            sh_V5[k * kMaxPending + j] = uv.y;
        } else {
            sh_U5[k * kMaxPending + j] = Ub[(size_t)j * ld + c];
            sh_V5[k * kMaxPending + j] = Vb[(size_t)j * ld + c];
        }
    }
    __syncthreads();
    if (tid < 25) {
        const int k = tid / 5, l = tid % 5;
        double v = SYM ? Sg[(size_t)min(idx5(k, lm), idx5(l, lm)) * ld + max(idx5(k, lm), idx5(l, lm))]   // (upper triangle)
                       : Sg[(size_t)idx5(k, lm) * ld + idx5(l, lm)];
        if (cp && k >= 3 && (l == 1 || l == 2)) v = cp[(size_t)l * ld + idx5(k, lm)];   // Sigma(c, 1), Sigma(c, 2)
        for (int j = 0; j < rc; j += 2)
            v = __builtin_fma(-sh_U5[k * kMaxPending + j + 1], sh_V5[l * kMaxPending + j + 1],
                              __builtin_fma(-sh_U5[k * kMaxPending + j], sh_V5[l * kMaxPending + j], v));
        sh_S55[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double theta, x, y;
        if (src.fresh_pose) {
            theta = st[0]; x = st[1]; y = st[2];
        } else {
            const double* sn = pv.snap + (size_t)b * 4;
            theta = sn[0]; x = sn[1]; y = sn[2];
        }
        if (src.write_snap && part == 0) {
            double* sn = pv.snap + (size_t)b * 4;
            sn[0] = theta; sn[1] = x; sn[2] = y;
        }
        MeasTerms m;
        measurement_terms(st[2 * lm + 3], st[2 * lm + 4], sx, sy, theta, x, y, m);
        double S55[5][5], S[2][2], Si[2][2];
        for (int k = 0; k < 5; k++)
            for (int l = 0; l < 5; l++) S55[k][l] = sh_S55[k * 5 + l];
        innovation_cov(S55, m.H, pv.p.r_meas, S);
        inv2(S, Si);
        for (int a = 0; a < 2; a++)
            for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = m.H[a][k];
        sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
        sh_nu[0] = m.z0 - m.zh0;                   // :182
        sh_nu[1] = normalize_angle(m.z1 - m.zh1);  // :183
        if (part == 0) {
            CorrRec rcd;
            rcd.nu0 = sh_nu[0]; rcd.nu1 = sh_nu[1]; rcd.active = 1; rcd.lm = lm; rcd.n_active = 0; rcd.pad = 0;
            pv.rec[b] = rcd;
            touch_landmark(pv, b, lm);
        }
    }
    __syncthreads();
    if (r >= ld) return;

    double2_t kv0 = zero2, kv1 = zero2, gv0 = zero2, gv1 = zero2, snew = zero2;
    if (r < N) {
        // current Sigma(r, c5[k]) and Sigma(c5[k], r) for r and r+1: base entries minus the pending pairs
        const bool two = r + 1 < N;  // N is odd: the last lane owns one real index and one pad index
        double2_t p[5], g[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int c = idx5(k, lm);
            g[k] = *reinterpret_cast<const double2_t*>(Sg + (size_t)c * ld + r);
            if (!SYM) {
                if (cp && (k < 3 || slot >= 0)) {   // the column as a panel row: one coalesced 16-byte load
                    p[k] = *reinterpret_cast<const double2_t*>(cp + (size_t)(k < 3 ? k : 3 + 2 * slot + (k - 3)) * ld + r);
                    if (!two) p[k].y = 0.0;
                } else {
                    p[k].x = Sg[(size_t)r * ld + c];
                    p[k].y = two ? Sg[(size_t)(r + 1) * ld + c] : 0.0;
                }
                if (cp && r < 4 && k >= 3) {   // Sigma(c, 1), Sigma(c, 2): the matrix's columns 1, 2 live in the panel
                    if (r == 0) g[k].y = cp[(size_t)1 * ld + c];
                    else g[k].x = cp[(size_t)2 * ld + c];
                }
            } else if (r < 4 && k >= 3) {   // Sigma(c, 1), Sigma(c, 2) from the rows 1, 2 (see Pending::symmetric == 2)
                if (r == 0) g[k].y = Sg[(size_t)1 * ld + c];
                else g[k].x = Sg[(size_t)2 * ld + c];
            }
        }
        // The factor rows stream through a ring of THREE register stages (pairs j, j + 2, j + 4 in flight while pair j is
        // folded in), unrolled by three so that a stage never changes registers: round 3's loop rotated the stages with
        // copies (ua = ua1 ...), and a copy of a register that a load is still writing waits for that load -- hipcc put
        // `s_waitcnt vmcnt(0)` at the top of every trip, i.e. ONE stage (64 B per lane) in flight: 4.0-4.8 TB/s of a
        // kernel that streams count x 32 B per lane and is bandwidth-bound.  Trips past the end re-read the last pair (a cache hit).
        struct Stage { double2_t ua, ub, va, vb; };
        auto ld4 = [&](int j, Stage& q) {
            const int jj = j < rc ? j : rc - 2;   // (past the end: the pair just read -- a cache hit, not a second fetch of pair 0)
            if (!SYM) {
                q.ua = *reinterpret_cast<const double2_t*>(Ub + (size_t)jj * ld + r);
                q.ub = *reinterpret_cast<const double2_t*>(Ub + (size_t)(jj + 1) * ld + r);
            }
            q.va = *reinterpret_cast<const double2_t*>(Vb + (size_t)jj * ld + r);
            q.vb = *reinterpret_cast<const double2_t*>(Vb + (size_t)(jj + 1) * ld + r);
        };
        auto fold = [&](int j, const Stage& q) {
            if (j >= rc) return;   // (uniform)
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const double v5a = sh_V5[k * kMaxPending + j], v5b = sh_V5[k * kMaxPending + j + 1];
                const double u5a = sh_U5[k * kMaxPending + j], u5b = sh_U5[k * kMaxPending + j + 1];
                if (!SYM) {
                    p[k].x = __builtin_fma(-q.ub.x, v5b, __builtin_fma(-q.ua.x, v5a, p[k].x));
                    p[k].y = __builtin_fma(-q.ub.y, v5b, __builtin_fma(-q.ua.y, v5a, p[k].y));
                }
                g[k].x = __builtin_fma(-u5b, q.vb.x, __builtin_fma(-u5a, q.va.x, g[k].x));
                g[k].y = __builtin_fma(-u5b, q.vb.y, __builtin_fma(-u5a, q.va.y, g[k].y));
            }
        };
        if (rc > 0) {
            Stage s0, s1, s2;
            s0.ua = s0.ub = s1.ua = s1.ub = s2.ua = s2.ub = zero2;
            ld4(0, s0); ld4(2, s1);
            for (int j = 0; j < rc; j += 6) {
                ld4(j + 4, s2); fold(j, s0);
                ld4(j + 6, s0); fold(j + 2, s1);
                ld4(j + 8, s1); fold(j + 4, s2);
            }
        }
        if (SYM) {
#pragma unroll
            for (int k = 0; k < 5; k++) { p[k] = g[k]; if (!two) p[k].y = 0.0; }
        }
        double2_t sht0 = zero2, sht1 = zero2;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sht0.x += p[k].x * sh_H[k];     sht0.y += p[k].y * sh_H[k];
            sht1.x += p[k].x * sh_H[5 + k]; sht1.y += p[k].y * sh_H[5 + k];
            gv0.x += sh_H[k] * g[k].x;      gv0.y += sh_H[k] * g[k].y;
            gv1.x += sh_H[5 + k] * g[k].x;  gv1.y += sh_H[5 + k] * g[k].y;
        }
        kv0.x = sht0.x * sh_Si[0] + sht1.x * sh_Si[2];  // K = (Sigma H^T) S^-1   :178
        kv1.x = sht0.x * sh_Si[1] + sht1.x * sh_Si[3];
        kv0.y = sht0.y * sh_Si[0] + sht1.y * sh_Si[2];
        kv1.y = sht0.y * sh_Si[1] + sht1.y * sh_Si[3];
        const double2_t sv = *reinterpret_cast<const double2_t*>(st + r);
        snew.x = sv.x + (kv0.x * sh_nu[0] + kv1.x * sh_nu[1]);  // :186
        snew.y = sv.y + (kv0.y * sh_nu[0] + kv1.y * sh_nu[1]);
        if (r == 0) snew.x = normalize_angle(snew.x);           // :187
        if (!two) { kv0.y = 0.0; kv1.y = 0.0; gv0.y = 0.0; gv1.y = 0.0; snew.y = 0.0; }  // pad stays 0
    }
    *reinterpret_cast<double2_t*>(Ub + (size_t)rc * ld + r) = kv0;
    *reinterpret_cast<double2_t*>(Ub + (size_t)(rc + 1) * ld + r) = kv1;
    *reinterpret_cast<double2_t*>(Vb + (size_t)rc * ld + r) = gv0;
    *reinterpret_cast<double2_t*>(Vb + (size_t)(rc + 1) * ld + r) = gv1;
    *reinterpret_cast<double2_t*>(so + r) = snew;
    if (!SYM) {
        const double2_t kq[2] = {kv0, kv1}, gq[2] = {gv0, gv1};
        uvc_append<2>(pend, b, pv.n, N, r, rc, kq, gq);
    }
}

// ---------------------------------------------------------------------------------------------
// TWO consecutive corrections of a measurement() call in delayed mode (log slots v and v + 1 of a known-association
// step) in one launch: every lane reads the pending factor rows of its indices ONCE for both corrections -- the
// O(kN) factor read is the dominant traffic of the delayed gain step (k_gain_delayed reads it per correction).
// Correction 2 sees the covariance after correction 1 through the 7 x 7 core block Sigma[C, C], C = c5(lm1) u c5(lm2),
// which every workgroup rebuilds from the base entries and the pending vectors at those 7 indices (tiny) and carries
// through correction 1 itself (the panel idea of ekf_callfused.hip): K1 and G1 at the indices of landmark 2, the
// landmark's position after the first state update, S2.  The lanes then build K1, G1 for their indices, apply the first
// correction to their 2 x 5 entries of landmark 2's rows / columns, and build K2, G2.  Both pairs are appended
// (rows count .. count + 3; a filter whose second slot is empty appends a zero pair, one whose slots are both empty two).
// Pose: the stale pose of the call (snap) for both corrections (ekf_slam.cpp:109-111).  grid (ceil(ld/512), B).
// ---------------------------------------------------------------------------------------------
template <bool SYM>
__global__ __launch_bounds__(256) void k_gain_delayed_pair(PoolView pv, CmdSrc src, Pending pend,
                                                           double* __restrict__ state_out) {
    int b, part;
    xcd_decode(blockIdx.x, (pv.ld / 2 + 255) / 256, pv.B, b, part);
    const int tid = threadIdx.x;
    const int N = pv.N, ld = pv.ld;
    const int rc = pend.count;
    __shared__ double sh_U7[7 * kMaxPending];
    __shared__ double sh_V7[7 * kMaxPending];
    __shared__ double sh_C[7][8];                 // core block Sigma[C, C] as it stands before the call's corrections
    __shared__ double sh_H[2][10], sh_Si[2][4], sh_nu[2][2];
    __shared__ double sh_K1B[5][2], sh_G1B[2][5]; // K1 / G1 at the five indices of landmark 2

    const int r = 2 * (part * 256 + tid);
    const double* st = pv.state + (size_t)b * ld;
    double* so = state_out + (size_t)b * ld;
    double* Ub = pend.U + (size_t)b * pend.cap * ld;
    double* Vb = pend.V + (size_t)b * pend.cap * ld;
    const double2_t zero2 = {0.0, 0.0};

    const size_t slot = (size_t)b * src.vmax + src.v;
    int lm1 = src.lm_idx[slot], lm2 = src.v + 1 < src.vmax ? src.lm_idx[slot + 1] : -1;
    if (lm1 >= pv.n) lm1 = -1;
    if (lm1 < 0 || lm2 >= pv.n) lm2 = -1;
    if (lm1 < 0) {  // nothing to correct: carry the state over, append two zero pairs
        if (r < ld) {
            *reinterpret_cast<double2_t*>(so + r) = *reinterpret_cast<const double2_t*>(st + r);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                *reinterpret_cast<double2_t*>(Ub + (size_t)(rc + q) * ld + r) = zero2;
                *reinterpret_cast<double2_t*>(Vb + (size_t)(rc + q) * ld + r) = zero2;
            }
            const double2_t z4[4] = {zero2, zero2, zero2, zero2};
            if (!SYM) uvc_append<4>(pend, b, pv.n, N, r, rc, z4, z4);
        }
        if (part == 0 && tid == 0) pv.rec[b].active = 0;
        if (!SYM && pend.cur && part == 0 && tid == 0) {   // the kept rows / columns stay as they are (see Pending::cur)
            const int nq = 1 + (pend.colp_rows - 3) / 2;
            for (int q = 0; q < nq; q++) pend.curv_out[(size_t)b * nq + q] = pend.curv_in[(size_t)b * nq + q];
            pend.apred_out[(size_t)b * 2] = pend.apred_in[(size_t)b * 2];           // (the deferred prediction map stays pending)
            pend.apred_out[(size_t)b * 2 + 1] = pend.apred_in[(size_t)b * 2 + 1];
        }
        return;
    }
    const bool two_corr = lm2 >= 0;
    const int lmB = two_corr ? lm2 : lm1;   // (a valid landmark for the address arithmetic of the second half)
    auto cidx = [&](int k) { return k < 3 ? k : k < 5 ? 3 + 2 * lm1 + (k - 3) : 3 + 2 * lmB + (k - 5); };   // C[0..6]
    // position of c5(lm1)[k] / c5(lm2)[k] inside C
    auto posA = [](int k) { return k; };
    auto posB = [](int k) { return k < 3 ? k : k + 2; };

    const double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    // column panel (see Pending; k_gain_delayed)
    const double* __restrict__ cp = (!SYM && pend.colp) ? pend.colp + (size_t)b * pend.colp_rows * ld : nullptr;
    const int slot1 = cp ? pend.lmslot[(size_t)b * pv.n + lm1] : -1;
    const int slot2 = cp ? pend.lmslot[(size_t)b * pv.n + lmB] : -1;
    // current rows / columns (see Pending::cur): the pose group and each landmark's group start from what was kept after
    // `st_*` pending vectors and fold only the vectors behind it; a group that was not kept starts from the stored entries
    const int ncurv = cp ? 1 + (pend.colp_rows - 3) / 2 : 0;
    double* __restrict__ cu = (cp && pend.cur) ? pend.cur + (size_t)b * (6 + 2 * (pend.colp_rows - 3)) * ld : nullptr;
    int st_p = 0, st_a = 0, st_b = two_corr ? 0 : rc;   // (one correction: the second landmark's group is not needed at all)
    bool kept_p = false, kept_a = false, kept_b = false;
    if (cu) {
        const int* cv = pend.curv_in + (size_t)b * ncurv;
        if (cv[0] >= 0) { kept_p = true; st_p = cv[0]; }
        if (slot1 >= 0 && cv[1 + slot1] >= 0) { kept_a = true; st_a = cv[1 + slot1]; }
        if (two_corr && slot2 >= 0 && cv[1 + slot2] >= 0) { kept_b = true; st_b = cv[1 + slot2]; }
    }
    // The lane's base entries Sigma(r, C[k]) and Sigma(C[k], r) are requested FIRST: they depend on nothing but the two
    // landmarks, and the workgroup's latency-bound prologue below (core block, the two corrections' terms on one lane,
    // three barriers: a tenth of a workgroup's life) then runs while they are in flight.
    const bool row_live = r < N;
    const bool two = r + 1 < N;  // N is odd: the last lane owns one real index and one pad index
    double2_t p[7], g[7];        // Sigma(r, C[k]) and Sigma(C[k], r) for r and r + 1
    if (row_live) {
        auto kept_of = [&](int k) { return k < 3 ? kept_p : k < 5 ? kept_a : kept_b; };
        // row of cur that keeps the column (col = true) or the row (col = false) of core index k
        auto cur_row = [&](int k, bool col) {
            return k < 3 ? (col ? k : 3 + k) : k < 5 ? 6 + 4 * slot1 + (col ? 0 : 2) + (k - 3) : 6 + 4 * slot2 + (col ? 0 : 2) + (k - 5);
        };
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int c = cidx(k);
            if (!SYM && kept_of(k)) {   // (uniform) the row as it stood after st_* pending vectors
                g[k] = *reinterpret_cast<const double2_t*>(cu + (size_t)cur_row(k, false) * ld + r);
                continue;
            }
            g[k] = *reinterpret_cast<const double2_t*>(Sg + (size_t)c * ld + r);
            if (SYM && r < 4 && k >= 3) {   // Sigma(c, 1), Sigma(c, 2) from the rows 1, 2 (see Pending::symmetric == 2)
                if (r == 0) g[k].y = Sg[(size_t)1 * ld + c];
                else g[k].x = Sg[(size_t)2 * ld + c];
            }
        }
        if (!SYM) {
            const double* rw0 = Sg + (size_t)r * ld;
            const double* rw1 = Sg + (size_t)(two ? r + 1 : r) * ld;
            if (cp) {
                // panel on: every column that has a panel row is ONE coalesced 16-byte load (r, r + 1 are neighbours in the
                // row); a landmark the plan did not hold is gathered from the matrix (its columns are current there)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    p[k] = *reinterpret_cast<const double2_t*>((kept_p ? cu + (size_t)cur_row(k, true) * ld : cp + (size_t)k * ld) + r);
                if (kept_a) {
                    p[3] = *reinterpret_cast<const double2_t*>(cu + (size_t)cur_row(3, true) * ld + r);
                    p[4] = *reinterpret_cast<const double2_t*>(cu + (size_t)cur_row(4, true) * ld + r);
                } else if (slot1 >= 0) {
                    p[3] = *reinterpret_cast<const double2_t*>(cp + (size_t)(3 + 2 * slot1) * ld + r);
                    p[4] = *reinterpret_cast<const double2_t*>(cp + (size_t)(4 + 2 * slot1) * ld + r);
                } else {
                    const D2u b0 = *reinterpret_cast<const D2u*>(rw0 + cidx(3)), b1 = *reinterpret_cast<const D2u*>(rw1 + cidx(3));
                    p[3].x = b0.x; p[4].x = b0.y; p[3].y = b1.x; p[4].y = b1.y;
                }
                if (kept_b) {
                    p[5] = *reinterpret_cast<const double2_t*>(cu + (size_t)cur_row(5, true) * ld + r);
                    p[6] = *reinterpret_cast<const double2_t*>(cu + (size_t)cur_row(6, true) * ld + r);
                } else if (slot2 >= 0) {
                    p[5] = *reinterpret_cast<const double2_t*>(cp + (size_t)(3 + 2 * slot2) * ld + r);
                    p[6] = *reinterpret_cast<const double2_t*>(cp + (size_t)(4 + 2 * slot2) * ld + r);
                } else {
                    const D2u c0 = *reinterpret_cast<const D2u*>(rw0 + cidx(5)), c1 = *reinterpret_cast<const D2u*>(rw1 + cidx(5));
                    p[5].x = c0.x; p[6].x = c0.y; p[5].y = c1.x; p[6].y = c1.y;
                }
                if (r < 4) {   // Sigma(c, 1), Sigma(c, 2) for the landmark rows c taken from the matrix: its columns 1, 2 live in
                               // the panel (a kept row carries them itself: prediction() keeps its entries 1, 2 up)
#pragma unroll
                    for (int k = 3; k < 7; k++) {
                        if (kept_of(k)) continue;
                        if (r == 0) g[k].y = cp[(size_t)1 * ld + cidx(k)];
                        else g[k].x = cp[(size_t)2 * ld + cidx(k)];
                    }
                }
            } else {
                // the seven column entries of a row as four loads: {0, 1}, {2}, and the two landmarks' neighbouring pairs
                const D2u a0 = *reinterpret_cast<const D2u*>(rw0), a1 = *reinterpret_cast<const D2u*>(rw1);
                const D2u b0 = *reinterpret_cast<const D2u*>(rw0 + cidx(3)), b1 = *reinterpret_cast<const D2u*>(rw1 + cidx(3));
                const D2u c0 = *reinterpret_cast<const D2u*>(rw0 + cidx(5)), c1 = *reinterpret_cast<const D2u*>(rw1 + cidx(5));
                p[0].x = a0.x; p[1].x = a0.y; p[2].x = rw0[2]; p[3].x = b0.x; p[4].x = b0.y; p[5].x = c0.x; p[6].x = c0.y;
                p[0].y = a1.x; p[1].y = a1.y; p[2].y = rw1[2]; p[3].y = b1.x; p[4].y = b1.y; p[5].y = c1.x; p[6].y = c1.y;
            }
            if (!two) {
#pragma unroll
                for (int k = 0; k < 7; k++) p[k].y = 0.0;
            }
            if (kept_p) {
                // the prediction(s) since these vectors were kept (Pending::apred_in): columns 1, 2 take column 0 * a, rows 1, 2
                // take a * row 0 -- k_predict's expressions -- at the indices >= 3 (the pose block was mapped by k_predict)
                const double as = pend.apred_in[(size_t)b * 2], bs = pend.apred_in[(size_t)b * 2 + 1];
                if (r >= 4) {
                    p[1].x = p[0].x * as + p[1].x; p[2].x = p[0].x * bs + p[2].x;
                    g[1].x = as * g[0].x + g[1].x; g[2].x = bs * g[0].x + g[2].x;
                }
                if (r >= 2 && two) {
                    p[1].y = p[0].y * as + p[1].y; p[2].y = p[0].y * bs + p[2].y;
                    g[1].y = as * g[0].y + g[1].y; g[2].y = bs * g[0].y + g[2].y;
                }
            }
        }
    }
    const double2_t* __restrict__ uc = (cp && pend.uvc) ? pend.uvc + (size_t)b * pend.colp_rows * pend.cap : nullptr;
    for (int idx = tid; idx < 7 * rc; idx += 256) {
        const int k = idx / rc, j = idx - k * rc;
        const int c = cidx(k);
        const int sl = k < 5 ? slot1 : slot2;
        const int pr = k < 3 ? k : (sl >= 0 ? 3 + 2 * sl + ((k - 3) & 1) : -1);
        if (uc && pr >= 0) {   // the transposed copy of the pending factors at the panel's indices: contiguous in j
            const double2_t uv = uc[(size_t)pr * pend.cap + j];
            sh_U7[k * kMaxPending + j] = uv.x;
            sh_V7[k * kMaxPending + j] = uv.y;
        } else {
            sh_U7[k * kMaxPending + j] = Ub[(size_t)j * ld + c];
            sh_V7[k * kMaxPending + j] = Vb[(size_t)j * ld + c];
        }
    }
    __syncthreads();
    if (tid < 49) {
        const int k = tid / 7, l = tid % 7;
        double v = SYM ? Sg[(size_t)min(cidx(k), cidx(l)) * ld + max(cidx(k), cidx(l))]   // (upper triangle)
                       : Sg[(size_t)cidx(k) * ld + cidx(l)];
        if (cp && k >= 3 && (l == 1 || l == 2)) v = cp[(size_t)l * ld + cidx(k)];   // Sigma(c, 1), Sigma(c, 2)
        for (int j = 0; j < rc; j += 2)
            v = __builtin_fma(-sh_U7[k * kMaxPending + j + 1], sh_V7[l * kMaxPending + j + 1],
                              __builtin_fma(-sh_U7[k * kMaxPending + j], sh_V7[l * kMaxPending + j], v));
        sh_C[k][l] = v;
    }
    __syncthreads();
    if (tid == 0) {
        const double* sn = pv.snap + (size_t)b * 4;
        double theta, x, y;
        if (src.fresh_pose) { theta = st[0]; x = st[1]; y = st[2]; }
        else { theta = sn[0]; x = sn[1]; y = sn[2]; }
        // ---- correction 1 on the core ----
        MeasTerms m;
        measurement_terms(st[2 * lm1 + 3], st[2 * lm1 + 4], src.z_xy[slot * 2], src.z_xy[slot * 2 + 1], theta, x, y, m);
        double S55[5][5], S[2][2], Si[2][2];
        for (int k = 0; k < 5; k++)
            for (int l = 0; l < 5; l++) S55[k][l] = sh_C[posA(k)][posA(l)];
        innovation_cov(S55, m.H, pv.p.r_meas, S);
        inv2(S, Si);
        for (int a = 0; a < 2; a++)
            for (int k = 0; k < 5; k++) sh_H[0][a * 5 + k] = m.H[a][k];
        sh_Si[0][0] = Si[0][0]; sh_Si[0][1] = Si[0][1]; sh_Si[0][2] = Si[1][0]; sh_Si[0][3] = Si[1][1];
        const double nu0 = m.z0 - m.zh0, nu1 = normalize_angle(m.z1 - m.zh1);   // :182-183
        sh_nu[0][0] = nu0; sh_nu[0][1] = nu1;
        if (two_corr) {
            // K1, G1 on the 7 core indices (the arithmetic of the lanes below, on core entries)
            double K1[7][2], G1[2][7];
            for (int i = 0; i < 7; i++) {
                double sht0 = 0.0, sht1 = 0.0, g0 = 0.0, g1 = 0.0;
                for (int k = 0; k < 5; k++) {
                    sht0 += sh_C[i][posA(k)] * m.H[0][k];
                    sht1 += sh_C[i][posA(k)] * m.H[1][k];
                    g0 += m.H[0][k] * sh_C[posA(k)][i];
                    g1 += m.H[1][k] * sh_C[posA(k)][i];
                }
                K1[i][0] = sht0 * Si[0][0] + sht1 * Si[1][0];
                K1[i][1] = sht0 * Si[0][1] + sht1 * Si[1][1];
                G1[0][i] = g0; G1[1][i] = g1;
            }
            for (int k = 0; k < 5; k++) {
                sh_K1B[k][0] = K1[posB(k)][0]; sh_K1B[k][1] = K1[posB(k)][1];
                sh_G1B[0][k] = G1[0][posB(k)]; sh_G1B[1][k] = G1[1][posB(k)];
            }
            // landmark 2 after the first state update (:186); the pose of the call stays the stale one
            const double t2x = st[2 * lm2 + 3] + (K1[5][0] * nu0 + K1[5][1] * nu1);
            const double t2y = st[2 * lm2 + 4] + (K1[6][0] * nu0 + K1[6][1] * nu1);
            MeasTerms m2;
            measurement_terms(t2x, t2y, src.z_xy[(slot + 1) * 2], src.z_xy[(slot + 1) * 2 + 1], theta, x, y, m2);
            for (int k = 0; k < 5; k++)
                for (int l = 0; l < 5; l++)   // the block of landmark 2 after correction 1 (:191-192 on the core)
                    S55[k][l] = sh_C[posB(k)][posB(l)] - (K1[posB(k)][0] * G1[0][posB(l)] + K1[posB(k)][1] * G1[1][posB(l)]);
            innovation_cov(S55, m2.H, pv.p.r_meas, S);
            inv2(S, Si);
            for (int a = 0; a < 2; a++)
                for (int k = 0; k < 5; k++) sh_H[1][a * 5 + k] = m2.H[a][k];
            sh_Si[1][0] = Si[0][0]; sh_Si[1][1] = Si[0][1]; sh_Si[1][2] = Si[1][0]; sh_Si[1][3] = Si[1][1];
            sh_nu[1][0] = m2.z0 - m2.zh0;
            sh_nu[1][1] = normalize_angle(m2.z1 - m2.zh1);
        }
        if (part == 0) {
            CorrRec rcd;
            rcd.nu0 = two_corr ? sh_nu[1][0] : nu0; rcd.nu1 = two_corr ? sh_nu[1][1] : nu1;
            rcd.active = 1; rcd.lm = two_corr ? lm2 : lm1; rcd.n_active = 0; rcd.pad = 0;
            pv.rec[b] = rcd;
            touch_landmark(pv, b, lm1);
            if (two_corr) touch_landmark(pv, b, lm2);
        }
    }
    __syncthreads();
    if (r >= ld) return;

    double2_t k1a = zero2, k1b = zero2, g1a = zero2, g1b = zero2, k2a = zero2, k2b = zero2, g2a = zero2, g2b = zero2;
    double2_t snew = zero2;
    if (row_live) {
        // three register stages, unrolled by three (see k_gain_delayed: no rotating copies, so that pairs j + 2 and j + 4
        // really are in flight while pair j is folded in)
        struct Stage { double2_t ua, ub, va, vb; };
        auto ld4 = [&](int j, Stage& q) {
            const int jj = j < rc ? j : rc - 2;   // (past the end: the pair just read -- a cache hit, not a second fetch of pair 0)
            if (!SYM) {
                q.ua = *reinterpret_cast<const double2_t*>(Ub + (size_t)jj * ld + r);
                q.ub = *reinterpret_cast<const double2_t*>(Ub + (size_t)(jj + 1) * ld + r);
            }
            q.va = *reinterpret_cast<const double2_t*>(Vb + (size_t)jj * ld + r);
            q.vb = *reinterpret_cast<const double2_t*>(Vb + (size_t)(jj + 1) * ld + r);
        };
        auto fold = [&](int j, const Stage& q) {
            if (j >= rc) return;   // (uniform)
#pragma unroll
            for (int k = 0; k < 7; k++) {
                if (j < (k < 3 ? st_p : k < 5 ? st_a : st_b)) continue;   // (uniform) this group was kept beyond vector j
                const double v5a = sh_V7[k * kMaxPending + j], v5b = sh_V7[k * kMaxPending + j + 1];
                const double u5a = sh_U7[k * kMaxPending + j], u5b = sh_U7[k * kMaxPending + j + 1];
                if (!SYM) {
                    p[k].x = __builtin_fma(-q.ub.x, v5b, __builtin_fma(-q.ua.x, v5a, p[k].x));
                    p[k].y = __builtin_fma(-q.ub.y, v5b, __builtin_fma(-q.ua.y, v5a, p[k].y));
                }
                g[k].x = __builtin_fma(-u5b, q.vb.x, __builtin_fma(-u5a, q.va.x, g[k].x));
                g[k].y = __builtin_fma(-u5b, q.vb.y, __builtin_fma(-u5a, q.va.y, g[k].y));
            }
        };
        const int j0 = min(st_p, min(st_a, st_b));   // the first pending vector any group still has to fold (even)
        if (rc > j0) {
            Stage s0, s1, s2;
            s0.ua = s0.ub = s1.ua = s1.ub = s2.ua = s2.ub = zero2;
            ld4(j0, s0); ld4(j0 + 2, s1);
            for (int j = j0; j < rc; j += 6) {
                ld4(j + 4, s2); fold(j, s0);
                ld4(j + 6, s0); fold(j + 2, s1);
                ld4(j + 8, s1); fold(j + 4, s2);
            }
        }
        if (!SYM && cu) {
            // the rows / columns as they stand NOW (every pending vector folded, this launch's corrections not yet): kept for
            // the next launch that meets the pose or these landmarks again (Pending::cur); curv_out says up to which vector
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int sl = k < 5 ? slot1 : slot2;
                if ((k >= 3 && sl < 0) || (k >= 5 && !two_corr)) continue;   // (uniform) no slot / no second landmark
                const int rowc = k < 3 ? k : 6 + 4 * sl + ((k - 3) & 1), rowr = k < 3 ? 3 + k : rowc + 2;
                *reinterpret_cast<double2_t*>(cu + (size_t)rowc * ld + r) = p[k];
                *reinterpret_cast<double2_t*>(cu + (size_t)rowr * ld + r) = g[k];
            }
        }
        if (SYM) {
#pragma unroll
            for (int k = 0; k < 7; k++) { p[k] = g[k]; if (!two) p[k].y = 0.0; }
        }
        // ---- correction 1: K1 = (Sigma H1^T) S1^-1, G1 = H1 Sigma on the lane's indices (:178) ----
        double2_t sht0 = zero2, sht1 = zero2;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sht0.x += p[k].x * sh_H[0][k];     sht0.y += p[k].y * sh_H[0][k];
            sht1.x += p[k].x * sh_H[0][5 + k]; sht1.y += p[k].y * sh_H[0][5 + k];
            g1a.x += sh_H[0][k] * g[k].x;      g1a.y += sh_H[0][k] * g[k].y;
            g1b.x += sh_H[0][5 + k] * g[k].x;  g1b.y += sh_H[0][5 + k] * g[k].y;
        }
        k1a.x = sht0.x * sh_Si[0][0] + sht1.x * sh_Si[0][2];
        k1b.x = sht0.x * sh_Si[0][1] + sht1.x * sh_Si[0][3];
        k1a.y = sht0.y * sh_Si[0][0] + sht1.y * sh_Si[0][2];
        k1b.y = sht0.y * sh_Si[0][1] + sht1.y * sh_Si[0][3];
        const double2_t sv = *reinterpret_cast<const double2_t*>(st + r);
        snew.x = sv.x + (k1a.x * sh_nu[0][0] + k1b.x * sh_nu[0][1]);  // :186
        snew.y = sv.y + (k1a.y * sh_nu[0][0] + k1b.y * sh_nu[0][1]);
        if (r == 0) snew.x = normalize_angle(snew.x);                 // :187
        if (two_corr) {
            // ---- the lane's entries of landmark 2's rows / columns after correction 1 (:191-192), then correction 2 ----
            sht0 = zero2; sht1 = zero2;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int q = k < 3 ? k : k + 2;   // position of c5(lm2)[k] in C
                const double px = p[q].x - (k1a.x * sh_G1B[0][k] + k1b.x * sh_G1B[1][k]);
                const double py = p[q].y - (k1a.y * sh_G1B[0][k] + k1b.y * sh_G1B[1][k]);
                const double gx = g[q].x - (sh_K1B[k][0] * g1a.x + sh_K1B[k][1] * g1b.x);
                const double gy = g[q].y - (sh_K1B[k][0] * g1a.y + sh_K1B[k][1] * g1b.y);
                sht0.x += px * sh_H[1][k];     sht0.y += py * sh_H[1][k];
                sht1.x += px * sh_H[1][5 + k]; sht1.y += py * sh_H[1][5 + k];
                g2a.x += sh_H[1][k] * gx;      g2a.y += sh_H[1][k] * gy;
                g2b.x += sh_H[1][5 + k] * gx;  g2b.y += sh_H[1][5 + k] * gy;
            }
            k2a.x = sht0.x * sh_Si[1][0] + sht1.x * sh_Si[1][2];
            k2b.x = sht0.x * sh_Si[1][1] + sht1.x * sh_Si[1][3];
            k2a.y = sht0.y * sh_Si[1][0] + sht1.y * sh_Si[1][2];
            k2b.y = sht0.y * sh_Si[1][1] + sht1.y * sh_Si[1][3];
            snew.x = snew.x + (k2a.x * sh_nu[1][0] + k2b.x * sh_nu[1][1]);
            snew.y = snew.y + (k2a.y * sh_nu[1][0] + k2b.y * sh_nu[1][1]);
            if (r == 0) snew.x = normalize_angle(snew.x);
        }
        if (!two) {  // pad stays 0
            k1a.y = 0.0; k1b.y = 0.0; g1a.y = 0.0; g1b.y = 0.0; k2a.y = 0.0; k2b.y = 0.0; g2a.y = 0.0; g2b.y = 0.0; snew.y = 0.0;
        }
    }
    *reinterpret_cast<double2_t*>(Ub + (size_t)rc * ld + r) = k1a;
    *reinterpret_cast<double2_t*>(Ub + (size_t)(rc + 1) * ld + r) = k1b;
    *reinterpret_cast<double2_t*>(Vb + (size_t)rc * ld + r) = g1a;
    *reinterpret_cast<double2_t*>(Vb + (size_t)(rc + 1) * ld + r) = g1b;
    *reinterpret_cast<double2_t*>(Ub + (size_t)(rc + 2) * ld + r) = k2a;
    *reinterpret_cast<double2_t*>(Ub + (size_t)(rc + 3) * ld + r) = k2b;
    *reinterpret_cast<double2_t*>(Vb + (size_t)(rc + 2) * ld + r) = g2a;
    *reinterpret_cast<double2_t*>(Vb + (size_t)(rc + 3) * ld + r) = g2b;
    *reinterpret_cast<double2_t*>(so + r) = snew;
    if (!SYM) {
        const double2_t kq[4] = {k1a, k1b, k2a, k2b}, gq[4] = {g1a, g1b, g2a, g2b};
        uvc_append<4>(pend, b, pv.n, N, r, rc, kq, gq);
        if (cu && part == 0 && tid == 0) {   // (the filter's other workgroups read curv_in: the counts go to the other buffer)
            int* co = pend.curv_out + (size_t)b * ncurv;
            const int* ci = pend.curv_in + (size_t)b * ncurv;
            for (int q = 0; q < ncurv; q++) co[q] = ci[q];
            co[0] = rc;
            if (slot1 >= 0) co[1 + slot1] = rc;
            if (two_corr && slot2 >= 0) co[1 + slot2] = rc;
            pend.apred_out[(size_t)b * 2] = 0.0;       // (applied by this launch, or nothing was kept to apply it to)
            pend.apred_out[(size_t)b * 2 + 1] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Flush: Sigma_base(r, c) -= sum over pending pairs (U[j](r) V[j](c) + U[j+1](r) V[j+1](c)).
// Same streaming structure as k_rank2 (strip of 256 double2 columns, 16-row register groups, loads
// before stores, non-temporal), with a loop over the pairs: U values are wave-uniform (scalar loads),
// V values are per lane and come from L2 (count x 16 KB per filter).  2*8*N^2 bytes, 2*count*N^2 flop.
// ---------------------------------------------------------------------------------------------
// Column panel (see Pending): the panel row that holds column `col` of Sigma_base, or -1
__device__ __forceinline__ int panel_row_of(int col, int N, const short* __restrict__ lms) {
    if (col < 3) return col;
    if (col >= N) return -1;
    const int sl = lms[(col - 3) >> 1];
    return sl >= 0 ? 3 + 2 * sl + ((col - 3) & 1) : -1;
}

template <int U_ROWS, bool NT>
__device__ __forceinline__ void flush_load(double2_t (&a)[U_ROWS], const double2_t* col, int r, int ld2n) {
#pragma unroll
    for (int u = 0; u < U_ROWS; u++) {
        if constexpr (NT) a[u] = __builtin_nontemporal_load(col + (size_t)(r + u) * ld2n);
        else a[u] = col[(size_t)(r + u) * ld2n];
    }
}
template <int U_ROWS, bool NT>
__device__ __forceinline__ void flush_store(const double2_t (&a)[U_ROWS], double2_t* col, int r, int ld2n) {
#pragma unroll
    for (int u = 0; u < U_ROWS; u++) {
        if constexpr (NT) __builtin_nontemporal_store(a[u], col + (size_t)(r + u) * ld2n);
        else col[(size_t)(r + u) * ld2n] = a[u];
    }
}
// a(u) -= sum over pending pairs of U[j](r+u) V[j](c): 4 * U_ROWS v_fma_f64 per pair, U as scalar operands
template <int U_ROWS>
__device__ __forceinline__ void flush_apply(double2_t (&a)[U_ROWS], const double* __restrict__ Ub,
                                            const double2_t* __restrict__ Vb, int r, int ld, int ld2n, int count) {
    // V of pair j+2 AND j+4 are in flight while pair j's FMAs run (PMC: with one pair of lookahead the waves
    // sat in s_waitcnt half of the time: an L2 hit under load outlasts one pair's 256 FMA cycles).  Three register
    // stages, unrolled by three: stages rotated by copies make hipcc wait for every load at the top of each trip (a copy
    // of a register that a load is still writing waits for that load).  Trips past the end re-read the last pair (a cache hit).
    struct Stage { double2_t v0, v1; };
    auto ldv = [&](int j, Stage& q) {
        const int jj = j < count ? j : count - 2;
        q.v0 = Vb[(size_t)jj * ld2n];
        q.v1 = Vb[(size_t)(jj + 1) * ld2n];
    };
    auto fold = [&](int j, const Stage& q) {
        if (j >= count) return;   // (uniform)
        const double* __restrict__ u0 = Ub + (size_t)j * ld + r;  // wave-uniform -> scalar loads
        const double* __restrict__ u1 = u0 + ld;
#pragma unroll
        for (int u = 0; u < U_ROWS; u++) {
            const double k0 = -u0[u], k1 = -u1[u];
            a[u].x = __builtin_fma(k1, q.v1.x, __builtin_fma(k0, q.v0.x, a[u].x));
            a[u].y = __builtin_fma(k1, q.v1.y, __builtin_fma(k0, q.v0.y, a[u].y));
        }
    };
    Stage s0, s1, s2;
    ldv(0, s0); ldv(2, s1);
    for (int j = 0; j < count; j += 6) {
        ldv(j + 4, s2); fold(j, s0);
        ldv(j + 6, s0); fold(j + 2, s1);
        ldv(j + 8, s1); fold(j + 4, s2);
    }
}

// One U_ROWS-row group at a time at 4 waves/SIMD: measured faster than loading two groups up front
// (206 VGPRs, 2 waves/SIMD): beyond ~16 pending corrections the kernel is bound by fp64 FMA issue
// (tools/flush_sweep.py), and wave-level parallelism hides the group loads better than a deeper pipeline.
template <int U_ROWS, bool NT>
__global__ __launch_bounds__(256) void k_flush(double* __restrict__ sigma, const double* __restrict__ Uall,
                                               const double* __restrict__ Vall, int N, int ld, size_t sigma_stride,
                                               int cap, int count, int rows_per_block, int strips,
                                               int row_blocks, int B, PanelIO pio, int n) {
    // 1-D grid, XCD-aware decode (speed only): workgroup ids are dealt round-robin to the 8 XCDs, each with
    // its own 4 MB L2.  All workgroups of one filter re-read that filter's factors (count x 32 KB), so a
    // filter is given to ONE XCD: id -> (xcd = id % 8, slot = id / 8), filter = 8 * (slot / P) + xcd.
    const int P = strips * row_blocks;
    int b, p;
    const int full = (B / 8) * 8 * P;
    if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        p = slot % P;
    } else {  // the last B % 8 filters: plain order
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        p = rest % P;
    }
    const int ld2n = ld >> 1;
    const int c2 = (p % strips) * 256 + threadIdx.x;
    const int row_begin = (p / strips) * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);
    if (c2 >= (N + 1) / 2) return;   // (the columns behind N are padding: zero, and a product with V's zero padding keeps them so)
    const double* __restrict__ Ub = Uall + (size_t)b * cap * ld;
    const double2_t* __restrict__ Vb = reinterpret_cast<const double2_t*>(Vall + (size_t)b * cap * ld) + c2;
    double2_t* __restrict__ col = reinterpret_cast<double2_t*>(sigma + (size_t)b * sigma_stride) + c2;

    // column panel: columns 1, 2 of the base come from the panel that has been on (pio.in), and the new values of the
    // planned columns go out as panel rows (pio.out) -- per lane at most two of its 2 x U_ROWS values per group
    const double* __restrict__ pin = pio.in ? pio.in + (size_t)b * pio.rows * ld : nullptr;
    double* __restrict__ pout = pio.out ? pio.out + (size_t)b * pio.rows * ld : nullptr;
    const bool fix1 = pin && c2 == 0, fix2 = pin && c2 == 1;   // column 1 = .y of double2 0, column 2 = .x of double2 1
    int prow0 = -1, prow1 = -1;
    if (pout) {
        const short* lms = pio.lmslot + (size_t)b * n;
        prow0 = panel_row_of(2 * c2, N, lms);
        prow1 = panel_row_of(2 * c2 + 1, N, lms);
    }
    auto panel_in = [&](auto& a, int r0, int rows) {
        if (fix1) for (int u = 0; u < rows; u++) a[u].y = pin[(size_t)ld + r0 + u];
        if (fix2) for (int u = 0; u < rows; u++) a[u].x = pin[(size_t)2 * ld + r0 + u];
    };
    auto panel_out = [&](const auto& a, int r0, int rows) {
        if (prow0 >= 0) for (int u = 0; u < rows; u++) pout[(size_t)prow0 * ld + r0 + u] = a[u].x;
        if (prow1 >= 0) for (int u = 0; u < rows; u++) pout[(size_t)prow1 * ld + r0 + u] = a[u].y;
    };
    int r = row_begin;
    for (; r + U_ROWS <= row_end; r += U_ROWS) {
        double2_t a[U_ROWS];
        flush_load<U_ROWS, NT>(a, col, r, ld2n);
        panel_in(a, r, U_ROWS);
        flush_apply<U_ROWS>(a, Ub, Vb, r, ld, ld2n, count);
        flush_store<U_ROWS, NT>(a, col, r, ld2n);
        panel_out(a, r, U_ROWS);
    }
    for (; r < row_end; r++) {
        double2_t a[1];
        flush_load<1, false>(a, col, r, ld2n);
        panel_in(a, r, 1);
        flush_apply<1>(a, Ub, Vb, r, ld, ld2n, count);
        flush_store<1, false>(a, col, r, ld2n);
        panel_out(a, r, 1);
    }
}

// ---------------------------------------------------------------------------------------------
// Flush, second form (pools that fill the chip): a workgroup of 16 waves owns a strip of 256 columns for a row range.
//   * The strip of all pending V rows (count x 2 KB, <= 160 KB = 80 vectors) is read ONCE into LDS and shared by the waves: V
//     costs no L2 bandwidth and no vector-memory wait in the FMA loop (k_flush: two L2 loads per pair and wave,
//     waited for in the trip that issues them -- hipcc collapses the source's rotating prefetch).
//   * A lane holds 8 rows x 4 columns (k_flush: 16 x 2): the same 64 accumulator registers, but a scalar operand
//     U[j](r) now feeds FOUR v_fma_f64, so one batch of scalar loads (64 SGPRs: 4 vectors x 8 rows) is followed by
//     128 FMAs instead of 64.  Scalar loads return out of order: only one batch can be in flight, its latency
//     (~1600 cycles under a saturated L2, from the measured 50% FMA utilisation of the 2-vector form) is exposed
//     once per batch and covered by the other three waves of the SIMD only -- halving the batches per unit of work
//     is what lifts the FMA pipe from ~50% busy towards the HBM time of the stream.
// Same order of fused multiply-adds per element as k_flush -> bit-identical results.
// ---------------------------------------------------------------------------------------------
constexpr int kStripWaves = 16;
constexpr int kStripCols2 = 128;   // double2 columns per strip: lane l holds columns l and l + 64
constexpr int kStripMaxVec = 80;   // pending vectors whose strip fits the 160 KB of LDS (2 KB each)
constexpr int kStripFromCount = 40;        // automatic choice: the strip form beyond this many pending vectors ...
constexpr int kStripMinWorkgroups = 512;   // ... on pools with at least this many strip workgroups (2 per CU)

// `col` = the lane's first column (base + c); its second column is 64 double2 (1 KB) further on: one address register
// pair per row serves both.  live0 / live1: lanes past the last column of the matrix take no part.
// `last`: the last row of the workgroup's range -- the rows of a partial group behind it re-read that row (never stored).
template <bool NT>
__device__ __forceinline__ void fl_load(double2_t (&a)[8][2], const double2_t* __restrict__ col, int r, int last, int ld2n,
                                        bool live0, bool live1) {
    const double2_t zero2 = {0.0, 0.0};
    if (live0) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t row = min(r + u, last);
            if constexpr (NT) a[u][0] = __builtin_nontemporal_load(col + row * ld2n);
            else a[u][0] = col[row * ld2n];
        }
    } else {   // (lanes past the last column: defined values, never stored)
#pragma unroll
        for (int u = 0; u < 8; u++) a[u][0] = zero2;
    }
    if (live1) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t row = min(r + u, last);
            if constexpr (NT) a[u][1] = __builtin_nontemporal_load(col + row * ld2n + 64);
            else a[u][1] = col[row * ld2n + 64];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; u++) a[u][1] = zero2;
    }
}
template <bool NT>
__device__ __forceinline__ void fl_store(const double2_t (&a)[8][2], double2_t* __restrict__ col, int r, int last, int ld2n,
                                         bool live0, bool live1) {
    if (live0) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (r + u > last) break;   // wave-uniform
            if constexpr (NT) __builtin_nontemporal_store(a[u][0], col + (size_t)(r + u) * ld2n);
            else col[(size_t)(r + u) * ld2n] = a[u][0];
        }
    }
    if (live1) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (r + u > last) break;
            if constexpr (NT) __builtin_nontemporal_store(a[u][1], col + (size_t)(r + u) * ld2n + 64);
            else col[(size_t)(r + u) * ld2n + 64] = a[u][1];
        }
    }
}
// a(u, .) -= sum over VEC vectors of U[j](r+u) V[j](c): one batch of scalar loads, 2 VEC ds_read_b128, 32 VEC FMAs
template <int VEC>
__device__ __forceinline__ void fl_batch(double2_t (&a)[8][2], const double* __restrict__ Ur, int ld,
                                         const double2_t* __restrict__ shv) {
    double k[VEC][8];
    double2_t v[VEC][2];
#pragma unroll
    for (int q = 0; q < VEC; q++) {
#pragma unroll
        for (int u = 0; u < 8; u++) k[q][u] = Ur[(size_t)q * ld + u];   // wave-uniform address -> s_load_dwordx16
        v[q][0] = shv[q * kStripCols2];
        v[q][1] = shv[q * kStripCols2 + 64];
    }
    // A scheduling barrier keeps the batch's s_load_dwordx16 together in front of the FMAs.  (Left alone, hipcc moves
    // each scalar load down to its use to shorten the SGPR live ranges, and the scalar-load latency is exposed once
    // per vector instead of once per batch.)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < VEC; q++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a[u][0].x = __builtin_fma(-k[q][u], v[q][0].x, a[u][0].x);
            a[u][0].y = __builtin_fma(-k[q][u], v[q][0].y, a[u][0].y);
            a[u][1].x = __builtin_fma(-k[q][u], v[q][1].x, a[u][1].x);
            a[u][1].y = __builtin_fma(-k[q][u], v[q][1].y, a[u][1].y);
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(64 * kStripWaves, 1) void k_flush_strip(double* __restrict__ sigma,
                                                                     const double* __restrict__ Uall,
                                                                     const double* __restrict__ Vall, int N, int ld,
                                                                     size_t sigma_stride, int cap, int count,
                                                                     int rows_per_block, int strips, int row_blocks, int B,
                                                                     PanelIO pio, int n) {
    extern __shared__ double2_t sh_V[];   // [count][kStripCols2]
    const int P = strips * row_blocks;
    int b, p;
    const int full = (B / 8) * 8 * P;     // (a filter's workgroups share one XCD's L2: see k_flush)
    if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        p = slot % P;
    } else {
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        p = rest % P;
    }
    const int ld2n = ld >> 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cbase = (p % strips) * kStripCols2 + lane;
    const int nc2 = (N + 1) >> 1;   // double2 columns that hold the matrix: the padding behind them (up to 1/32 of a row) stays zero untouched
    const bool live0 = cbase < nc2, live1 = cbase + 64 < nc2;
    const int row_begin = (p / strips) * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);
    const double* __restrict__ Ub = Uall + (size_t)b * cap * ld;
    const double2_t* __restrict__ Vb = reinterpret_cast<const double2_t*>(Vall + (size_t)b * cap * ld);
    double2_t* __restrict__ col = reinterpret_cast<double2_t*>(sigma + (size_t)b * sigma_stride) + cbase;

    // 8-row groups, the last one possibly partial: its missing rows re-read the last row and are never stored (their
    // scalar operands come from the padding of the U rows: r + 7 < ld).  Wave w takes the groups w, w + 16, ...
    const int ngroups = (row_end - row_begin + 7) >> 3;
    int g = wave;
    double2_t a[8][2];
    // column panel (see k_flush): at most four of a lane's 32 values per group are panel entries
    const double* __restrict__ pin = pio.in ? pio.in + (size_t)b * pio.rows * ld : nullptr;
    double* __restrict__ pout = pio.out ? pio.out + (size_t)b * pio.rows * ld : nullptr;
    const bool fix1 = pin && cbase == 0, fix2 = pin && cbase == 1;
    int prow[4] = {-1, -1, -1, -1};
    if (pout) {
        const short* lms = pio.lmslot + (size_t)b * n;
        if (live0) { prow[0] = panel_row_of(2 * cbase, N, lms); prow[1] = panel_row_of(2 * cbase + 1, N, lms); }
        if (live1) { prow[2] = panel_row_of(2 * cbase + 128, N, lms); prow[3] = panel_row_of(2 * cbase + 129, N, lms); }
    }
    const bool any_out = (prow[0] & prow[1] & prow[2] & prow[3]) >= 0;   // (some row index is not -1)
    auto panel_in = [&](int r0) {
        if (fix1) for (int u = 0; u < 8; u++) a[u][0].y = pin[(size_t)ld + min(r0 + u, row_end - 1)];
        if (fix2) for (int u = 0; u < 8; u++) a[u][0].x = pin[(size_t)2 * ld + min(r0 + u, row_end - 1)];
    };
    auto panel_out = [&](int r0) {
        if (!any_out) return;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (prow[q] < 0) continue;
            double* dst = pout + (size_t)prow[q] * ld + r0;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (r0 + u < row_end) dst[u] = (q & 1) ? a[u][q >> 1].y : a[u][q >> 1].x;
        }
    };
    // the first group's loads go out BEFORE the V strip is staged: the two latencies overlap (a workgroup that owns a
    // few hundred rows only would otherwise spend ~10 % of its life in front of the barrier)
    if (g < ngroups) { fl_load<NT>(a, col, row_begin + 8 * g, row_end - 1, ld2n, live0, live1); panel_in(row_begin + 8 * g); }

    const double2_t zero2 = {0.0, 0.0};
    {   // V strip -> LDS: wave w stages the vectors w, w + 16, ... (up to kStripMaxVec / 16 = 5), their loads in flight together
        constexpr int kPerWave = kStripMaxVec / kStripWaves;
        double2_t v[kPerWave][2];
#pragma unroll
        for (int q = 0; q < kPerWave; q++) {
            const int j = min(wave + kStripWaves * q, count - 1);
            v[q][0] = live0 ? Vb[(size_t)j * ld2n + cbase] : zero2;
            v[q][1] = live1 ? Vb[(size_t)j * ld2n + cbase + 64] : zero2;
        }
#pragma unroll
        for (int q = 0; q < kPerWave; q++) {
            const int j = wave + kStripWaves * q;
            if (j < count) {
                sh_V[j * kStripCols2 + lane] = v[q][0];
                sh_V[j * kStripCols2 + 64 + lane] = v[q][1];
            }
        }
    }
    __syncthreads();
    const double2_t* __restrict__ shv = sh_V + lane;

    while (g < ngroups) {
        const int r = row_begin + 8 * g;
        int j = 0;
        for (; j + 4 <= count; j += 4) fl_batch<4>(a, Ub + (size_t)j * ld + r, ld, shv + j * kStripCols2);
        if (j < count) fl_batch<2>(a, Ub + (size_t)j * ld + r, ld, shv + j * kStripCols2);   // (count is even)
        // An explicit vmcnt(0): nothing is outstanding here but the previous group's stores, a whole FMA phase old.
        // Without it hipcc keeps treating the group's loads as possibly pending (the zero-trip path of the loop above)
        // and puts an `s_waitcnt vmcnt(7)` in front of every store: at most 8 stores of a wave in flight, a stall for
        // a write acknowledgement before the next group's loads can be issued.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        fl_store<NT>(a, col, r, row_end - 1, ld2n, live0, live1);
        panel_out(r);
        g += kStripWaves;
        if (g < ngroups) { fl_load<NT>(a, col, row_begin + 8 * g, row_end - 1, ld2n, live0, live1); panel_in(row_begin + 8 * g); }
    }
}

// ---------------------------------------------------------------------------------------------
// Mirrored flush (the symmetric option of ekf_set_update_mode): Sigma_base -= sum_j U[j] V[j]^T is formed for the
// tiles ON AND ABOVE the diagonal only and every tile above it is written twice, in place and mirrored -- 4 N^2 bytes
// read + 8 N^2 written instead of 8 + 8, and half of the multiply-adds.  The elements on and above the diagonal carry
// the same fused multiply-adds in the same order as k_flush's (bit-identical there); below it Sigma_base becomes their
// mirror image, which is what the symmetric gain step (k_gain_delayed<true>) reads anyway.
//   * Tile = 32 rows x 256 columns, one workgroup of 4 waves (four workgroups per CU: their load, multiply-add and
//     store phases interleave; 64-row tiles with 8 waves were 11 % slower at 64 vectors); wave w owns rows 8 w ..
//     8 w + 7, lane l the double2 columns l and l + 64 (1-KB row segments per instruction, 8 rows x 4 columns per lane:
//     k_flush_strip's register tile and its fl_batch).  Row tile ti's column groups start at its diagonal square:
//     group g covers columns 32 ti + 256 g ..; the square (the first 32 columns of group 0) is formed whole and
//     written in place only.
//   * U values are wave-uniform scalar operands (8 rows x 4 vectors per batch = 128 FMAs per lane), V comes through
//     LDS in double-buffered chunks of 8 vectors (wave w stages the vectors w and w + 4 of a chunk).
//   * The mirrored copy goes through LDS in four 64-column transposes, so that a store instruction covers four 256-B
//     row segments of the mirrored tile.
// grid: B x (tiles per filter), XCD-aware decode as in k_flush.
// ---------------------------------------------------------------------------------------------
constexpr int kSymCols = 256;
static_assert(kSymSquare == 32, "k_predict keeps columns 1, 2 up inside the first diagonal square");
constexpr int kSymWaves = kSymSquare / 8;         // measured: 8 waves (64-row tiles) 41.0 ms, 4 waves 36.9 ms, 2 waves 40.7 ms at 64 vectors
constexpr int kSymMinDim = 256;     // below this the plain full flush is used (a handful of tiles per filter)

__host__ __device__ inline int sym_groups(int ti, int ld, int tile_rows) {
    return (ld - ti * tile_rows + kSymCols - 1) / kSymCols;
}

template <bool NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_flush_sym(double* __restrict__ sigma, const double* __restrict__ Uall,
                                                          const double* __restrict__ Vall, int N, int ld,
                                                          size_t sigma_stride, int cap, int count, int P, int B) {
    constexpr int TR = 8 * WAVES;          // tile rows
    constexpr int LS = TR + 2;             // row pitch of the transposes (doubles; even: 16-B aligned read-back)
    constexpr int CH = WAVES >= 4 ? 8 : 4; // vectors per V chunk
    constexpr int VPW = CH / WAVES;        // vectors of a chunk a wave stages
    constexpr int SH = 64 * LS > 2 * CH * 256 ? 64 * LS : 2 * CH * 256;
    __shared__ double sh[SH];              // V chunks: 2 x CH x 128 double2; transposes: 64 x LS doubles
    int b, p;
    const int full = (B / 8) * 8 * P;
    if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        p = slot % P;
    } else {
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        p = rest % P;
    }
    int ti = 0;
    for (;; ti++) {
        const int ng = sym_groups(ti, ld, TR);
        if (p < ng) break;
        p -= ng;
    }
    const int g = p;
    const int ld2n = ld >> 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = ti * TR + 8 * wave;                             // the wave's first row
    const int cfirst = ti * TR + g * kSymCols;                     // the tile's first column
    const int c2 = (cfirst >> 1) + lane;                           // the lane's first double2 column (second: + 64)
    const bool live0 = c2 < ((N + 1) >> 1), live1 = c2 + 64 < ((N + 1) >> 1);   // (padding columns: see k_flush_strip)
    const double* __restrict__ Ub = Uall + (size_t)b * cap * ld;
    const double2_t* __restrict__ Vb = reinterpret_cast<const double2_t*>(Vall + (size_t)b * cap * ld) + c2;
    double* __restrict__ Sg = sigma + (size_t)b * sigma_stride;
    double2_t* __restrict__ col = reinterpret_cast<double2_t*>(Sg) + c2;
    const double2_t zero2 = {0.0, 0.0};
    const int rlast = N - 1;

    double2_t a[8][2];
    fl_load<NT>(a, col, r0, rlast, ld2n, live0, live1);   // (a wave whose rows all lie past the matrix re-reads the last row)

    // V chunks through LDS: wave w stages the vectors w, w + WAVES, .. of a chunk
    double2_t* shV = reinterpret_cast<double2_t*>(sh);   // [2][CH][128]
    double2_t pv[VPW][2];
    auto v_fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < VPW; q++) {
            const int j = min(j0 + wave + WAVES * q, count - 1);
            pv[q][0] = live0 ? Vb[(size_t)j * ld2n] : zero2;
            pv[q][1] = live1 ? Vb[(size_t)j * ld2n + 64] : zero2;
        }
    };
    auto v_put = [&](int buf) {
#pragma unroll
        for (int q = 0; q < VPW; q++) {
            shV[(buf * CH + wave + WAVES * q) * 128 + lane] = pv[q][0];
            shV[(buf * CH + wave + WAVES * q) * 128 + 64 + lane] = pv[q][1];
        }
    };
    v_fetch(0);
    v_put(0);
    __syncthreads();
    const int rU = min(r0, ld - 8);   // (rows past the matrix: any finite operand, never stored)
    int buf = 0;
    for (int j0 = 0; j0 < count; j0 += CH) {
        const bool more = j0 + CH < count;
        if (more) v_fetch(j0 + CH);
        const int nv = min(CH, count - j0);
        const double2_t* __restrict__ sv = shV + buf * CH * 128 + lane;
        int jj = 0;
        for (; jj + 4 <= nv; jj += 4) fl_batch<4>(a, Ub + (size_t)(j0 + jj) * ld + rU, ld, sv + jj * kStripCols2);
        if (jj < nv) fl_batch<2>(a, Ub + (size_t)(j0 + jj) * ld + rU, ld, sv + jj * kStripCols2);   // (count is even)
        if (more) v_put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // in place
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see k_flush_strip
    if (r0 <= rlast) fl_store<NT>(a, col, r0, rlast, ld2n, live0, live1);
    // mirrored: chunk q = the tile's columns 64 q .. 64 q + 63 (lanes 32 (q & 1) .., second column pair for q >= 2),
    // transposed through LDS: sh[c][r], c = column inside the chunk, r = row inside the tile
    constexpr int LPR = TR / 2;            // lanes per mirrored row segment (TR doubles)
    constexpr int RPI = 64 / LPR;          // mirrored rows per store instruction
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // the diagonal square (the first TR columns of group 0) is written in place only; chunks outside the matrix: nothing
        if ((g == 0 && 64 * (q + 1) <= TR) || cfirst + 64 * q >= N) continue;   // (uniform)
        if ((lane >> 5) == (q & 1)) {
            const int cl = 2 * (lane & 31);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double2_t v = q < 2 ? a[u][0] : a[u][1];
                sh[cl * LS + 8 * wave + u] = v.x;
                sh[(cl + 1) * LS + 8 * wave + u] = v.y;
            }
        }
        __syncthreads();
        // mirrored rows cfirst + 64 q + c, columns TR ti .. + TR - 1: LPR lanes per row, RPI rows per instruction
#pragma unroll
        for (int it = 0; it < 64 / (WAVES * RPI); it++) {
            const int c = WAVES * RPI * it + RPI * wave + lane / LPR;
            const int tr = cfirst + 64 * q + c, tc = ti * TR + 2 * (lane % LPR);
            if (tr < N && tc < N && !(g == 0 && 64 * q + c < TR)) {
                const double2_t v = *reinterpret_cast<const double2_t*>(sh + c * LS + 2 * (lane % LPR));
                double2_t* dst = reinterpret_cast<double2_t*>(Sg + (size_t)tr * ld + tc);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
        __syncthreads();
    }
}

void launch_gain_delayed(const PoolView& pv, const CmdSrc& src, const Pending& pend, double* state_out,
                         hipStream_t s) {
    const dim3 grid((unsigned)((long long)((pv.ld / 2 + 255) / 256) * pv.B));
    if (pend.symmetric) hipLaunchKernelGGL(k_gain_delayed<true>, grid, dim3(256), 0, s, pv, src, pend, state_out);
    else hipLaunchKernelGGL(k_gain_delayed<false>, grid, dim3(256), 0, s, pv, src, pend, state_out);
}

void launch_gain_delayed_pair(const PoolView& pv, const CmdSrc& src, const Pending& pend, double* state_out,
                              hipStream_t s) {
    const dim3 grid((unsigned)((long long)((pv.ld / 2 + 255) / 256) * pv.B));
    if (pend.symmetric) hipLaunchKernelGGL(k_gain_delayed_pair<true>, grid, dim3(256), 0, s, pv, src, pend, state_out);
    else hipLaunchKernelGGL(k_gain_delayed_pair<false>, grid, dim3(256), 0, s, pv, src, pend, state_out);
}

// dynamic LDS beyond 64 KB has to be allowed per kernel and per device: once each, whichever host thread comes first
static bool strip_flush_allowed() {
    constexpr int kMaxDev = 64;
    static std::once_flag once[kMaxDev];
    static bool ok[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return false;
    std::call_once(once[dev], [dev]() {
        const int bytes = kStripMaxVec * kStripCols2 * (int)sizeof(double2_t);
        ok[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flush_strip<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flush_strip<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    });
    return ok[dev];
}

// Columns 1 and 2 below the first diagonal square from the rows 1 and 2: ends a run whose last predictions (symmetric
// == 2: rows only) were not followed by a mirrored flush.  grid (ceil(N / 256), B).
__global__ __launch_bounds__(256) void k_sym_repair(double* __restrict__ sigma, int N, int ld, size_t sigma_stride) {
    const int k = kSymSquare + blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    double* Sg = sigma + (size_t)blockIdx.y * sigma_stride;
    Sg[(size_t)k * ld + 1] = Sg[(size_t)1 * ld + k];
    Sg[(size_t)k * ld + 2] = Sg[(size_t)2 * ld + k];
}

void launch_sym_repair(const PoolView& pv, hipStream_t s) {
    if (pv.N <= kSymSquare) return;
    hipLaunchKernelGGL(k_sym_repair, dim3((pv.N - kSymSquare + 255) / 256, pv.B), dim3(256), 0, s, pv.sigma, pv.N, pv.ld,
                       pv.sigma_stride);
}

bool sym_flush_applies(const PoolView& pv, const Pending& pend, const Rank2Tuning& t) {
    return pend.symmetric && pv.N >= kSymMinDim && t.rows_per_block == 0;
}

// ---------------------------------------------------------------------------------------------
// Column panel: the plan (which landmarks' columns the next flush writes out as rows) and the repair.
// One thread per filter walks the log slots of the next `nsteps` steps in order and gives the first `slots` distinct
// landmarks a slot each; the previous plan's entries of lmslot are cleared from plan_list first.  Sequential per filter,
// a few dozen entries: microseconds.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_panel_plan(int B, int n, const int* __restrict__ lm_idx, int nsteps, int vmax,
                                                   int slots, short* __restrict__ lmslot, int* __restrict__ plan_list,
                                                   int* __restrict__ curv_reset, double* __restrict__ apred_reset) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    if (curv_reset) {   // the flush behind this launch folds everything: no current row / column outlives it
        for (int q = 0; q <= slots; q++) curv_reset[(size_t)b * (1 + slots) + q] = -1;
        apred_reset[(size_t)b * 2] = 0.0; apred_reset[(size_t)b * 2 + 1] = 0.0;
    }
    short* ls = lmslot + (size_t)b * n;
    int* pl = plan_list + (size_t)b * slots;
    for (int q = 0; q < slots; q++) {
        const int lm = pl[q];
        if (lm >= 0 && lm < n) ls[lm] = -1;
        pl[q] = -1;
    }
    int used = 0;
    for (int t = 0; t < nsteps && used < slots; t++)
        for (int v = 0; v < vmax && used < slots; v++) {
            const int lm = lm_idx[((size_t)t * B + b) * vmax + v];
            if (lm < 0 || lm >= n || ls[lm] >= 0) continue;
            ls[lm] = (short)used;
            pl[used++] = lm;
        }
}

void launch_panel_plan(const PoolView& pv, const int* lm_idx, int nsteps, int vmax, int slots, short* lmslot, int* plan_list,
                       hipStream_t s, int* curv_reset, double* apred_reset) {
    hipLaunchKernelGGL(k_panel_plan, dim3((pv.B + 63) / 64), dim3(64), 0, s, pv.B, pv.n, lm_idx, nsteps, vmax, slots, lmslot,
                       plan_list, curv_reset, apred_reset);
}

// matrix columns 1, 2 <- panel rows 1, 2 (the strided writes prediction() skipped while the panel was on).  grid (ceil(N / 256), B)
__global__ __launch_bounds__(256) void k_panel_repair(double* __restrict__ sigma, const double* __restrict__ colp, int colp_rows,
                                                      int N, int ld, size_t sigma_stride) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    double* Sg = sigma + (size_t)blockIdx.y * sigma_stride;
    const double* cp = colp + (size_t)blockIdx.y * colp_rows * ld;
    Sg[(size_t)k * ld + 1] = cp[(size_t)ld + k];
    Sg[(size_t)k * ld + 2] = cp[(size_t)2 * ld + k];
}

void launch_panel_repair(const PoolView& pv, const double* colp, int colp_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_panel_repair, dim3((pv.N + 255) / 256, pv.B), dim3(256), 0, s, pv.sigma, colp, colp_rows, pv.N, pv.ld,
                       pv.sigma_stride);
}

int launch_flush(const PoolView& pv, const Pending& pend, const Rank2Tuning& t, hipStream_t s, const PanelIO& panel) {
    if (pend.count <= 0) return 0;
    const size_t pool_bytes = (size_t)pv.B * pv.sigma_stride * sizeof(double);
    const bool nt = t.nontemporal >= 0 ? t.nontemporal != 0 : pool_bytes > ((size_t)192 << 20);
    if (sym_flush_applies(pv, pend, t)) {
        constexpr int tr = 8 * kSymWaves;
        int P = 0;
        for (int ti = 0; ti * tr < pv.N; ti++) P += sym_groups(ti, pv.ld, tr);
        dim3 grid((unsigned)((long long)P * pv.B));
#define EKF_FS_ARGS pv.sigma, pend.U, pend.V, pv.N, pv.ld, pv.sigma_stride, pend.cap, pend.count, P, pv.B
        if (nt) hipLaunchKernelGGL((k_flush_sym<true, kSymWaves>), grid, dim3(64 * kSymWaves), 0, s, EKF_FS_ARGS);
        else hipLaunchKernelGGL((k_flush_sym<false, kSymWaves>), grid, dim3(64 * kSymWaves), 0, s, EKF_FS_ARGS);
#undef EKF_FS_ARGS
        return 6;
    }
    // Strip form: the V strip in LDS (count x 2 KB, one workgroup of 16 waves per CU) and >= 2 workgroups per CU of
    // work.  Its time hardly depends on the count, the plain form's does (the stream floor up to 32 vectors, 1.4x at
    // 64: tools/flush_sweep.py) -- the strip form takes over beyond kStripFromCount vectors on pools that fill the chip.
    const bool forced = t.strip_flush == 2;   // (EKF_FORM_STRIP_FLUSH_ALWAYS: pools of any size, any count <= 80)
    if ((forced || (t.strip_flush == 1 && t.rows_per_block == 0 && pv.N >= 256 && pend.count > kStripFromCount)) &&
        pend.count <= kStripMaxVec) {
        const int strips = (pv.ld / 2 + kStripCols2 - 1) / kStripCols2;
        int row_blocks = 1;
        while ((long long)pv.B * strips * row_blocks < 1024 && pv.N / (row_blocks * 2) >= 512) row_blocks *= 2;
        if ((forced || (long long)pv.B * strips * row_blocks >= kStripMinWorkgroups) && strip_flush_allowed()) {
            const int rows = (((pv.N + row_blocks - 1) / row_blocks) + 7) & ~7;
            row_blocks = (pv.N + rows - 1) / rows;
            dim3 grid((unsigned)((long long)strips * row_blocks * pv.B));
            const size_t lds = (size_t)pend.count * kStripCols2 * sizeof(double2_t);
            if (nt) hipLaunchKernelGGL((k_flush_strip<true>), grid, dim3(64 * kStripWaves), lds, s, pv.sigma, pend.U,
                                       pend.V, pv.N, pv.ld, pv.sigma_stride, pend.cap, pend.count, rows, strips,
                                       row_blocks, pv.B, panel, pv.n);
            else hipLaunchKernelGGL((k_flush_strip<false>), grid, dim3(64 * kStripWaves), lds, s, pv.sigma, pend.U,
                                    pend.V, pv.N, pv.ld, pv.sigma_stride, pend.cap, pend.count, rows, strips,
                                    row_blocks, pv.B, panel, pv.n);
            return 1;
        }
    }
    int rows = t.rows_per_block > 0 ? t.rows_per_block : 16;  // measured: tools/flush_sweep.py
    const long long all_strips = (long long)pv.B * ((pv.ld / 2 + 255) / 256);
    if (t.rows_per_block <= 0 && all_strips * pv.N < 256LL * 8 * 32) rows = 8;
    const int strips = (pv.ld / 2 + 255) / 256, row_blocks = (pv.N + rows - 1) / rows;
    dim3 grid((unsigned)((long long)strips * row_blocks * pv.B));
#define EKF_FL_ARGS pv.sigma, pend.U, pend.V, pv.N, pv.ld, pv.sigma_stride, pend.cap, pend.count, rows, strips, \
    row_blocks, pv.B, panel, pv.n
    if (rows >= 16) {
        if (nt) hipLaunchKernelGGL((k_flush<16, true>), grid, dim3(256), 0, s, EKF_FL_ARGS);
        else hipLaunchKernelGGL((k_flush<16, false>), grid, dim3(256), 0, s, EKF_FL_ARGS);
    } else {
        if (nt) hipLaunchKernelGGL((k_flush<8, true>), grid, dim3(256), 0, s, EKF_FL_ARGS);
        else hipLaunchKernelGGL((k_flush<8, false>), grid, dim3(256), 0, s, EKF_FL_ARGS);
    }
#undef EKF_FL_ARGS
    return 0;
}

int max_pending() { return kMaxPending; }

}  // namespace ekf
