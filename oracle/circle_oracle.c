/*
 * oracle/circle_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of rigid2d::CircleFitting (rigid2d/src/circle_fitting.cpp), the
 * perception front end that turns a 360-beam laser scan into the (x, y) measurements fed to
 * EKF_SLAM::data_association (nuslam/src/landmarks.cpp:141 -> unknown_data_assoc.cpp:309-320).
 * SURVEY.md section 8(f) row f3.
 *
 * PARITY STATUS: PINNED on the reference's own known-answer tests (nuslam/tests/circle_tests.cpp:8-76:
 * clustering, two circle regressions, classification) -- tests/test_circle_oracle.py.  The reference's
 * Armadillo calls svd / eig_gen / solve (circle_fitting.cpp:168,184,211) are restated with Jacobi
 * iterations (one-sided Hestenes SVD of the n x 4 design matrix, two-sided Jacobi for the symmetric
 * 4 x 4 eigenproblem, Y^-1 = V diag(1/s) V^T for the solve); oracle/np_restatement.py holds a literal
 * LAPACK-backed transcription (numpy.linalg.svd / eig / solve) as the second opinion.
 *
 * Reference quirks kept on purpose:
 *  - the last beam always closes the running cluster and is itself dropped (:31,:38-42);
 *  - clusters need MORE than 6 points (:34);
 *  - the wrap-around test compares the first kept cluster with the last kept cluster wherever they
 *    lie in the scan (:54-70); with exactly one kept cluster whose two ends are within the threshold
 *    the cluster is prepended to itself and then popped, leaving NO cluster;
 *  - no kept cluster at all is undefined behaviour in the reference (:54 indexes an empty vector);
 *    here it yields zero circles.
 */
#include <math.h>
#include <string.h>

#define CF_PI 3.14159265358979323846 /* rigid2d.hpp:13 */
#define CF_MAX_BEAMS 2048
#define CF_MAX_CLUSTERS 300

/* rigid2d/src/rigid2d.cpp:336-345 */
static double cf_normalize_angle(double rad) {
    double reduced_ang = fmod(rad, (2 * CF_PI));
    double ang = fmod((reduced_ang + (2 * CF_PI)), (2 * CF_PI));
    if (ang > CF_PI) ang = ang - (2 * CF_PI);
    return ang;
}

typedef struct {
    int n;                       /* points in the cluster */
    int seg_start[2], seg_len[2];/* beams: seg 0 then seg 1 (seg 1 only after a wrap-around merge) */
} cf_cluster;

/* circle_fitting.cpp:11-90.  Returns the number of kept clusters. */
int cf_clustering(const double *ranges, int num_readings, cf_cluster *out, int max_out) {
    const double thres = 0.2;                                   /* :17 */
    int nc = 0;
    int cur_start = 0, cur_len = 1;                             /* :23 curr_cluster = {ranges[0]} */
    if (num_readings < 1) return 0;
    for (int i = 1; i < num_readings; i++) {                    /* :30 */
        if ((fabs(ranges[i] - ranges[i - 1]) < thres) && (i != (num_readings - 1))) {
        } else {
            if (cur_len > 6 && nc < max_out) {                  /* :34 */
                out[nc].n = cur_len;
                out[nc].seg_start[0] = cur_start; out[nc].seg_len[0] = cur_len;
                out[nc].seg_start[1] = 0;         out[nc].seg_len[1] = 0;
                nc++;
            }
            cur_start = i; cur_len = 0;                         /* :38-39 */
        }
        cur_len++;                                              /* :42 */
    }
    if (nc == 0) return 0;                                      /* reference: UB at :54 */
    /* :54-70 wrap-around check between the first and the last KEPT cluster */
    double first_elem_of_first = ranges[out[0].seg_start[0]];
    const cf_cluster last = out[nc - 1];
    double last_elem_of_last = ranges[last.seg_start[0] + last.seg_len[0] - 1];
    if (fabs(first_elem_of_first - last_elem_of_last) < thres) {
        if (nc == 1) return 0;   /* prepended to itself (:63-66), then popped (:68-69) */
        /* last cluster's points, in order, go in front of the first cluster's points */
        out[0].seg_start[1] = out[0].seg_start[0]; out[0].seg_len[1] = out[0].seg_len[0];
        out[0].seg_start[0] = last.seg_start[0];   out[0].seg_len[0] = last.seg_len[0];
        out[0].n = out[0].seg_len[0] + out[0].seg_len[1];
        nc--;
    }
    return nc;
}

/* Cartesian coordinates of beam i (:25-28, :44-47) */
static void cf_beam_xy(const double *ranges, int num_readings, int i, double *x, double *y) {
    double angle_resolution = 2 * CF_PI / (double)num_readings;
    if (i == 0) { *x = ranges[0] * cos(0.0); *y = ranges[0] * sin(0.0); return; }
    *x = ranges[i] * cos(cf_normalize_angle(i * angle_resolution));
    *y = ranges[i] * sin(cf_normalize_angle(i * angle_resolution));
}

/* One-sided Jacobi (Hestenes) SVD of Z (n x 4, row-major): on return the columns of Z are U*diag(s),
 * V (4x4, row-major) holds the right singular vectors, s is sorted DESCENDING like arma::svd. */
static void cf_svd4(double *Z, int n, double s[4], double V[16]) {
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int k = 0; k < n; k++) {
                    double zp = Z[4 * k + p], zq = Z[4 * k + q];
                    alpha += zp * zp; beta += zq * zq; gamma += zp * zq;
                }
                if (gamma == 0.0) continue;
                double lim = sqrt(alpha * beta);
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * lim) continue;
                if (fabs(gamma) > off) off = fabs(gamma) / (lim > 0 ? lim : 1.0);
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < n; k++) {
                    double zp = Z[4 * k + p], zq = Z[4 * k + q];
                    Z[4 * k + p] = c * zp - sn * zq;
                    Z[4 * k + q] = sn * zp + c * zq;
                }
                for (int k = 0; k < 4; k++) {
                    double vp = V[4 * k + p], vq = V[4 * k + q];
                    V[4 * k + p] = c * vp - sn * vq;
                    V[4 * k + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < 4; j++) {
        double a = 0.0;
        for (int k = 0; k < n; k++) a += Z[4 * k + j] * Z[4 * k + j];
        s[j] = sqrt(a);
    }
    for (int i = 0; i < 3; i++)          /* sort descending, permuting the columns of V */
        for (int j = i + 1; j < 4; j++)
            if (s[j] > s[i]) {
                double t = s[i]; s[i] = s[j]; s[j] = t;
                for (int k = 0; k < 4; k++) { double v = V[4 * k + i]; V[4 * k + i] = V[4 * k + j]; V[4 * k + j] = v; }
            }
}

/* Two-sided Jacobi for a symmetric 4x4: eigenvalues w, eigenvectors in the columns of E. */
static void cf_eig4_sym(double A[16], double w[4], double E[16]) {
    for (int i = 0; i < 16; i++) E[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 4; i++) {
            diag += A[5 * i] * A[5 * i];
            for (int j = i + 1; j < 4; j++) off += A[4 * i + j] * A[4 * i + j];
        }
        if (off <= 1e-34 * diag || off == 0.0) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double apq = A[4 * p + q];
                if (apq == 0.0) continue;
                double theta = (A[5 * q] - A[5 * p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 4; k++) {          /* A <- A J */
                    double akp = A[4 * k + p], akq = A[4 * k + q];
                    A[4 * k + p] = c * akp - sn * akq;
                    A[4 * k + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {          /* A <- J^T A */
                    double apk = A[4 * p + k], aqk = A[4 * q + k];
                    A[4 * p + k] = c * apk - sn * aqk;
                    A[4 * q + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    double ekp = E[4 * k + p], ekq = E[4 * k + q];
                    E[4 * k + p] = c * ekp - sn * ekq;
                    E[4 * k + q] = sn * ekp + c * ekq;
                }
            }
    }
    for (int i = 0; i < 4; i++) w[i] = A[5 * i];
}

/* circle_fitting.cpp:104-232 for ONE cluster of m points (xs, ys).  out = {centre x, centre y, radius}. */
void cf_regress(const double *xs, const double *ys, int m, double out[3]) {
    double Z[4 * CF_MAX_BEAMS];
    double x_sum = 0.0, y_sum = 0.0;
    if (m > CF_MAX_BEAMS) m = CF_MAX_BEAMS;
    for (int k = 0; k < m; k++) { x_sum += xs[k]; y_sum += ys[k]; }      /* :112-117 */
    double x_mean = x_sum / (double)m, y_mean = y_sum / (double)m;      /* :119-120 */
    double z_sum = 0.0;
    for (int j = 0; j < m; j++) {                                        /* :124-141 */
        double x = xs[j] - x_mean, y = ys[j] - y_mean;
        double zi = pow(x, 2.0) + pow(y, 2.0);
        z_sum += zi;
        Z[4 * j + 0] = zi; Z[4 * j + 1] = x; Z[4 * j + 2] = y; Z[4 * j + 3] = 1.0;
    }
    double z_mean = z_sum / (double)m;                                   /* :131 */
    double s[4], V[16], A[4];
    cf_svd4(Z, m, s, V);                                                 /* :168 */
    if (s[3] < 1e-12) {                                                  /* :171-175 */
        for (int k = 0; k < 4; k++) A[k] = V[4 * k + 3];
    } else {
        double Y[16], Hinv[16], T[16], Q[16], w[4], E[16];
        for (int i = 0; i < 4; i++)                                      /* :177 Y = V diag(s) V^T */
            for (int j = 0; j < 4; j++) {
                double a = 0.0;
                for (int k = 0; k < 4; k++) a += V[4 * i + k] * s[k] * V[4 * j + k];
                Y[4 * i + j] = a;
            }
        memset(Hinv, 0, sizeof(Hinv));                                   /* :156-161 */
        Hinv[0 * 4 + 3] = 0.5; Hinv[1 * 4 + 1] = 1.0; Hinv[2 * 4 + 2] = 1.0; Hinv[3 * 4 + 0] = 0.5;
        Hinv[3 * 4 + 3] = -2.0 * z_mean;
        for (int i = 0; i < 4; i++)                                      /* :178 Q = Y Hinv Y */
            for (int j = 0; j < 4; j++) {
                double a = 0.0;
                for (int k = 0; k < 4; k++) a += Y[4 * i + k] * Hinv[4 * k + j];
                T[4 * i + j] = a;
            }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double a = 0.0;
                for (int k = 0; k < 4; k++) a += T[4 * i + k] * Y[4 * k + j];
                Q[4 * i + j] = a;
            }
        for (int i = 0; i < 4; i++)                                      /* Q is symmetric up to rounding */
            for (int j = i + 1; j < 4; j++) { double a = 0.5 * (Q[4 * i + j] + Q[4 * j + i]); Q[4 * i + j] = a; Q[4 * j + i] = a; }
        cf_eig4_sym(Q, w, E);                                            /* :184 */
        int smallest_eig_index = 0;                                      /* :187-197 */
        double smallest_eig_val = 1000.0;
        for (int e = 0; e < 4; e++)
            if (w[e] > 0 && w[e] < smallest_eig_val) { smallest_eig_val = w[e]; smallest_eig_index = e; }
        double As[4], tmp[4];
        for (int k = 0; k < 4; k++) As[k] = E[4 * k + smallest_eig_index];
        /* :211 A = solve(Y, A_star) with Y^-1 = V diag(1/s) V^T */
        for (int k = 0; k < 4; k++) {
            double a = 0.0;
            for (int i = 0; i < 4; i++) a += V[4 * i + k] * As[i];
            tmp[k] = a / s[k];
        }
        for (int i = 0; i < 4; i++) {
            double a = 0.0;
            for (int k = 0; k < 4; k++) a += V[4 * i + k] * tmp[k];
            A[i] = a;
        }
    }
    double a = -A[1] / (2 * A[0]);                                       /* :220-222 */
    double b = -A[2] / (2 * A[0]);
    double R_sqr = (pow(A[1], 2.0) + pow(A[2], 2.0) - 4 * A[0] * A[3]) / (4 * pow(A[0], 2.0));
    out[0] = a + x_mean;                                                 /* :224 */
    out[1] = b + y_mean;
    out[2] = sqrt(R_sqr);                                                /* :228 */
}

/* circle_fitting.cpp:234-296 for one cluster: mean inscribed angle + radius test. */
int cf_is_circle(const double *xs, const double *ys, int m, double radius) {
    double p1x = xs[0], p1y = ys[0], p2x = xs[m - 1], p2y = ys[m - 1];   /* :244-245 */
    double sum_angle = 0.0;
    for (int k = 1; k < m - 1; k++) {                                    /* :248-261 */
        double pp1x = p1x - xs[k], pp1y = p1y - ys[k], pp2x = p2x - xs[k], pp2y = p2y - ys[k];
        double top_part = pp1x * pp2x + pp1y * pp2y;
        double bot_part = sqrt(pow(pp1x, 2.0) + pow(pp1y, 2.0)) * sqrt(pow(pp2x, 2.0) + pow(pp2y, 2.0));
        sum_angle += acos(top_part / bot_part);
    }
    double mean_angle = sum_angle / (m - 2);                             /* :263 */
    return (mean_angle > 1.5708 && mean_angle < 2.3562 && radius < 0.2) ? 1 : 0;  /* :264-271 */
}

/* gathers the cluster's points (seg 0 then seg 1) */
static int cf_points(const double *ranges, int n, const cf_cluster *c, double *xs, double *ys) {
    int m = 0;
    for (int sg = 0; sg < 2; sg++)
        for (int k = 0; k < c->seg_len[sg]; k++) { cf_beam_xy(ranges, n, c->seg_start[sg] + k, &xs[m], &ys[m]); m++; }
    return m;
}

/* circle_fitting.cpp:298-304 approxCirclePositions.  all_out (nullable, [clusters][4]) receives every
 * cluster's {x, y, r, is_circle}; clean_xy ([max_out][2]) and clean_r the classified circles.
 * Returns the number of classified circles; *n_clusters the number of clusters. */
int cf_approx_circle_positions(const double *ranges, int num_readings, int max_out, double *clean_xy,
                               double *clean_r, double *all_out, int *n_clusters) {
    static __thread cf_cluster cl[CF_MAX_CLUSTERS];
    static __thread double xs[CF_MAX_BEAMS], ys[CF_MAX_BEAMS];
    if (num_readings > CF_MAX_BEAMS) num_readings = CF_MAX_BEAMS;
    int nc = cf_clustering(ranges, num_readings, cl, CF_MAX_CLUSTERS);
    int count = 0;
    for (int c = 0; c < nc; c++) {
        int m = cf_points(ranges, num_readings, &cl[c], xs, ys);
        double o[3];
        cf_regress(xs, ys, m, o);
        int ok = cf_is_circle(xs, ys, m, o[2]);
        if (all_out) { all_out[4 * c] = o[0]; all_out[4 * c + 1] = o[1]; all_out[4 * c + 2] = o[2]; all_out[4 * c + 3] = ok; }
        if (ok && count < max_out) {
            clean_xy[2 * count] = o[0]; clean_xy[2 * count + 1] = o[1];
            if (clean_r) clean_r[count] = o[2];
            count++;
        }
    }
    if (n_clusters) *n_clusters = nc;
    return count;
}

/* test doors: cluster sizes / first range of each cluster (get_point_cluster of the reference) */
int cf_cluster_summary(const double *ranges, int num_readings, int max_out, int *sizes, double *first_range) {
    static __thread cf_cluster cl[CF_MAX_CLUSTERS];
    int nc = cf_clustering(ranges, num_readings, cl, CF_MAX_CLUSTERS);
    for (int c = 0; c < nc && c < max_out; c++) {
        sizes[c] = cl[c].n;
        first_range[c] = ranges[cl[c].seg_start[0]];
    }
    return nc;
}
