/*
 * oracle/ekf_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the reference hot path rigid2d::EKF_SLAM
 * (rigid2d/src/ekf_slam.cpp) and of the two rigid2d helpers it calls.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product (libekfslam_hip.so) never does.
 *
 * PARITY STATUS
 *   - normalize_angle / body twist: PINNED.  Checked against the reference's
 *     own KATs (rigid2d/tests/tests.cpp:322-331) and against oracle/_ref, the
 *     reference's rigid2d.cpp + diff_drive.cpp compiled as they lie.
 *   - EKF_SLAM itself: PARITY UNPINNED.  The reference's ekf_slam.cpp needs
 *     Armadillo (ekf_slam.hpp:10; version unpinned, linked as bare `armadillo`
 *     in rigid2d/CMakeLists.txt:179-181), which is absent from this image, so
 *     the reference is unbuildable here; and no reference test or fixture
 *     touches EKF_SLAM (rigid2d/tests/tests.cpp:2-3 include only rigid2d.hpp
 *     and diff_drive.hpp).  This file restates the algorithm twice:
 *       mode 0 "dense literal": every arma expression executed as the dense
 *              matrix product it denotes (N^3 loops, dense 2xN H, dense Q,
 *              dense (I-KH)), in the reference's operand order;
 *       mode 1 "structured": the O(N^2) formulation (5-column gather, rank-2
 *              update) that the HIP kernels also use -- the on-box checker at
 *              large n and the timed CPU baseline ("port").
 *     Both must agree to <=1e-12 per block (tests/test_oracle.py).
 *
 * State order [theta, x, y, m1x, m1y, ...] (ekf_slam.cpp:15-21,72-74).
 * Covariance is stored ROW-major N x N here (Armadillo is column-major; the
 * layout is not observable through the reference API).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* rigid2d/include/rigid2d/rigid2d.hpp:13 */
#define EKFO_PI 3.14159265358979323846

typedef struct {
    double sigma0_landmark; /* ekf_slam.cpp:32   100      */
    double q_pose;          /* ekf_slam.cpp:41-43 1e-4    */
    double r_meas;          /* ekf_slam.cpp:174-175 0.01  */
    double gate_new;        /* ekf_slam.cpp:293  10.0     */
    double gate_update;     /* ekf_slam.cpp:330  1.0      */
    double straight_eps;    /* ekf_slam.cpp:79   1e-6     */
} ekfo_params;

typedef struct {
    int n, N, mode;
    int landmark_init_flag;
    double *state;  /* N               ekf_slam.hpp:61 */
    double *sigma;  /* N*N row-major   ekf_slam.hpp:63 */
    double *Q;      /* N*N dense       ekf_slam.hpp:62 (mode 0 only) */
    double *t1, *t2, *t3; /* N*N scratch (mode 0), 4N scratch (mode 1) */
    ekfo_params p;
} ekfo;

void ekfo_default_params(ekfo_params *p) {
    p->sigma0_landmark = 100.0;
    p->q_pose = 0.0001;
    p->r_meas = 0.01;
    p->gate_new = 10.0;
    p->gate_update = 1.0;
    p->straight_eps = 0.000001;
}

/* rigid2d/src/rigid2d.cpp:336-345 */
double ekfo_normalize_angle(double rad) {
    double reduced_ang = fmod(rad, (2 * EKFO_PI));
    double ang = fmod((reduced_ang + (2 * EKFO_PI)), (2 * EKFO_PI));
    if (ang > EKFO_PI) {
        ang = ang - (2 * EKFO_PI);
    }
    return ang;
}

/* rigid2d/src/diff_drive.cpp:38-47 (input side: wheel deltas -> body twist) */
void ekfo_body_twist(double wheel_base, double wheel_radius, double left_angle,
                     double right_angle, double *out_dtheta_dx) {
    double D = wheel_base * 0.5;
    double r = wheel_radius;
    out_dtheta_dx[0] = (r / (2.0 * D)) * (right_angle - left_angle);
    out_dtheta_dx[1] = (r / 2.0) * (right_angle + left_angle);
}

/* rigid2d/src/ekf_slam.cpp:27-53 */
ekfo *ekfo_create(int n, int mode, const ekfo_params *params) {
    if (n < 0) return NULL;
    ekfo *o = (ekfo *)calloc(1, sizeof(ekfo));
    if (!o) return NULL;
    int N = 3 + 2 * n;
    o->n = n;
    o->N = N;
    o->mode = mode;
    if (params) o->p = *params; else ekfo_default_params(&o->p);
    size_t NN = (size_t)N * N;
    o->state = (double *)calloc(N, sizeof(double));
    o->sigma = (double *)calloc(NN, sizeof(double));
    if (mode == 0) {
        o->Q = (double *)calloc(NN, sizeof(double));
        o->t1 = (double *)calloc(NN > (size_t)8 * N ? NN : (size_t)8 * N, sizeof(double));
        o->t2 = (double *)calloc(NN, sizeof(double));
        o->t3 = (double *)calloc(NN, sizeof(double));
        o->Q[0 * N + 0] = o->p.q_pose;
        o->Q[1 * N + 1] = o->p.q_pose;
        o->Q[2 * N + 2] = o->p.q_pose;
    } else {
        o->t1 = (double *)calloc((size_t)4 * N, sizeof(double));
    }
    for (int i = 3; i < N; i++) o->sigma[(size_t)i * N + i] = 1.0 * o->p.sigma0_landmark;
    o->landmark_init_flag = 0;
    return o;
}

void ekfo_destroy(ekfo *o) {
    if (!o) return;
    free(o->state); free(o->sigma); free(o->Q); free(o->t1); free(o->t2); free(o->t3);
    free(o);
}

int ekfo_dim(const ekfo *o) { return o->N; }
void ekfo_get_state(const ekfo *o, double *out) { memcpy(out, o->state, sizeof(double) * o->N); }
void ekfo_set_state(ekfo *o, const double *in) { memcpy(o->state, in, sizeof(double) * o->N); }
void ekfo_get_cov(const ekfo *o, double *out) { memcpy(out, o->sigma, sizeof(double) * (size_t)o->N * o->N); }
void ekfo_set_cov(ekfo *o, const double *in) { memcpy(o->sigma, in, sizeof(double) * (size_t)o->N * o->N); }
void ekfo_set_init_flag(ekfo *o, int f) { o->landmark_init_flag = f; }
int ekfo_get_init_flag(const ekfo *o) { return o->landmark_init_flag; }

/* C = A(ra x ca) * B(ca x cb), all row-major, k summed in ascending order. */
static void matmul(const double *A, const double *B, double *C, int ra, int ca, int cb) {
    for (int i = 0; i < ra; i++) {
        double *c = C + (size_t)i * cb;
        for (int j = 0; j < cb; j++) c[j] = 0.0;
        for (int k = 0; k < ca; k++) {
            double a = A[(size_t)i * ca + k];
            const double *b = B + (size_t)k * cb;
            for (int j = 0; j < cb; j++) c[j] += a * b[j];
        }
    }
}

/* motion-model increments, ekf_slam.cpp:67-96; out: upd[3], a10, a20 */
static void motion_terms(const ekfo *o, double dtheta, double dx, double *upd, double *a10, double *a20) {
    double theta = o->state[0];
    if (fabs(dtheta) < o->p.straight_eps) {
        upd[0] = 0;
        upd[1] = dx * cos(theta);
        upd[2] = dx * sin(theta);
        *a10 = -dx * sin(theta);
        *a20 = dx * cos(theta);
    } else {
        upd[0] = dtheta;
        upd[1] = -(dx / dtheta) * sin(theta) + (dx / dtheta) * sin(theta + dtheta);
        upd[2] = (dx / dtheta) * cos(theta) - (dx / dtheta) * cos(theta + dtheta);
        *a10 = -(dx / dtheta) * cos(theta) + (dx / dtheta) * cos(theta + dtheta);
        *a20 = -(dx / dtheta) * sin(theta) + (dx / dtheta) * sin(theta + dtheta);
    }
}

/* rigid2d/src/ekf_slam.cpp:55-106.  twist.linearY() is ignored (:70). */
void ekfo_prediction(ekfo *o, double dtheta, double dx) {
    int N = o->N;
    double upd[3], a10, a20;
    motion_terms(o, dtheta, dx, upd, &a10, &a20);
    /* state = state + update (:99); theta is NOT wrapped here */
    o->state[0] = o->state[0] + upd[0];
    o->state[1] = o->state[1] + upd[1];
    o->state[2] = o->state[2] + upd[2];

    if (o->mode == 0) {
        /* At = eye + A (:101); sigma = At*sigma*At.t() + Q (:102), dense */
        double *At = o->t1, *T = o->t2, *AtT = o->t3;
        memset(At, 0, sizeof(double) * (size_t)N * N);
        for (int i = 0; i < N; i++) At[(size_t)i * N + i] = 1.0;
        At[(size_t)1 * N + 0] += a10;
        At[(size_t)2 * N + 0] += a20;
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) AtT[(size_t)i * N + j] = At[(size_t)j * N + i];
        matmul(At, o->sigma, T, N, N, N);
        matmul(T, AtT, o->sigma, N, N, N);
        for (size_t i = 0; i < (size_t)N * N; i++) o->sigma[i] += o->Q[i];
    } else {
        /* Only rows 1,2 and columns 1,2 of sigma change (A has 2 non-zeros). */
        double *S = o->sigma;
        double c[3][3], T[3][3];
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c[r][k] = S[(size_t)r * N + k];
        for (int k = 0; k < 3; k++) {
            T[0][k] = c[0][k];
            T[1][k] = a10 * c[0][k] + c[1][k];
            T[2][k] = a20 * c[0][k] + c[2][k];
        }
        for (int r = 0; r < 3; r++) {
            S[(size_t)r * N + 0] = T[r][0];
            S[(size_t)r * N + 1] = T[r][0] * a10 + T[r][1];
            S[(size_t)r * N + 2] = T[r][0] * a20 + T[r][2];
        }
        S[0] += o->p.q_pose;
        S[(size_t)1 * N + 1] += o->p.q_pose;
        S[(size_t)2 * N + 2] += o->p.q_pose;
        for (int k = 3; k < N; k++) {
            double s0 = S[k];
            S[(size_t)1 * N + k] = a10 * s0 + S[(size_t)1 * N + k];
            S[(size_t)2 * N + k] = a20 * s0 + S[(size_t)2 * N + k];
            double r0 = S[(size_t)k * N + 0];
            S[(size_t)k * N + 1] = r0 * a10 + S[(size_t)k * N + 1];
            S[(size_t)k * N + 2] = r0 * a20 + S[(size_t)k * N + 2];
        }
    }
}

/* measurement-model terms shared by :137-170, :224-259, :335-368 */
typedef struct {
    double z[2];    /* (r, phi) from the Cartesian reading, :142-146 */
    double zhat[2]; /* (:152-155) bearing wrapped */
    double H5[2][5];/* non-zero columns {0,1,2,3+2i,4+2i} of Hj, :164-166 */
} meas_terms;

static void measurement_terms(const ekfo *o, int i, double sx, double sy, double theta,
                              double x, double y, meas_terms *m) {
    double tx = o->state[i * 2 + 3], ty = o->state[i * 2 + 3 + 1];
    m->z[0] = sqrt(pow(sx, 2) + pow(sy, 2));
    m->z[1] = atan2(sy, sx);
    m->zhat[0] = sqrt(pow(tx - x, 2.0) + pow(ty - y, 2.0));
    m->zhat[1] = ekfo_normalize_angle(atan2(ty - y, tx - x) - theta);
    double delta_x = tx - x, delta_y = ty - y;
    double d = pow(delta_x, 2) + pow(delta_y, 2);
    m->H5[0][0] = 0;  m->H5[0][1] = -delta_x / sqrt(d); m->H5[0][2] = -delta_y / sqrt(d);
    m->H5[1][0] = -1; m->H5[1][1] = delta_y / d;        m->H5[1][2] = -delta_x / d;
    m->H5[0][3] = delta_x / sqrt(d); m->H5[0][4] = delta_y / sqrt(d);
    m->H5[1][3] = -delta_y / d;      m->H5[1][4] = delta_x / d;
}

static void inv2(const double S[2][2], double Si[2][2]) {
    double det = S[0][0] * S[1][1] - S[0][1] * S[1][0];
    Si[0][0] = S[1][1] / det;  Si[0][1] = -S[0][1] / det;
    Si[1][0] = -S[1][0] / det; Si[1][1] = S[0][0] / det;
}

static void fill_dense_H(const ekfo *o, int i, const meas_terms *m, double *H) {
    int N = o->N;
    memset(H, 0, sizeof(double) * 2 * (size_t)N);
    const int idx[5] = {0, 1, 2, 3 + 2 * i, 4 + 2 * i};
    for (int a = 0; a < 2; a++)
        for (int k = 0; k < 5; k++) H[(size_t)a * N + idx[k]] = m->H5[a][k];
}

/* S = Hj*sigma*Hj.t() + R  (:178 inner, :267, :376 inner) */
static void innovation_cov(const ekfo *o, int i, const meas_terms *m, double S[2][2], double *HSout) {
    int N = o->N;
    const int idx[5] = {0, 1, 2, 3 + 2 * i, 4 + 2 * i};
    if (o->mode == 0) {
        double *H = o->t1;            /* 2 x N */
        double *HS = o->t1 + 2 * (size_t)N; /* 2 x N */
        fill_dense_H(o, i, m, H);
        matmul(H, o->sigma, HS, 2, N, N);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                double s = 0.0;
                for (int k = 0; k < N; k++) s += HS[(size_t)a * N + k] * H[(size_t)b * N + k];
                S[a][b] = s;
            }
        if (HSout) memcpy(HSout, HS, sizeof(double) * 2 * (size_t)N);
    } else {
        double HS5[2][5];
        for (int a = 0; a < 2; a++)
            for (int l = 0; l < 5; l++) {
                double s = 0.0;
                for (int k = 0; k < 5; k++) s += m->H5[a][k] * o->sigma[(size_t)idx[k] * N + idx[l]];
                HS5[a][l] = s;
            }
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                double s = 0.0;
                for (int l = 0; l < 5; l++) s += HS5[a][l] * m->H5[b][l];
                S[a][b] = s;
            }
    }
    S[0][0] += o->p.r_meas;
    S[1][1] += o->p.r_meas;
}

/* One landmark correction: ekf_slam.cpp:137-192 (and its copy :331-390).
 * (theta,x,y) is the pose the caller captured: stale in measurement() (:109-111),
 * fresh in data_association() (:331-333). */
static void correct(ekfo *o, int i, double sx, double sy, double theta, double x, double y) {
    int N = o->N;
    const int idx[5] = {0, 1, 2, 3 + 2 * i, 4 + 2 * i};
    meas_terms m;
    measurement_terms(o, i, sx, sy, theta, x, y, &m);
    double S[2][2], Si[2][2];
    double zd[2];

    if (o->mode == 0) {
        double *H = o->t1, *HS = o->t1 + 2 * (size_t)N;
        double *SHt = o->t1 + 4 * (size_t)N; /* N x 2 */
        double *K = o->t1 + 6 * (size_t)N;   /* N x 2 */
        innovation_cov(o, i, &m, S, NULL);   /* fills H, HS in t1 */
        (void)HS;
        /* sigma*Hj.t() : (N x N)(N x 2) */
        for (int r = 0; r < N; r++)
            for (int a = 0; a < 2; a++) {
                double s = 0.0;
                for (int k = 0; k < N; k++) s += o->sigma[(size_t)r * N + k] * H[(size_t)a * N + k];
                SHt[(size_t)r * 2 + a] = s;
            }
        inv2(S, Si);
        for (int r = 0; r < N; r++)
            for (int b = 0; b < 2; b++)
                K[(size_t)r * 2 + b] = SHt[(size_t)r * 2 + 0] * Si[0][b] + SHt[(size_t)r * 2 + 1] * Si[1][b];
        zd[0] = m.z[0] - m.zhat[0];
        zd[1] = ekfo_normalize_angle(m.z[1] - m.zhat[1]);         /* :182-183 */
        for (int r = 0; r < N; r++)                               /* :186 */
            o->state[r] = o->state[r] + (K[(size_t)r * 2 + 0] * zd[0] + K[(size_t)r * 2 + 1] * zd[1]);
        o->state[0] = ekfo_normalize_angle(o->state[0]);          /* :187 */
        /* kh = Ki*Hj (:191); sigma = (eye - kh)*sigma (:192), dense N^3 */
        double *M = o->t2, *out = o->t3;
        for (int r = 0; r < N; r++)
            for (int c = 0; c < N; c++) {
                double kh = K[(size_t)r * 2 + 0] * H[c] + K[(size_t)r * 2 + 1] * H[(size_t)N + c];
                M[(size_t)r * N + c] = (r == c ? 1.0 : 0.0) - kh;
            }
        matmul(M, o->sigma, out, N, N, N);
        memcpy(o->sigma, out, sizeof(double) * (size_t)N * N);
    } else {
        double *Kc = o->t1;              /* N x 2 */
        double *G = o->t1 + 2 * (size_t)N; /* 2 x N : Hj*sigma */
        innovation_cov(o, i, &m, S, NULL);
        inv2(S, Si);
        for (int r = 0; r < N; r++) {
            double sht[2];
            for (int a = 0; a < 2; a++) {
                double s = 0.0;
                for (int k = 0; k < 5; k++) s += o->sigma[(size_t)r * N + idx[k]] * m.H5[a][k];
                sht[a] = s;
            }
            Kc[(size_t)r * 2 + 0] = sht[0] * Si[0][0] + sht[1] * Si[1][0];
            Kc[(size_t)r * 2 + 1] = sht[0] * Si[0][1] + sht[1] * Si[1][1];
        }
        for (int a = 0; a < 2; a++)
            for (int c = 0; c < N; c++) {
                double s = 0.0;
                for (int k = 0; k < 5; k++) s += m.H5[a][k] * o->sigma[(size_t)idx[k] * N + c];
                G[(size_t)a * N + c] = s;
            }
        zd[0] = m.z[0] - m.zhat[0];
        zd[1] = ekfo_normalize_angle(m.z[1] - m.zhat[1]);
        for (int r = 0; r < N; r++)
            o->state[r] = o->state[r] + (Kc[(size_t)r * 2 + 0] * zd[0] + Kc[(size_t)r * 2 + 1] * zd[1]);
        o->state[0] = ekfo_normalize_angle(o->state[0]);
        /* rank-2 update: sigma -= K * (H sigma): 16 N^2 bytes, 4 N^2 flop */
        #pragma omp parallel for schedule(static) if (N >= 512)
        for (int r = 0; r < N; r++) {
            double k0 = Kc[(size_t)r * 2 + 0], k1 = Kc[(size_t)r * 2 + 1];
            double *row = o->sigma + (size_t)r * N;
            const double *g0 = G, *g1 = G + N;
            for (int c = 0; c < N; c++) row[c] = row[c] - (k0 * g0[c] + k1 * g1[c]);
        }
    }
}

/* rigid2d/src/ekf_slam.cpp:200-214 (the std::cout at :213 is dropped) */
static void initialize_landmark(ekfo *o, double sx, double sy, int i) {
    double theta = o->state[0], x = o->state[1], y = o->state[2];
    double ri = sqrt(pow(sx, 2) + pow(sy, 2));
    double phii = atan2(sy, sx);
    o->state[i * 2 + 3] = x + ri * cos(phii + theta);
    o->state[i * 2 + 3 + 1] = y + ri * sin(phii + theta);
}

/* rigid2d/src/ekf_slam.cpp:108-197.  known_list is unused by the reference. */
void ekfo_measurement(ekfo *o, const double *sensor_xy, const unsigned char *visible) {
    double theta = o->state[0], x = o->state[1], y = o->state[2]; /* :109-111 captured ONCE */
    if (!o->landmark_init_flag) {                                 /* :113-128 */
        for (int i = 0; i < o->n; i++) {
            double sx = sensor_xy[i * 2], sy = sensor_xy[i * 2 + 1];
            double ri = sqrt(pow(sx, 2) + pow(sy, 2));
            double phii = atan2(sy, sx);
            o->state[i * 2 + 3] = x + ri * cos(phii + theta);
            o->state[i * 2 + 3 + 1] = y + ri * sin(phii + theta);
        }
        o->landmark_init_flag = 1;
    }
    for (int i = 0; i < o->n; i++) {                              /* :132-194 */
        if (!visible[i]) continue;
        correct(o, i, sensor_xy[i * 2], sensor_xy[i * 2 + 1], theta, x, y);
    }
}

/* Same call on the compact log format (visible readings only, ascending
 * landmark index; idx < 0 terminates).  init_xy (2n) is consumed on the first
 * call only, exactly like sensor_reading at :113-128. */
int ekfo_measurement_compact(ekfo *o, const double *init_xy, const int *lm_idx,
                             const double *z_xy, int vmax) {
    double theta = o->state[0], x = o->state[1], y = o->state[2];
    int done = 0;
    if (!o->landmark_init_flag) {
        for (int i = 0; i < o->n; i++) {
            double sx = init_xy[i * 2], sy = init_xy[i * 2 + 1];
            double ri = sqrt(pow(sx, 2) + pow(sy, 2));
            double phii = atan2(sy, sx);
            o->state[i * 2 + 3] = x + ri * cos(phii + theta);
            o->state[i * 2 + 3 + 1] = y + ri * sin(phii + theta);
        }
        o->landmark_init_flag = 1;
    }
    for (int v = 0; v < vmax; v++) {
        if (lm_idx[v] < 0) break;
        correct(o, lm_idx[v], z_xy[2 * v], z_xy[2 * v + 1], theta, x, y);
        done++;
    }
    return done;
}

/* rigid2d/src/ekf_slam.cpp:217-276.  Innovation bearing NOT wrapped (:269);
 * the std::cout at :268 is dropped. */
double ekfo_maha(ekfo *o, double sx, double sy, int i) {
    double theta = o->state[0], x = o->state[1], y = o->state[2];
    meas_terms m;
    measurement_terms(o, i, sx, sy, theta, x, y, &m);
    double S[2][2], Si[2][2];
    innovation_cov(o, i, &m, S, NULL);
    inv2(S, Si);
    double v0 = m.z[0] - m.zhat[0], v1 = m.z[1] - m.zhat[1];
    /* (v.t()*psi.i())*v */
    double t0 = v0 * Si[0][0] + v1 * Si[1][0];
    double t1 = v0 * Si[0][1] + v1 * Si[1][1];
    return t0 * v0 + t1 * v1;
}

/* rigid2d/src/ekf_slam.cpp:278-402.  known is in/out (:323).  Returns the
 * number of corrections applied; assoc_out (J ints, nullable) records the
 * landmark each measurement was matched to (-1 = dropped).
 *
 * margins (nullable, 5 doubles, MIN-accumulated: the caller starts them at
 * +inf) records how far the discrete decisions of :293-330 sit from flipping --
 * the only place where another summation order (real Armadillo / BLAS against
 * this restatement) could change a result by more than rounding:
 *   [0] min over every scored (reading, landmark) pair of |d - gate_new| / gate_new       (:293,305)
 *   [1] min over every scored pair of |d - gate_update| / gate_update                     (:330)
 *   [2] min over readings with a winner of (runner_up - winner) / runner_up               (:305-309)
 *   [3] min over every scored pair of d itself (a score is a positive-definite form)
 *   [4] the DECISION-RELEVANT minimum: per reading only the smallest score d1 takes part in a comparison whose
 *       outcome matters (:305-309 keeps the first strict minimum) -- min of |d1 - gate_new| / gate_new,
 *       |d1 - gate_update| / gate_update (when d1 < gate_new) and the gap [2].  [0], [1] are stricter: they also
 *       count pairs that lose to another landmark whatever side of a gate they fall on.          */
int ekfo_data_association_m(ekfo *o, const double *meas_xy, int J, unsigned char *known, int *assoc_out,
                            double *margins) {
    int known_count = 0;
    for (int i = 0; i < o->n; i++) {          /* :281-288 leading run of true */
        if (known[i]) known_count++; else break;
    }
    int done = 0;
    for (int j = 0; j < J; j++) {             /* :291 */
        double mx = meas_xy[2 * j], my = meas_xy[2 * j + 1];
        double min_maha_dis = o->p.gate_new;  /* :293 */
        int min_maha_idx = known_count;       /* :294 */
        double first = INFINITY, second = INFINITY;   /* (margins only) the two smallest scores of this reading */
        for (int i = 0; i < known_count; i++) {
            double d = ekfo_maha(o, mx, my, i);
            if (d < min_maha_dis) { min_maha_dis = d; min_maha_idx = i; }
            if (margins && d == d) {
                double a = fabs(d - o->p.gate_new) / o->p.gate_new, b = fabs(d - o->p.gate_update) / o->p.gate_update;
                if (a < margins[0]) margins[0] = a;
                if (b < margins[1]) margins[1] = b;
                if (d < margins[3]) margins[3] = d;
                if (d < first) { second = first; first = d; } else if (d < second) second = d;
            }
        }
        if (margins && min_maha_idx < known_count && second < INFINITY) {
            double g = (second - first) / second;
            if (g < margins[2]) margins[2] = g;
            if (g < margins[4]) margins[4] = g;
        }
        if (margins && first < INFINITY) {
            double a = fabs(first - o->p.gate_new) / o->p.gate_new;
            if (a < margins[4]) margins[4] = a;
            if (first < o->p.gate_new) {
                double b = fabs(first - o->p.gate_update) / o->p.gate_update;
                if (b < margins[4]) margins[4] = b;
            }
        }
        if (min_maha_idx == known_count && min_maha_idx < o->n) { /* :318-327 */
            initialize_landmark(o, mx, my, min_maha_idx);
            known[known_count] = 1;
            known_count++;
            min_maha_dis = 0.0;
        }
        if (assoc_out) assoc_out[j] = -1;
        if (min_maha_dis < o->p.gate_update) {                   /* :330-390 */
            correct(o, min_maha_idx, mx, my, o->state[0], o->state[1], o->state[2]);
            if (assoc_out) assoc_out[j] = min_maha_idx;
            done++;
        }
    }
    return done;
}

int ekfo_data_association(ekfo *o, const double *meas_xy, int J, unsigned char *known, int *assoc_out) {
    return ekfo_data_association_m(o, meas_xy, J, known, assoc_out, NULL);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Replays a compact known-association log over B independent filters
 * (OpenMP over filters).  Layouts: twist[T][B][2] (dtheta, dx),
 * lm_idx[T][B][vmax], z_xy[T][B][vmax][2], init_xy[B][2n].  Steps
 * [0, t_warm) run untimed.  Outputs: out_state[B][N]; out_cov[B][N*N] if not
 * NULL; stats[0] = timed seconds, stats[1] = timed corrections, stats[2] =
 * threads used.  This is the cpu_baseline "port" leg of bench.py. */
int ekfo_batch_run_known(int B, int n, int mode, int T, int t_warm, int vmax,
                         const double *twist, const int *lm_idx, const double *z_xy,
                         const double *init_xy, double *out_state, double *out_cov,
                         int nthreads, double *stats) {
    int N = 3 + 2 * n;
    ekfo **f = (ekfo **)calloc(B, sizeof(ekfo *));
    if (!f) return -1;
    for (int b = 0; b < B; b++) {
        f[b] = ekfo_create(n, mode, NULL);
        if (!f[b]) return -1;
    }
    long long corr = 0;
    double t0 = 0.0, t1 = 0.0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int used = omp_get_max_threads();
    if (B == 1) omp_set_nested(0);
#else
    int used = 1;
#endif
    for (int phase = 0; phase < 2; phase++) {
        int ta = phase == 0 ? 0 : t_warm, tb = phase == 0 ? t_warm : T;
        if (phase == 1) t0 = now_s();
        long long c = 0;
        #pragma omp parallel for schedule(dynamic, 1) reduction(+ : c) if (B > 1)
        for (int b = 0; b < B; b++) {
            for (int t = ta; t < tb; t++) {
                size_t tb_ = (size_t)t * B + b;
                ekfo_prediction(f[b], twist[tb_ * 2], twist[tb_ * 2 + 1]);
                c += ekfo_measurement_compact(f[b], init_xy + (size_t)b * 2 * n,
                                              lm_idx + tb_ * vmax, z_xy + tb_ * vmax * 2, vmax);
            }
        }
        if (phase == 1) { t1 = now_s(); corr = c; }
    }
    for (int b = 0; b < B; b++) {
        ekfo_get_state(f[b], out_state + (size_t)b * N);
        if (out_cov) ekfo_get_cov(f[b], out_cov + (size_t)b * N * N);
        ekfo_destroy(f[b]);
    }
    free(f);
    if (stats) { stats[0] = t1 - t0; stats[1] = (double)corr; stats[2] = (double)used; }
    return 0;
}
