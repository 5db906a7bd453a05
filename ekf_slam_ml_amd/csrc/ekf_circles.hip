// ekf_circles.hip -- batched rigid2d::CircleFitting on gfx950: the
// perception front end that turns 360-beam laser scans into the (x, y) measurements consumed by
// EKF_SLAM::data_association (rigid2d/src/circle_fitting.cpp:11-304, called from
// nuslam/src/landmarks.cpp:141).  One wavefront per scan: beams are spread over the 64 lanes for the
// polar -> Cartesian conversion, lane 0 runs the (inherently sequential, 360-step) clustering state
// machine with all of the reference's quirks, then one LANE per cluster does the algebraic circle fit
// (one-sided Jacobi SVD of the m x 4 design matrix held in LDS, symmetric 4 x 4 Jacobi eigen-solve)
// and the inscribed-angle classification.  Tiny matrices, no reuse: latency-bound by construction;
// throughput comes from many scans in flight (grid = scans).
#include "ekf_kernels.hpp"

namespace ekf {

constexpr int kMaxBeams = 1024;
constexpr int kMaxClusters = 128;  // > kMaxBeams / 7

struct Cluster {
    int n, s0, l0, s1, l1;  // points; segment 0 (start, len), segment 1 (start, len) after a wrap merge
};

__device__ void cf_svd4(double* Z, int n, double s[4], double V[16]) {
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int k = 0; k < n; k++) {
                    const double zp = Z[4 * k + p], zq = Z[4 * k + q];
                    alpha += zp * zp; beta += zq * zq; gamma += zp * zq;
                }
                if (gamma == 0.0) continue;
                const double lim = sqrt(alpha * beta);
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * lim) continue;
                if (fabs(gamma) > off) off = fabs(gamma) / (lim > 0 ? lim : 1.0);
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < n; k++) {
                    const double zp = Z[4 * k + p], zq = Z[4 * k + q];
                    Z[4 * k + p] = c * zp - sn * zq;
                    Z[4 * k + q] = sn * zp + c * zq;
                }
                for (int k = 0; k < 4; k++) {
                    const double vp = V[4 * k + p], vq = V[4 * k + q];
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
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 4; j++)
            if (s[j] > s[i]) {
                const double t = s[i]; s[i] = s[j]; s[j] = t;
                for (int k = 0; k < 4; k++) { const double v = V[4 * k + i]; V[4 * k + i] = V[4 * k + j]; V[4 * k + j] = v; }
            }
}

__device__ void cf_eig4_sym(double A[16], double w[4], double E[16]) {
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
                const double apq = A[4 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[5 * q] - A[5 * p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 4; k++) {
                    const double akp = A[4 * k + p], akq = A[4 * k + q];
                    A[4 * k + p] = c * akp - sn * akq;
                    A[4 * k + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    const double apk = A[4 * p + k], aqk = A[4 * q + k];
                    A[4 * p + k] = c * apk - sn * aqk;
                    A[4 * q + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double ekp = E[4 * k + p], ekq = E[4 * k + q];
                    E[4 * k + p] = c * ekp - sn * ekq;
                    E[4 * k + q] = sn * ekp + c * ekq;
                }
            }
    }
    for (int i = 0; i < 4; i++) w[i] = A[5 * i];
}

// point k of a cluster (segment 0 first, then segment 1)
__device__ __forceinline__ int cf_beam(const Cluster& c, int k) { return k < c.l0 ? c.s0 + k : c.s1 + (k - c.l0); }

__global__ __launch_bounds__(64) void k_circles(const double* __restrict__ ranges, int nb, int max_out,
                                                double* __restrict__ centres, double* __restrict__ radii,
                                                int* __restrict__ counts, double* __restrict__ all_out,
                                                int* __restrict__ n_clusters) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* r = sm;            // [nb]
    double* xs = r + nb;       // [nb]
    double* ys = xs + nb;      // [nb]
    double* Zb = ys + nb;      // [nb][4] design-matrix rows, partitioned by cluster
    __shared__ Cluster cl[kMaxClusters];
    __shared__ int zoff[kMaxClusters];
    __shared__ double res[kMaxClusters][4];
    __shared__ int nc_sh;

    const int s = blockIdx.x, lane = threadIdx.x;
    const double* rg = ranges + (size_t)s * nb;
    const double angle_resolution = 2 * kPI / (double)nb;  // circle_fitting.cpp:16
    for (int i = lane; i < nb; i += 64) {
        const double ri = rg[i];
        r[i] = ri;
        if (i == 0) { xs[0] = ri * cos(0.0); ys[0] = ri * sin(0.0); }           // :25-26
        else {
            const double a = normalize_angle(i * angle_resolution);            // :44-45
            xs[i] = ri * cos(a);
            ys[i] = ri * sin(a);
        }
    }
    __syncthreads();

    if (lane == 0) {  // clusteringRanges(), :11-90
        const double thres = 0.2;
        int nc = 0, cur_start = 0, cur_len = 1;
        for (int i = 1; i < nb; i++) {
            if ((fabs(r[i] - r[i - 1]) < thres) && (i != (nb - 1))) {
            } else {
                if (cur_len > 6 && nc < kMaxClusters) { cl[nc] = Cluster{cur_len, cur_start, cur_len, 0, 0}; nc++; }
                cur_start = i; cur_len = 0;
            }
            cur_len++;
        }
        if (nc > 0) {  // :54-70 (an empty list is UB in the reference; here: no circles)
            const double first_elem_of_first = r[cl[0].s0];
            const Cluster last = cl[nc - 1];
            const double last_elem_of_last = r[last.s0 + last.l0 - 1];
            if (fabs(first_elem_of_first - last_elem_of_last) < thres) {
                if (nc == 1) nc = 0;  // prepended to itself, then popped
                else {
                    cl[0] = Cluster{last.l0 + cl[0].l0, last.s0, last.l0, cl[0].s0, cl[0].l0};
                    nc--;
                }
            }
        }
        int off = 0;
        for (int c = 0; c < nc; c++) { zoff[c] = off; off += cl[c].n; }
        nc_sh = nc;
    }
    __syncthreads();
    const int nc = nc_sh;

    for (int c = lane; c < nc; c += 64) {  // circleRegression() + classifyCircle(), one lane per cluster
        const Cluster cc = cl[c];
        const int m = cc.n;
        double* Z = Zb + 4 * (size_t)zoff[c];
        double x_sum = 0.0, y_sum = 0.0;
        for (int k = 0; k < m; k++) { const int bi = cf_beam(cc, k); x_sum += xs[bi]; y_sum += ys[bi]; }  // :112-117
        const double x_mean = x_sum / (double)m, y_mean = y_sum / (double)m;
        double z_sum = 0.0;
        for (int j = 0; j < m; j++) {                                                              // :124-141
            const int bi = cf_beam(cc, j);
            const double x = xs[bi] - x_mean, y = ys[bi] - y_mean;
            const double zi = x * x + y * y;
            z_sum += zi;
            Z[4 * j + 0] = zi; Z[4 * j + 1] = x; Z[4 * j + 2] = y; Z[4 * j + 3] = 1.0;
        }
        const double z_mean = z_sum / (double)m;
        double sv[4], V[16], A[4];
        cf_svd4(Z, m, sv, V);                                                                      // :168
        if (sv[3] < 1e-12) {                                                                       // :171-175
            for (int k = 0; k < 4; k++) A[k] = V[4 * k + 3];
        } else {
            double Y[16], Hinv[16], T[16], Q[16], w[4], E[16];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    double a = 0.0;
                    for (int k = 0; k < 4; k++) a += V[4 * i + k] * sv[k] * V[4 * j + k];
                    Y[4 * i + j] = a;
                }
            for (int i = 0; i < 16; i++) Hinv[i] = 0.0;                                            // :156-161
            Hinv[3] = 0.5; Hinv[5] = 1.0; Hinv[10] = 1.0; Hinv[12] = 0.5; Hinv[15] = -2.0 * z_mean;
            for (int i = 0; i < 4; i++)
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
            for (int i = 0; i < 4; i++)
                for (int j = i + 1; j < 4; j++) { const double a = 0.5 * (Q[4 * i + j] + Q[4 * j + i]); Q[4 * i + j] = a; Q[4 * j + i] = a; }
            cf_eig4_sym(Q, w, E);                                                                  // :184
            int idx = 0;
            double best = 1000.0;                                                                  // :187-197
            for (int e = 0; e < 4; e++)
                if (w[e] > 0 && w[e] < best) { best = w[e]; idx = e; }
            double As[4], tmp[4];
            for (int k = 0; k < 4; k++) As[k] = E[4 * k + idx];
            for (int k = 0; k < 4; k++) {                                                          // :211
                double a = 0.0;
                for (int i = 0; i < 4; i++) a += V[4 * i + k] * As[i];
                tmp[k] = a / sv[k];
            }
            for (int i = 0; i < 4; i++) {
                double a = 0.0;
                for (int k = 0; k < 4; k++) a += V[4 * i + k] * tmp[k];
                A[i] = a;
            }
        }
        const double a = -A[1] / (2 * A[0]);                                                       // :220-222
        const double bq = -A[2] / (2 * A[0]);
        const double R_sqr = (A[1] * A[1] + A[2] * A[2] - 4 * A[0] * A[3]) / (4 * (A[0] * A[0]));
        const double cx = a + x_mean, cy = bq + y_mean, rad = sqrt(R_sqr);
        // classifyCircle(), :234-296
        const int b1 = cf_beam(cc, 0), b2 = cf_beam(cc, m - 1);
        const double p1x = xs[b1], p1y = ys[b1], p2x = xs[b2], p2y = ys[b2];
        double sum_angle = 0.0;
        for (int k = 1; k < m - 1; k++) {
            const int bi = cf_beam(cc, k);
            const double pp1x = p1x - xs[bi], pp1y = p1y - ys[bi], pp2x = p2x - xs[bi], pp2y = p2y - ys[bi];
            const double top_part = pp1x * pp2x + pp1y * pp2y;
            const double bot_part = sqrt(pp1x * pp1x + pp1y * pp1y) * sqrt(pp2x * pp2x + pp2y * pp2y);
            sum_angle += acos(top_part / bot_part);
        }
        const double mean_angle = sum_angle / (m - 2);
        const int ok = (mean_angle > 1.5708 && mean_angle < 2.3562 && rad < 0.2) ? 1 : 0;           // :264-271
        res[c][0] = cx; res[c][1] = cy; res[c][2] = rad; res[c][3] = (double)ok;
    }
    __syncthreads();
    if (lane == 0) {  // :284-291 keep the classified circles, in cluster order
        int count = 0;
        for (int c = 0; c < nc; c++) {
            if (all_out) for (int k = 0; k < 4; k++) all_out[((size_t)s * kMaxClusters + c) * 4 + k] = res[c][k];
            if (res[c][3] != 0.0 && count < max_out) {
                centres[((size_t)s * max_out + count) * 2] = res[c][0];
                centres[((size_t)s * max_out + count) * 2 + 1] = res[c][1];
                radii[(size_t)s * max_out + count] = res[c][2];
                count++;
            }
        }
        counts[s] = count;
        if (n_clusters) n_clusters[s] = nc;
    }
}

int circles_max_beams() { return kMaxBeams; }
int circles_max_clusters() { return kMaxClusters; }

void launch_circles(const double* ranges, int S, int nb, int max_out, double* centres, double* radii, int* counts,
                    double* all_out, int* n_clusters, hipStream_t s) {
    const size_t lds = sizeof(double) * (size_t)nb * 7;
    hipLaunchKernelGGL(k_circles, dim3(S), dim3(64), lds, s, ranges, nb, max_out, centres, radii, counts, all_out,
                       n_clusters);
}

}  // namespace ekf
