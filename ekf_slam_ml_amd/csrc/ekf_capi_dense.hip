// ekf_capi_dense.hip -- C ABI of include/ekfslam.h, dense general-F covariance propagation (configs[3], fp32 MFMA).
#include "ekf_runtime.hpp"

using namespace ekfrt;

struct ekf_dense_s {
    int device = -1, N = 0, ld = 0;
    hipStream_t stream = nullptr;
    float *F = nullptr, *S = nullptr, *T = nullptr, *Q = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

extern "C" {

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
    for (hipEvent_t e : {d->e0, d->e1})
        if (e) (void)hipEventDestroy(e);
    if (d->stream) (void)hipStreamDestroy(d->stream);
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
    for (int it = 0; it < iterations; it++) {
        ekf::launch_dense_gemm(d->F, d->S, d->T, nullptr, d->ld, false, d->stream, d->N);  // T = At*sigma (:102)
        ekf::launch_dense_gemm(d->T, d->F, d->S, d->Q, d->ld, true, d->stream, d->N);      // sigma = T*At.t() + Q
    }
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

ekf_status ekf_dense_launch_info(ekf_dense_handle d, int* ld, int* tiles, int* n_big, int* n_tail) {
    if (!d) return fail(EKF_ERR_INVALID, "null handle");
    if (ld) *ld = d->ld;
    ekf::dense_gemm_split(d->ld, tiles, n_big, n_tail);
    return EKF_OK;
}

ekf_status ekf_dense_tile_map(ekf_dense_handle d, unsigned char* map) {
    if (!d || !map) return fail(EKF_ERR_INVALID, "null argument");
    ekf::dense_gemm_tile_map(d->ld, map);
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
