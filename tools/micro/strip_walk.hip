// Microbenchmark: the strip-form flush's memory walk, without its multiply-adds, over two layouts of the same bytes.
//   row-major   : Sigma[b][row][ld]              -- a strip workgroup reads 2 KB of every row, 16 KB apart (the shipped layout)
//   strip-major : Sigma[b][strip][row][256 cols] -- a strip workgroup reads one contiguous block of N x 2 KB
// One workgroup of 16 waves per (filter, strip), 8-row groups per wave, non-temporal loads, +1, non-temporal stores: what
// k_flush_strip does to memory (ekf_delayed.hip).  160 KB of dynamic LDS keeps it at one workgroup per CU like the flush.
// build: hipcc --offload-arch=gfx950 -O3 -o strip_walk strip_walk.hip      run: ./strip_walk [B=4096] [N=2003]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));

// map: 0 = a filter's workgroups on one XCD (the flush's decode); 1 = plain order (a filter's strips dealt round the XCDs);
// rows_per_block / row_blocks: the strip cut into row blocks (workgroup = one row block of one strip); order: 0 = strips
// fastest (row-major dispatch), 1 = row blocks fastest
template <bool STRIP_MAJOR, int kWaves = 16, int ROWS = 8, bool CONTIG = false, bool DYN = false>
__global__ __launch_bounds__(64 * kWaves, 1) void k_walk(double* __restrict__ sigma, int N, int ld, int strips, int B, int map,
                                                         int row_blocks, int rows_per_block, int order) {
    extern __shared__ double2_t lds[];
    const int P = strips * row_blocks;
    int b, pp;
    const int full = (B / 8) * 8 * P;
    if (map == 1) {
        b = blockIdx.x / P;
        pp = blockIdx.x % P;
    } else if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        pp = slot % P;
    } else {
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        pp = rest % P;
    }
    const int p = order ? pp / row_blocks : pp % strips;
    const int rb = order ? pp % row_blocks : pp / strips;
    const int row_begin = rb * rows_per_block, row_end = row_begin + rows_per_block < N ? row_begin + rows_per_block : N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) lds[0] = double2_t{0.0, 0.0};
    const size_t filter = (size_t)b * N * ld;
    // double2 units
    const size_t rs = STRIP_MAJOR ? 128 : (size_t)(ld >> 1);
    double2_t* col = reinterpret_cast<double2_t*>(sigma + filter) + (STRIP_MAJOR ? (size_t)p * N * 128 : (size_t)p * 128) + lane;
    const int ngroups = (row_end - row_begin + ROWS - 1) / ROWS;
    // CONTIG: wave w takes a contiguous run of groups instead of every kWaves-th one
    const int per_wave = (ngroups + kWaves - 1) / kWaves;
    const int g_begin = CONTIG ? wave * per_wave : wave, g_step = CONTIG ? 1 : kWaves;
    const int g_end = CONTIG ? (g_begin + per_wave < ngroups ? g_begin + per_wave : ngroups) : ngroups;
    unsigned* ctr = reinterpret_cast<unsigned*>(lds + 8);
    if (DYN) {   // the waves take the strip's groups from a counter instead of every kWaves-th one
        if (threadIdx.x == 0) *ctr = kWaves;
        __syncthreads();
    }
    for (int g = g_begin; g < g_end;) {
        const int r = row_begin + ROWS * g;
        int gnext = g + g_step;
        if (DYN) { unsigned v = 0; if (lane == 0) v = atomicAdd(ctr, 1u); gnext = __builtin_amdgcn_readfirstlane(v); }
        double2_t a[ROWS][2];
#pragma unroll
        for (int u = 0; u < ROWS; u++) {
            const size_t row = r + u < row_end ? r + u : row_end - 1;
            a[u][0] = __builtin_nontemporal_load(col + row * rs);
            a[u][1] = __builtin_nontemporal_load(col + row * rs + 64);
        }
#pragma unroll
        for (int u = 0; u < ROWS; u++) { a[u][0] += 1.0; a[u][1] += 1.0; }
#pragma unroll
        for (int u = 0; u < ROWS; u++) {
            if (r + u >= row_end) break;
            __builtin_nontemporal_store(a[u][0], col + (size_t)(r + u) * rs);
            __builtin_nontemporal_store(a[u][1], col + (size_t)(r + u) * rs + 64);
        }
        g = gnext;
    }
}

// the plain flush's shape: a workgroup of T threads = RB rows x (2 T) columns, dispatched strips-fastest, one group per wave
template <int T, int RB>
__global__ __launch_bounds__(T) void k_tile(double* __restrict__ sigma, int N, int ld, int strips, int row_blocks, int B) {
    extern __shared__ double2_t lds[];
    const int P = strips * row_blocks;
    int b, pp;
    const int full = (B / 8) * 8 * P;
    if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        pp = slot % P;
    } else {
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        pp = rest % P;
    }
    if (threadIdx.x == 0) lds[0] = double2_t{0.0, 0.0};
    const int c2 = (pp % strips) * T + threadIdx.x, r0 = (pp / strips) * RB;
    if (c2 >= (ld >> 1)) return;
    double2_t* col = reinterpret_cast<double2_t*>(sigma + (size_t)b * N * ld) + c2;
    const size_t rs = ld >> 1;
    double2_t a[RB];
#pragma unroll
    for (int u = 0; u < RB; u++) a[u] = __builtin_nontemporal_load(col + (size_t)(r0 + u < N ? r0 + u : N - 1) * rs);
#pragma unroll
    for (int u = 0; u < RB; u++) a[u] += 1.0;
#pragma unroll
    for (int u = 0; u < RB; u++)
        if (r0 + u < N) __builtin_nontemporal_store(a[u], col + (size_t)(r0 + u) * rs);
}

// the same tiles taken by RESIDENT workgroups: workgroup w of its XCD takes the entries w, w + G, w + 2 G, ... of the XCD's
// row-major tile list (what the hardware dispatcher would have handed to short-lived workgroups, in software)
template <int T, int RB, int MODE>
__global__ __launch_bounds__(T) void k_tile_resident(double* __restrict__ sigma, int N, int ld, int strips, int row_blocks, int B,
                                                     unsigned* __restrict__ queue) {
    extern __shared__ double2_t lds[];
    if (threadIdx.x == 0) lds[0] = double2_t{0.0, 0.0};
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, wgs = gridDim.x >> 3;
    const int P = strips * row_blocks;
    const long long list = (long long)((B - xcd + 7) / 8) * P;
    const size_t rs = ld >> 1;
    auto colof = [&](long long s) -> double2_t* {
        const int b = (int)(s / P) * 8 + xcd, pp = (int)(s % P);
        const int c2 = (pp % strips) * T + threadIdx.x, r0 = (pp / strips) * RB;
        return reinterpret_cast<double2_t*>(sigma + (size_t)b * N * ld) + (c2 < (ld >> 1) ? c2 : 0) + (size_t)r0 * rs;
    };
    auto rowsof = [&](long long s) { const int pp = (int)(s % P); const int r0 = (pp / strips) * RB; return N - r0 < RB ? N - r0 : RB; };
    if (MODE == 3) {
        // two register sets: the next tile's loads go out BEFORE this tile's stores (s_waitcnt vmcnt counts loads and stores in
        // one queue on gfx9: a load issued behind stores is not "there" before the stores are acknowledged)
        double2_t a[RB], n[RB];
        long long s = wg;
        if (s >= list) return;
        double2_t* col = colof(s);
        int rows = rowsof(s);
#pragma unroll
        for (int u = 0; u < RB; u++) a[u] = __builtin_nontemporal_load(col + (size_t)(u < rows ? u : rows - 1) * rs);
        for (;;) {
            const long long s2 = s + wgs;
            double2_t* col2 = s2 < list ? colof(s2) : col;
            const int rows2 = s2 < list ? rowsof(s2) : rows;
#pragma unroll
            for (int u = 0; u < RB; u++) n[u] = __builtin_nontemporal_load(col2 + (size_t)(u < rows2 ? u : rows2 - 1) * rs);
#pragma unroll
            for (int u = 0; u < RB; u++) a[u] += 1.0;
#pragma unroll
            for (int u = 0; u < RB; u++)
                if (u < rows) __builtin_nontemporal_store(a[u], col + (size_t)u * rs);
            if (s2 >= list) break;
#pragma unroll
            for (int u = 0; u < RB; u++) a[u] = n[u];
            s = s2; col = col2; rows = rows2;
        }
        return;
    }
    if (MODE == 5) {   // ONE queue for the whole chip (the dispatcher itself deals workgroup i to XCD i % 8: a fixed eighth each)
        __shared__ unsigned next5;
        const long long all = (long long)B * P;
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) next5 = atomicAdd(queue, 1u);
            __syncthreads();
            const long long s5 = next5;
            if (s5 >= all) break;
            const int b = (int)(s5 / P), pp = (int)(s5 % P);
            const int c2 = (pp % strips) * T + threadIdx.x, r0 = (pp / strips) * RB;
            double2_t* col = reinterpret_cast<double2_t*>(sigma + (size_t)b * N * ld) + (c2 < (ld >> 1) ? c2 : 0) + (size_t)r0 * rs;
            const int rows = N - r0 < RB ? N - r0 : RB;
            double2_t a[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) a[u] = __builtin_nontemporal_load(col + (size_t)(u < rows ? u : rows - 1) * rs);
#pragma unroll
            for (int u = 0; u < RB; u++) a[u] += 1.0;
#pragma unroll
            for (int u = 0; u < RB; u++)
                if (u < rows) __builtin_nontemporal_store(a[u], col + (size_t)u * rs);
        }
        return;
    }
    if (MODE == 4) {   // a queue per XCD: a workgroup takes the next entry when it is done with its last (what the dispatcher does)
        __shared__ unsigned next;
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) next = atomicAdd(queue + xcd * 32, 1u);
            __syncthreads();
            const long long s = next;
            if (s >= list) break;
            double2_t* col = colof(s);
            const int rows = rowsof(s);
            double2_t a[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) a[u] = __builtin_nontemporal_load(col + (size_t)(u < rows ? u : rows - 1) * rs);
#pragma unroll
            for (int u = 0; u < RB; u++) a[u] += 1.0;
#pragma unroll
            for (int u = 0; u < RB; u++)
                if (u < rows) __builtin_nontemporal_store(a[u], col + (size_t)u * rs);
        }
        return;
    }
    for (long long s = wg; s < list; s += wgs) {
        double2_t* col = colof(s);
        const int rows = rowsof(s);
        double2_t a[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const double2_t* src = col + (size_t)(u < rows ? u : rows - 1) * rs;
            a[u] = MODE == 2 ? *src : __builtin_nontemporal_load(src);
        }
#pragma unroll
        for (int u = 0; u < RB; u++) a[u] += 1.0;
#pragma unroll
        for (int u = 0; u < RB; u++)
            if (u < rows) { if (MODE == 0) __builtin_nontemporal_store(a[u], col + (size_t)u * rs); else col[(size_t)u * rs] = a[u]; }
    }
}

template <int T, int RB, int MODE>
static void resident_variant(double* s, int N, int ld, int B, size_t bytes, int per_cu, hipEvent_t e0, hipEvent_t e1) {
    static unsigned* queue = nullptr;
    if (!queue) hipMalloc(&queue, 8 * 32 * sizeof(unsigned));
    const int lds = 160 * 1024 / per_cu - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_resident<T, RB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int strips = (ld / 2 + T - 1) / T, row_blocks = (N + RB - 1) / RB;
    const dim3 grid(256 * per_cu);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(queue, 0, 8 * 32 * sizeof(unsigned));
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_tile_resident<T, RB, MODE>), grid, dim3(T), lds, 0, s, N, ld, strips, row_blocks, B, queue);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("RESIDENT workgroups, %d per CU, tiles of %2d rows x %4d columns in the dispatcher's order, mode %d: %.2f ms, %.3f TB/s\n", per_cu, RB, 2 * T, MODE, best,
           2.0 * bytes / (best * 1e-3) / 1e12);
}

template <int T, int RB>
static void tile_variant(double* s, int N, int ld, int B, size_t bytes, int lds, hipEvent_t e0, hipEvent_t e1) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile<T, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds > 0 ? lds : 16);
    const int strips = (ld / 2 + T - 1) / T, row_blocks = (N + RB - 1) / RB;
    const dim3 grid((unsigned)((long long)B * strips * row_blocks));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_tile<T, RB>), grid, dim3(T), lds > 0 ? lds : 16, 0, s, N, ld, strips, row_blocks, B);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("tiles of %2d rows x %4d columns (%4d threads), LDS %3d KB per workgroup: %.2f ms, %.3f TB/s\n", RB, 2 * T, T, lds / 1024, best,
           2.0 * bytes / (best * 1e-3) / 1e12);
}

// the strip walk shared by SHARE workgroups per strip, interleaved by rounds of 16 waves x 8 rows (workgroup j takes the rounds
// j, j + SHARE, ...): SHARE x 8 workgroups sweep ONE window of a filter together; each stages `count` V rows of its strip
// (2 KB each) into LDS first, like the flush (the first group's loads go out before the staging)
template <int SHARE>
__global__ __launch_bounds__(1024, 1) void k_shared(double* __restrict__ sigma, const double* __restrict__ vall, int N, int ld,
                                                   int strips, int B, int count) {
    extern __shared__ double2_t lds[];
    const int P = strips * SHARE;
    int b, pp;
    const int full = (B / 8) * 8 * P;
    if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        pp = slot % P;
    } else {
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        pp = rest % P;
    }
    const int p = pp % strips, j = pp / strips;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rs = ld >> 1;
    double2_t* col = reinterpret_cast<double2_t*>(sigma + (size_t)b * N * ld) + (size_t)p * 128 + lane;
    const int ngroups = (N + 7) >> 3;
    int g = 16 * j + wave;
    double2_t a[8][2];
    auto load = [&](int gg) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t row = 8 * gg + u < N ? 8 * gg + u : N - 1;
            a[u][0] = __builtin_nontemporal_load(col + row * rs);
            a[u][1] = __builtin_nontemporal_load(col + row * rs + 64);
        }
    };
    if (g < ngroups) load(g);
    const double2_t* vb = reinterpret_cast<const double2_t*>(vall + (size_t)b * 96 * ld) + (size_t)p * 128 + lane;
    for (int q = wave; q < count; q += 16) {
        lds[q * 128 + lane] = vb[(size_t)q * rs];
        lds[q * 128 + 64 + lane] = vb[(size_t)q * rs + 64];
    }
    __syncthreads();
    double2_t acc = lds[(wave % (count > 0 ? count : 1)) * 128 + lane];
    while (g < ngroups) {
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u][0] += acc; a[u][1] += 1.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (8 * g + u >= N) break;
            __builtin_nontemporal_store(a[u][0], col + (size_t)(8 * g + u) * rs);
            __builtin_nontemporal_store(a[u][1], col + (size_t)(8 * g + u) * rs + 64);
        }
        g += 16 * SHARE;
        if (g < ngroups) load(g);
    }
}

template <int SHARE>
static void shared_variant(double* s, const double* v, int N, int ld, int strips, int B, size_t bytes, int count, hipEvent_t e0, hipEvent_t e1) {
    const int lds = 160 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shared<SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const dim3 grid((unsigned)((long long)B * strips * SHARE));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_shared<SHARE>), grid, dim3(1024), lds, 0, s, v, N, ld, strips, B, count);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("strips shared by %d workgroups (rounds interleaved), %2d V rows staged: %.2f ms, %.3f TB/s\n", SHARE, count, best,
           2.0 * bytes / (best * 1e-3) / 1e12);
}

template <int W, int R, bool C, bool D = false>
static void variant(double* s, int N, int ld, int strips, int B, size_t bytes, int lds, hipEvent_t e0, hipEvent_t e1) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_walk<false, W, R, C, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const dim3 grid((unsigned)((long long)B * strips));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_walk<false, W, R, C, D>), grid, dim3(64 * W), lds, 0, s, N, ld, strips, B, 0, 1, (N + 7) & ~7, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("whole strips, row-major, %2d waves x %2d rows per group, %s: %.2f ms, %.3f TB/s\n", W, R, D ? "groups from a counter" : C ? "contiguous runs " : "interleaved     ", best,
           2.0 * bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 2003;
    const int ld = (N + 255) / 256 * 256, strips = ld / 256;
    const size_t bytes = (size_t)B * N * ld * sizeof(double);
    double* s = nullptr;
    if (hipMalloc(&s, bytes) != hipSuccess) { printf("hipMalloc of %.1f GB failed\n", bytes / 1e9); return 1; }
    hipMemset(s, 0, bytes);
    const int lds = 160 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_walk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    constexpr int kWaves = 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Cfg { int map, row_blocks, order; const char* name; } cfgs[] = {
        {0, 1, 0, "whole strips, filter on one XCD"}, {1, 1, 0, "whole strips, strips round the XCDs"},
        {0, 16, 0, "16 row blocks, strips fastest "}, {0, 16, 1, "16 row blocks, blocks fastest "},
        {1, 16, 0, "16 row blocks, strips fastest, plain order"}, {0, 4, 0, "4 row blocks, strips fastest  "}};
    for (const Cfg& c : cfgs) {
        const int rows = ((N + c.row_blocks - 1) / c.row_blocks + 7) & ~7, rbl = (N + rows - 1) / rows;
        const dim3 grid((unsigned)((long long)B * strips * rbl));
        for (int rep = 0; rep < 3; rep++)
            for (int mode = 0; mode < 2; mode++) {
                hipEventRecord(e0);
                if (mode) hipLaunchKernelGGL(k_walk<true>, grid, dim3(64 * kWaves), lds, 0, s, N, ld, strips, B, c.map, rbl, rows, c.order);
                else hipLaunchKernelGGL(k_walk<false>, grid, dim3(64 * kWaves), lds, 0, s, N, ld, strips, B, c.map, rbl, rows, c.order);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) printf("%-44s %s: %.2f ms, %.3f TB/s\n", c.name, mode ? "strip-major" : "row-major  ", ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
            }
    }
    double* v = nullptr;
    if (hipMalloc(&v, (size_t)B * 96 * ld * sizeof(double)) != hipSuccess) { printf("hipMalloc of V failed\n"); return 1; }
    hipMemset(v, 0, (size_t)B * 96 * ld * sizeof(double));
    for (int count : {2, 64}) {
        shared_variant<1>(s, v, N, ld, strips, B, bytes, count, e0, e1);
        shared_variant<2>(s, v, N, ld, strips, B, bytes, count, e0, e1);
        shared_variant<4>(s, v, N, ld, strips, B, bytes, count, e0, e1);
    }
    resident_variant<256, 16, 0>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 16, 1>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 16, 2>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 16, 3>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 16, 3>(s, N, ld, B, bytes, 2, e0, e1);
    resident_variant<256, 16, 0>(s, N, ld, B, bytes, 2, e0, e1);
    resident_variant<256, 16, 4>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 16, 4>(s, N, ld, B, bytes, 2, e0, e1);
    resident_variant<256, 16, 5>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 32, 5>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 8, 5>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<128, 32, 5>(s, N, ld, B, bytes, 1, e0, e1);
    resident_variant<256, 16, 5>(s, N, ld, B, bytes, 2, e0, e1);
    tile_variant<256, 16>(s, N, ld, B, bytes, 0, e0, e1);
    tile_variant<256, 16>(s, N, ld, B, bytes, 40 * 1024, e0, e1);
    tile_variant<256, 16>(s, N, ld, B, bytes, 80 * 1024, e0, e1);
    tile_variant<256, 16>(s, N, ld, B, bytes, 160 * 1024, e0, e1);
    tile_variant<128, 16>(s, N, ld, B, bytes, 0, e0, e1);
    tile_variant<128, 32>(s, N, ld, B, bytes, 0, e0, e1);
    tile_variant<1024, 16>(s, N, ld, B, bytes, 0, e0, e1);
    tile_variant<1024, 16>(s, N, ld, B, bytes, 160 * 1024, e0, e1);
    variant<16, 8, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<16, 8, false, true>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<8, 8, false, true>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<16, 4, false, true>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<16, 8, true>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<16, 12, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<8, 16, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<8, 24, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<8, 8, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<4, 32, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<4, 8, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<4, 16, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<2, 8, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    variant<4, 4, false>(s, N, ld, strips, B, bytes, lds, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    hipFree(s);
    return 0;
}
