// rocsparse_cmp.hip — non-gating cross-check (SURVEY §8d "Profiling evidence"): the vendor
// library's CSR SpMV on the same matrix, same device, same timing protocol as tools/kbench,
// beside this library's mi_spmv_dev.  Development tool only; nothing in the product links rocSPARSE.
//
//   hipcc --offload-arch=gfx950 -O3 -Wno-deprecated-declarations tools/rocsparse_cmp.hip synth_csr.o \
//         -Inavierstokes_amd/csrc -Iinclude -Lnavierstokes_amd/csrc -lmi355spmv -lrocsparse -o tools/rocsparse_cmp
//   tools/rocsparse_cmp [kind 0|1|2] [n] [alg mask: 1 rowsplit, 2 adaptive, 4 lrb, 8 nnzsplit, 16 adaptive via rocsparse_dcsrmv; default 13]
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi355_spmv.h"

extern "C" long long synth_count(int kind, unsigned long long seed, int n, int w, long long rb, long long re);
extern "C" int synth_rows(int kind, unsigned long long seed, int n, int w, long long rb, long long re, int* ptrow, int* indcol,
                          double* coef);

#define HIPC(x)                                                                               \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)
#define RSC(x)                                                              \
    do {                                                                    \
        rocsparse_status s_ = (x);                                          \
        if (s_ != rocsparse_status_success) {                               \
            fprintf(stderr, "%s:%d %s -> %d\n", __FILE__, __LINE__, #x, (int)s_); \
            exit(3);                                                        \
        }                                                                   \
    } while (0)

static double g_b2b_us = 0;
static double median(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const int n = argc > 2 ? atoi(argv[2]) : 5000000;
    const int mask = argc > 3 ? atoi(argv[3]) : 13;
    const unsigned long long seed = 0x5EED;
    const int w = 2000;
    const long long nnz = synth_count(kind, seed, n, w, 0, n);
    std::vector<int> ptrow(n + 1), indcol(nnz);
    std::vector<double> coef(nnz), x(n), yref(n), y(n);
    synth_rows(kind, seed, n, w, 0, n, ptrow.data(), indcol.data(), coef.data());
    for (int i = 0; i < n; i++) x[i] = sin(0.001 * i);
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) s = fma(coef[k], x[indcol[k]], s);
        yref[i] = s;
    }
    const double B = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("matrix kind=%d n=%d nnz=%lld  algorithmic bytes/SpMV = %.1f MB\n", kind, n, nnz, B / 1e6);

    int *d_ptrow, *d_indcol;
    double *d_coef, *d_x, *d_y;
    HIPC(hipMalloc(&d_ptrow, sizeof(int) * (n + 1)));
    HIPC(hipMalloc(&d_indcol, sizeof(int) * nnz));
    HIPC(hipMalloc(&d_coef, sizeof(double) * nnz));
    HIPC(hipMalloc(&d_x, sizeof(double) * n));
    HIPC(hipMalloc(&d_y, sizeof(double) * n));
    HIPC(hipMemcpy(d_ptrow, ptrow.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(d_indcol, indcol.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(d_coef, coef.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(d_x, x.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    hipStream_t st;
    HIPC(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));

    auto report = [&](const char* name, double us, double prep_ms) {
        HIPC(hipMemcpy(y.data(), d_y, sizeof(double) * n, hipMemcpyDeviceToHost));
        long long diff = 0;
        double num = 0, den = 0;
        for (int i = 0; i < n; i++) {
            if (memcmp(&y[i], &yref[i], 8)) diff++;
            num += (y[i] - yref[i]) * (y[i] - yref[i]);
            den += yref[i] * yref[i];
        }
        printf("%-44s %8.1f us (back-to-back %6.1f us = %6.1f GFLOP/s)  %7.1f GB/s  %5.1f %% of 8 TB/s  %7.1f GFLOP/s  setup %8.2f ms  rows!=fma-chain %lld  rel_err %.2e\n",
               name, us, g_b2b_us, 2.0 * nnz / g_b2b_us / 1e3, B / us / 1e3, B / us / 1e3 / 80.0, 2.0 * nnz / us / 1e3, prep_ms, diff, sqrt(num / den));
    };
    auto time_it = [&](auto&& launch) {
        for (int i = 0; i < 5; i++) launch();
        HIPC(hipStreamSynchronize(st));
        std::vector<float> t;
        for (int r = 0; r < 30; r++) {
            HIPC(hipEventRecord(e0, st));
            launch();
            HIPC(hipEventRecord(e1, st));
            HIPC(hipEventSynchronize(e1));
            float ms;
            HIPC(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3f);
        }
        // ... and back to back (no synchronisation between launches), the protocol of bench.py
        HIPC(hipEventRecord(e0, st));
        for (int r = 0; r < 30; r++) launch();
        HIPC(hipEventRecord(e1, st));
        HIPC(hipEventSynchronize(e1));
        float ms;
        HIPC(hipEventElapsedTime(&ms, e0, e1));
        g_b2b_us = ms * 1e3 / 30;
        return median(t);
    };

    // ---- rocSPARSE, every CSR algorithm it offers
    rocsparse_handle h;
    RSC(rocsparse_create_handle(&h));
    RSC(rocsparse_set_stream(h, st));
    rocsparse_dnvec_descr vx, vy;
    RSC(rocsparse_create_dnvec_descr(&vx, n, d_x, rocsparse_datatype_f64_r));
    RSC(rocsparse_create_dnvec_descr(&vy, n, d_y, rocsparse_datatype_f64_r));
    const double alpha = 1.0, beta = 0.0;
    struct Alg {
        rocsparse_spmv_alg id;
        const char* name;
    } algs[] = {{rocsparse_spmv_alg_csr_rowsplit, "rocsparse_spmv csr_rowsplit (stream)"},
                {rocsparse_spmv_alg_csr_adaptive, "rocsparse_spmv csr_adaptive"},
                {rocsparse_spmv_alg_csr_lrb, "rocsparse_spmv csr_lrb"},
                {rocsparse_spmv_alg_csr_nnzsplit, "rocsparse_spmv csr_nnzsplit"}};
    int bit = 1;
    for (const Alg& a : algs) {
        const bool on = mask & bit;
        bit <<= 1;
        if (!on) continue;
        rocsparse_spmat_descr A; // a fresh descriptor per algorithm: the analysis data lives in it
        RSC(rocsparse_create_csr_descr(&A, n, n, nnz, d_ptrow, d_indcol, d_coef, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                       rocsparse_index_base_zero, rocsparse_datatype_f64_r));
        size_t bs = 0;
        printf("[%s] buffer_size...\n", a.name);
        rocsparse_status s = rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, a.id,
                                            rocsparse_spmv_stage_buffer_size, &bs, nullptr);
        if (s != rocsparse_status_success) {
            printf("%-44s not available (status %d)\n", a.name, (int)s);
            continue;
        }
        void* buf = nullptr;
        HIPC(hipMalloc(&buf, bs ? bs : 8));
        HIPC(hipDeviceSynchronize());
        printf("[%s] buffer %zu B, preprocess...\n", a.name, bs);
        HIPC(hipEventRecord(e0, st));
        s = rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, a.id,
                           rocsparse_spmv_stage_preprocess, &bs, buf);
        HIPC(hipEventRecord(e1, st));
        HIPC(hipEventSynchronize(e1));
        float prep = 0;
        HIPC(hipEventElapsedTime(&prep, e0, e1));
        if (s != rocsparse_status_success) {
            printf("%-44s preprocess failed (status %d)\n", a.name, (int)s);
            (void)hipFree(buf);
            continue;
        }
        HIPC(hipMemsetAsync(d_y, 0xff, sizeof(double) * n, st));
        HIPC(hipStreamSynchronize(st));
        printf("[%s] compute...\n", a.name);
        const double us = time_it([&] {
            RSC(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, a.id,
                               rocsparse_spmv_stage_compute, &bs, buf));
        });
        report(a.name, us, prep);
        (void)hipFree(buf);
        RSC(rocsparse_destroy_spmat_descr(A));
    }

    // ---- csr_adaptive through the level-2 API (rocsparse_dcsrmv_analysis + rocsparse_dcsrmv): the same algorithm the generic
    //      rocsparse_spmv(csr_adaptive) wraps, with the analysis data in a rocsparse_mat_info the caller owns.  The generic
    //      path faulted on a nil address at 5 M rows in round 1 (mask bit 2); this one is mask bit 16.
    if (mask & 16) {
        rocsparse_mat_descr descr;
        rocsparse_mat_info info;
        RSC(rocsparse_create_mat_descr(&descr));
        RSC(rocsparse_create_mat_info(&info));
        printf("[rocsparse_dcsrmv adaptive] analysis...\n");
        HIPC(hipEventRecord(e0, st));
        rocsparse_status s = rocsparse_dcsrmv_analysis(h, rocsparse_operation_none, n, n, (rocsparse_int)nnz, descr, d_coef, d_ptrow, d_indcol, info);
        HIPC(hipEventRecord(e1, st));
        HIPC(hipEventSynchronize(e1));
        float prep = 0;
        HIPC(hipEventElapsedTime(&prep, e0, e1));
        if (s != rocsparse_status_success) printf("rocsparse_dcsrmv_analysis failed (status %d)\n", (int)s);
        else {
            HIPC(hipMemsetAsync(d_y, 0xff, sizeof(double) * n, st));
            HIPC(hipStreamSynchronize(st));
            printf("[rocsparse_dcsrmv adaptive] compute...\n");
            const double us = time_it([&] {
                RSC(rocsparse_dcsrmv(h, rocsparse_operation_none, n, n, (rocsparse_int)nnz, &alpha, descr, d_coef, d_ptrow, d_indcol, info, d_x, &beta, d_y));
            });
            report("rocsparse_dcsrmv + analysis (adaptive)", us, prep);
        }
        RSC(rocsparse_destroy_mat_info(info));
        RSC(rocsparse_destroy_mat_descr(descr));
    }

    // ---- this library, through the C-ABI
    mi_csr_t M = nullptr;
    HIPC(hipDeviceSynchronize());
    printf("[mi_spmv_dev] create...\n");
    HIPC(hipEventRecord(e0, st));
    if (mi_csr_create(n, n, ptrow.data(), indcol.data(), coef.data(), &M) != MI_OK) {
        fprintf(stderr, "mi_csr_create failed\n");
        return 4;
    }
    HIPC(hipEventRecord(e1, st));
    HIPC(hipEventSynchronize(e1));
    float prep = 0;
    HIPC(hipEventElapsedTime(&prep, e0, e1));
    HIPC(hipMemsetAsync(d_y, 0xff, sizeof(double) * n, st));
    const double us = time_it([&] { mi_spmv_dev(M, d_x, d_y, (mi_stream_t)st); });
    char nm[96];
    snprintf(nm, sizeof nm, "mi_spmv_dev (%s)", mi_csr_kernel_name(M));
    report(nm, us, prep);
    mi_csr_destroy(M);
    return 0;
}
