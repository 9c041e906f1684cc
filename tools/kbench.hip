// tools/kbench.hip — standalone kernel microbenchmark (development tool, not part
// of the product library or of bench.py).  Generates an S15/SVAR/SFE matrix with
// the product's generator, runs every SpMV kernel variant in ONE process with
// interleaved rounds (cdna_hip_programming.md §5.4 rule 24), verifies each result
// bitwise against a host fma chain, and prints achieved algorithmic GB/s next to
// two calibration streams (device copy; A-stream read without the x gather).
//
//   hipcc -O3 --offload-arch=gfx950 -I../include -I../navierstokes_amd/csrc \
//         kbench.hip ../navierstokes_amd/csrc/synth_csr.c -o kbench
//   ./kbench [kind=0|1|2] [n=5000000] [w=2000] [rounds=5] [iters=50]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "partition.hpp"
#include "spmv_kernels.hpp"
#include "spmv_rowpar.hpp"
#include "spmv_experimental.hpp"

extern "C" long long synth_count(int kind, unsigned long long seed, int n, int w, long long rb, long long re);
extern "C" int synth_rows(int kind, unsigned long long seed, int n, int w, long long rb, long long re, int* ptrow,
                          int* indcol, double* coef);
extern "C" void synth_x_sin(long long jb, long long je, double* x);

using namespace mi355;

#define CK(e)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (e);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void copy_kernel(const double2* __restrict__ a, double2* __restrict__ b, size_t n2)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) b[i] = a[i];
}

// reads coef + indcol like an SpMV would (unit stride) but gathers nothing
__global__ void astream_kernel(const double* __restrict__ coef, const int* __restrict__ indcol, size_t nnz,
                               double* __restrict__ sink)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) s += coef[i] * (double)indcol[i];
    if (s == 123.456) sink[0] = s;
}

// like astream_kernel, but organised as the ring kernels stream the matrix: every
// workgroup walks its OWN contiguous run in 2048-element chunks (tests whether
// hundreds of concurrent sequential streams cost DRAM efficiency)
__global__ __launch_bounds__(512) void astream_runs_kernel(const double* __restrict__ coef, const int* __restrict__ indcol,
                                                           size_t nnz, double* __restrict__ sink)
{
    const size_t per = (nnz + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < nnz ? lo + per : nnz;
    double s = 0;
#pragma unroll 4
    for (size_t base = lo; base + 2048 <= hi; base += 2048) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const size_t k = base + threadIdx.x + i * 512;
            s += coef[k] * (double)indcol[k];
        }
    }
    if (s == 123.456) sink[0] = s;
}

// run-organised stream with U loads of each array in flight per thread
template <int T, int U>
__global__ __launch_bounds__(T) void astream_runs2_kernel(const double* __restrict__ coef, const int* __restrict__ indcol,
                                                          size_t nnz, double* __restrict__ sink)
{
    const size_t per = (nnz + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < nnz ? lo + per : nnz;
    double s = 0;
    for (size_t base = lo; base + (size_t)T * U <= hi; base += (size_t)T * U) {
        double c[U];
        int j[U];
#pragma unroll
        for (int i = 0; i < U; i++) {
            c[i] = coef[base + threadIdx.x + (size_t)i * T];
            j[i] = indcol[base + threadIdx.x + (size_t)i * T];
        }
#pragma unroll
        for (int i = 0; i < U; i++) s += c[i] * (double)j[i];
    }
    if (s == 123.456) sink[0] = s;
}

// run-organised stream, software-pipelined like the ring kernels: D stages of PER loads each,
// consume the oldest stage, refill it, repeat (FEAT bit0: clamp indices to a per-block "last";
// bit1: one 8-byte store per thread per block; bit2: per-block metadata through LDS)
template <int T, int PER, int D, int FEAT>
__global__ __launch_bounds__(T) void astream_pipe_kernel(const double* __restrict__ coef, const unsigned* __restrict__ indcol,
                                                         size_t nnz, double* __restrict__ sink, double* __restrict__ ydummy)
{
    __shared__ int s_meta[1024];
    const size_t per = (nnz + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < nnz ? lo + per : nnz;
    const int nblk = (int)((hi - lo) / ((size_t)T * PER));
    if (FEAT & 4) {
        for (int i = threadIdx.x; i < 1024; i += T) s_meta[i] = (T * PER - 1) - (i & 1);
        __syncthreads();
    }
    double c[D][PER];
    unsigned j[D][PER];
    auto issue = [&](int b, int s) {
        const size_t base = lo + (size_t)min(b, nblk - 1) * T * PER;
        int last = T * PER - 1;
        if (FEAT & 4) last = s_meta[b & 1023];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            int k = threadIdx.x + i * T;
            if (FEAT & 1) k = min(k, last);
            c[s][i] = coef[base + k];
            j[s][i] = indcol[base + k];
        }
    };
#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    double acc = 0;
    for (int g = 0; g < nblk; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) t += c[s][i] * (double)j[s][i];
            acc += t;
            issue(g + s + D, s);
            if (FEAT & 2) ydummy[(size_t)blockIdx.x * T * 4 + ((g + s) & 3) * T + threadIdx.x] = t;
            if (FEAT & 8) __builtin_nontemporal_store(t, &ydummy[(size_t)blockIdx.x * T * 4 + ((g + s) & 3) * T + threadIdx.x]);
            if ((FEAT & 16) && threadIdx.x >= T - 64) ydummy[(size_t)blockIdx.x * T * 4 + ((g + s) & 3) * T + threadIdx.x] = t; // one wave stores
            if ((FEAT & 32) && threadIdx.x < 273) ydummy[(size_t)blockIdx.x * T * 4 + ((g + s) & 3) * T + threadIdx.x] = t;     // 273 rows like S15
            if ((FEAT & 64) && ((g + s) & 3) == 3) { // every 4th block, 4x the data
                double4 v4 = make_double4(t, t, t, t);
                *reinterpret_cast<double4*>(&ydummy[(size_t)blockIdx.x * T * 4 + threadIdx.x * 4]) = v4;
            }
            if ((FEAT & 256) && threadIdx.x < 273) ydummy[(size_t)blockIdx.x * 19600 + (size_t)(g + s) * 273 + threadIdx.x] = t; // fresh lines, like y
            if ((FEAT & 512) && threadIdx.x < 273) __builtin_nontemporal_store(t, &ydummy[(size_t)blockIdx.x * 19600 + (size_t)(g + s) * 273 + threadIdx.x]);
            if ((FEAT & 1024) && threadIdx.x < 273) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(&ydummy[(size_t)blockIdx.x * 19600 + (size_t)(g + s) * 273 + threadIdx.x]), "v"(t) : "memory"); }
            if ((FEAT & 2048) && ((g + s) & 7) == 7) { // fresh lines, one 16 KB burst per 8 blocks
                double4 v4 = make_double4(t, t, t, t);
                *reinterpret_cast<double4*>(&ydummy[(size_t)blockIdx.x * 19600 + (size_t)((g + s) >> 3) * 2048 + threadIdx.x * 4]) = v4;
            }
            if ((FEAT & 4096) && ((g + s) & 31) == 31) { // fresh lines, 64 KB burst per 32 blocks (4 x double4 per thread)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    double4 v4 = make_double4(t, t, t, t);
                    *reinterpret_cast<double4*>(&ydummy[(size_t)blockIdx.x * 19600 + (size_t)((g + s) >> 5) * 8192 + q * 2048 + threadIdx.x * 4]) = v4;
                }
            }
            if (FEAT & 8192) { // fresh lines, 8x the y volume: every thread stores PER values per block
#pragma unroll
                for (int i = 0; i < PER; i++) ydummy[(size_t)blockIdx.x * (size_t)(nblk + 4) * T * PER + (size_t)(g + s) * T * PER + i * T + threadIdx.x] = t + i;
            }
            if (FEAT & 16384) { // fresh lines, all 512 threads store one value per block (2x the y-like volume)
                ydummy[(size_t)blockIdx.x * (size_t)(nblk + 4) * T + (size_t)(g + s) * T + threadIdx.x] = t;
            }
            if (FEAT & 128) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(&ydummy[(size_t)blockIdx.x * T * 4 + ((g + s) & 3) * T + threadIdx.x]), "v"(t) : "memory"); }
        }
    }
    if (acc == 123.456) sink[0] = acc;
}

// background writer for the interference experiment: few workgroups write fresh lines at a throttled rate
__global__ __launch_bounds__(256) void bg_writer_kernel(double* __restrict__ dst, size_t n, int sleep_cycles, int reps)
{
    for (int r = 0; r < reps; r++) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            dst[i] = (double)i;
            for (int q = 0; q < sleep_cycles; q++) __builtin_amdgcn_s_sleep(16); // ~1024 cycles each
        }
    }
}

// "front order" variant of the pipelined stream: workgroup g takes blocks g, g+W, g+2W, ... so the whole
// chip reads (and writes) one moving front instead of W private runs.  STORE: 0 none, 1 y-like (273 x 8 B per block)
template <int T, int PER, int D, int STORE>
__global__ __launch_bounds__(T) void astream_front_kernel(const double* __restrict__ coef, const unsigned* __restrict__ indcol,
                                                          size_t nnz, double* __restrict__ sink, double* __restrict__ yout)
{
    const int W = gridDim.x;
    const int nblk_total = (int)(nnz / ((size_t)T * PER));
    const int mine = (nblk_total - (int)blockIdx.x + W - 1) / W; // blocks of this workgroup
    double c[D][PER];
    unsigned j[D][PER];
    auto issue = [&](int it, int s) {
        const int b = min((int)blockIdx.x + it * W, nblk_total - 1);
        const size_t base = (size_t)b * T * PER;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            c[s][i] = coef[base + threadIdx.x + i * T];
            j[s][i] = indcol[base + threadIdx.x + i * T];
        }
    };
#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    double acc = 0;
    for (int g = 0; g < mine; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) t += c[s][i] * (double)j[s][i];
            acc += t;
            issue(g + s + D, s);
            if (STORE == 1 && threadIdx.x < 273) {
                const int b = min((int)blockIdx.x + (g + s) * W, nblk_total - 1);
                yout[(size_t)b * 273 + threadIdx.x] = t;
            }
        }
    }
    if (acc == 123.456) sink[0] = acc;
}

// run-round-robin order: runs of R consecutive blocks dealt to the W workgroups in turn
// (R = 1: one moving front; R = blocks/W: private contiguous runs).  STORE 1: y-like stores.
template <int T, int PER, int D, int STORE>
__global__ __launch_bounds__(T) void astream_rr_kernel(const double* __restrict__ coef, const unsigned* __restrict__ indcol,
                                                       size_t nnz, double* __restrict__ sink, double* __restrict__ yout, int R)
{
    const int W = gridDim.x;
    const int nblk_total = (int)(nnz / ((size_t)T * PER));
    const int nruns = (nblk_total + R - 1) / R;
    const int myruns = (nruns - (int)blockIdx.x + W - 1) / W;
    const int mine = myruns * R;
    double c[D][PER];
    unsigned j[D][PER];
    auto blk_of = [&](int it) { return min(((int)blockIdx.x + (it / R) * W) * R + (it % R), nblk_total - 1); };
    auto issue = [&](int it, int s) {
        const size_t base = (size_t)blk_of(it) * T * PER;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            c[s][i] = coef[base + threadIdx.x + i * T];
            j[s][i] = indcol[base + threadIdx.x + i * T];
        }
    };
#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    double acc = 0;
    for (int g = 0; g < mine; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) t += c[s][i] * (double)j[s][i];
            acc += t;
            issue(g + s + D, s);
            if (STORE == 1 && threadIdx.x < (T * PER) / 15) yout[(size_t)blk_of(g + s) * ((T * PER) / 15) + threadIdx.x] = t;
        }
    }
    if (acc == 123.456) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const int n = argc > 2 ? atoi(argv[2]) : 5000000;
    const int w = argc > 3 ? atoi(argv[3]) : 2000;
    const int rounds = argc > 4 ? atoi(argv[4]) : 5;
    const int iters = argc > 5 ? atoi(argv[5]) : 50;
    const unsigned long long seed = 0x5EED;
    const char* filter = argc > 6 ? argv[6] : nullptr; // comma-separated substrings of variant names to keep

    const long long nnz = synth_count(kind, seed, n, w, 0, n);
    std::vector<int> ptrow(n + 1), indcol(nnz);
    std::vector<double> coef(nnz), x(n), yref(n);
    if (synth_rows(kind, seed, n, w, 0, n, ptrow.data(), indcol.data(), coef.data())) return 1;
    synth_x_sin(0, n, x.data());
    for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) s = fma(coef[k], x[indcol[k]], s);
        yref[i] = s;
    }
    const double B = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("matrix kind=%d n=%d nnz=%lld w=%d  algorithmic bytes/SpMV = %.1f MB\n", kind, n, nnz, w, B / 1e6);

    int *d_ptrow, *d_indcol;
    double *d_coef, *d_x, *d_y, *d_sink;
    CK(hipMalloc(&d_ptrow, sizeof(int) * (n + 1)));
    CK(hipMalloc(&d_indcol, sizeof(int) * (nnz + 64)));
    CK(hipMalloc(&d_coef, sizeof(double) * (nnz + 64)));
    CK(hipMalloc(&d_x, sizeof(double) * (n + 64)));
    CK(hipMalloc(&d_y, sizeof(double) * ((size_t)n + 8192 * 1024))); // + scratch behind y for the unconditional-store experiment
    CK(hipMalloc(&d_sink, 64));
    CK(hipMemset(d_indcol, 0, sizeof(int) * (nnz + 64)));
    CK(hipMemset(d_coef, 0, sizeof(double) * (nnz + 64)));
    CK(hipMemcpy(d_ptrow, ptrow.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_indcol, indcol.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_coef, coef.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, x.data(), sizeof(double) * n, hipMemcpyHostToDevice));

    // row-block tables for the block sizes under test
    struct Tab { int nnzb; int nblk; int2* d_blk; int2* d_span; int4* d_meta; };
    auto make_tab = [&](int nnzb, int max_rows) {
        std::vector<int> rows, ptrs;
        build_row_blocks(n, ptrow.data(), nnzb, max_rows, rows, ptrs);
        Tab T;
        T.nnzb = nnzb;
        T.nblk = (int)rows.size() - 1;
        std::vector<int2> h(rows.size());
        for (size_t i = 0; i < rows.size(); i++) h[i] = make_int2(rows[i], ptrs[i]);
        std::vector<int2> sp(T.nblk);
        for (int b = 0; b < T.nblk; b++) {
            int lo = 1 << 30, hi = 0;
            for (int k = ptrs[b]; k < ptrs[b + 1]; k++) { lo = std::min(lo, indcol[k]); hi = std::max(hi, indcol[k]); }
            sp[b] = make_int2(lo, hi);
        }
        CK(hipMalloc(&T.d_blk, sizeof(int2) * h.size()));
        CK(hipMemcpy(T.d_blk, h.data(), sizeof(int2) * h.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&T.d_span, sizeof(int2) * (T.nblk + 1)));
        CK(hipMemcpy(T.d_span, sp.data(), sizeof(int2) * T.nblk, hipMemcpyHostToDevice));
        std::vector<int4> mt(T.nblk + 1);
        for (int b = 0; b < T.nblk; b++) mt[b] = make_int4(rows[b], ptrs[b], sp[b].x, sp[b].y);
        mt[T.nblk] = make_int4(rows[T.nblk], ptrs[T.nblk], 0, 0);
        CK(hipMalloc(&T.d_meta, sizeof(int4) * mt.size()));
        CK(hipMemcpy(T.d_meta, mt.data(), sizeof(int4) * mt.size(), hipMemcpyHostToDevice));
        return T;
    };
    auto view = [&](const Tab& T) {
        CsrView V;
        V.n = n; V.ncols = n; V.ptrow = d_ptrow; V.indcol = d_indcol; V.coef = d_coef; V.rowmap = nullptr;
        V.blk = T.d_blk; V.blk_span = T.d_span; V.nblk = T.nblk;
        return V;
    };
    Tab T1k = make_tab(1024, 1024), T2k = make_tab(2048, 1024), T4k = make_tab(4096, 1024);

    // host-side window plan for the ring5 kernels
    struct HostTab { std::vector<int> rows, ptrs; std::vector<int2> span; };
    int g_row_align = 1;
    auto host_tab = [&](int nnzb) {
        HostTab H;
        build_row_blocks(n, ptrow.data(), nnzb, 1024, H.rows, H.ptrs, g_row_align);
        const int nblk = (int)H.rows.size() - 1;
        H.span.resize(nblk);
        for (int b = 0; b < nblk; b++) {
            int lo = 1 << 30, hi = -1;
            for (int k = H.ptrs[b]; k < H.ptrs[b + 1]; k++) { lo = std::min(lo, indcol[k]); hi = std::max(hi, indcol[k]); }
            H.span[b] = make_int2(lo, hi);
        }
        return H;
    };
    std::map<int, HostTab> htabs;
    int g_stagger = 0;
    auto make_plan = [&](int nnzb, int ring, int maxb, int min_wgs, const int4** P, const int** OK, int* wgs_out, int* bpw_out) {
        const int hkey = nnzb * 64 + g_row_align;
        if (!htabs.count(hkey)) htabs[hkey] = host_tab(nnzb);
        const HostTab& H = htabs[hkey];
        const int nblk = (int)H.rows.size() - 1;
        int wgs = std::max(min_wgs, (nblk + maxb - 1) / maxb);
        wgs = ((wgs + 7) / 8) * 8;
        int bpw = (nblk + wgs - 1) / wgs;
        const int R = g_plan_rr; // > 0: runs of R consecutive blocks dealt round-robin to the workgroups
        const int nruns = R ? (nblk + R - 1) / R : 0;
        if (R) bpw = ((nruns + wgs - 1) / wgs) * R;
        const size_t nslots = R ? (size_t)wgs * bpw : (size_t)nblk;
        std::vector<int4> plan(2 * nslots);
        std::vector<int> ok(wgs, 1);
        int nrestart = 0;
        for (int g = 0; g < wgs; g++) {
            int wlo = 0, whi = 0, base = 0;
            bool live = false;
            const int b0 = g * bpw, b1 = R ? b0 + bpw : std::min(nblk, (g + 1) * bpw), cnt = b1 - b0;
            if (cnt <= 0) continue;
            // processing order of the run: rotated by a per-run offset (stagger) so that the
            // workgroups do not all sit at the same phase of their equally long runs
            const int rot = g_stagger ? (int)(((long long)g * g_stagger) % cnt) : 0;
            for (int pos = 0; pos < cnt; pos++) {
                int b = b0 + (pos + rot) % cnt; // actual block
                if (R) {
                    const long long q = g + (long long)(pos / R) * wgs;
                    b = q < nruns ? (int)(q * R + pos % R) : nblk;
                    if (b > nblk) b = nblk;
                }
                const int slot = b0 + pos;             // where the kernel finds it
                if (b >= nblk) { // padding: an empty block at the end of the matrix
                    plan[2 * slot] = make_int4(H.rows[nblk], H.ptrs[nblk], 0, 0);
                    plan[2 * slot + 1] = make_int4(0, 0, base, 0);
                    continue;
                }
                const int nn = H.ptrs[b + 1] - H.ptrs[b], nrows = H.rows[b + 1] - H.rows[b];
                plan[2 * slot] = make_int4(H.rows[b], H.ptrs[b], nrows, nn);
                plan[2 * slot + 1] = make_int4(0, 0, base, 0);
                if (nn == 0) continue;
                const int cmin = H.span[b].x, cmax = H.span[b].y;
                bool use = nn <= nnzb && (cmax - cmin + 1 <= ring);
                if (use) {
                    int lo = live ? wlo : cmin, hi = live ? whi : cmin;
                    bool restart = !live;
                    if (cmin < lo || cmin > hi) { lo = cmin; hi = cmin; restart = true; } // jump: restart the window
                    const int nhi = std::max(hi, cmax + 1), nlo = std::max(lo, nhi - ring);
                    if (cmin < nlo) use = false;
                    else {
                        if (restart) { base = (lo / ring) * ring; nrestart++; }
                        while (nlo - base >= ring) base += ring;
                        plan[2 * slot + 1] = make_int4(hi, nhi - hi, base, 1);
                        wlo = nlo; whi = nhi; live = true;
                    }
                }
                if (!use) ok[g] = 0;
            }
        }
        if (g_want_slots) { // 16-bit ring slots in thread order (the product's build_ring_slots, for this plan layout)
            const int T = g_want_slots > 1 ? g_want_slots : 512, per = nnzb / T;
            std::vector<unsigned short> sl(nslots * nnzb, 0);
            for (size_t b = 0; b < nslots; b++) {
                const int4 m0 = plan[2 * b], m1 = plan[2 * b + 1];
                if (!m1.w || m0.w <= 0 || m0.w > nnzb) continue;
                for (int t = 0; t < T; t++)
                    for (int i = 0; i < per; i++) {
                        const int kk = std::min(t + i * T, m0.w - 1);
                        int p = indcol[m0.y + kk] - m1.z;
                        if (p >= ring) p -= ring;
                        sl[b * nnzb + (size_t)t * per + i] = (unsigned short)p;
                    }
            }
            unsigned short* dS;
            CK(hipMalloc(&dS, sizeof(unsigned short) * sl.size()));
            CK(hipMemcpy(dS, sl.data(), sizeof(unsigned short) * sl.size(), hipMemcpyHostToDevice));
            g_last_slots = dS;
        }
        int4* dP; int* dOK;
        CK(hipMalloc(&dP, sizeof(int4) * plan.size()));
        CK(hipMemcpy(dP, plan.data(), sizeof(int4) * plan.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&dOK, sizeof(int) * ok.size()));
        CK(hipMemcpy(dOK, ok.data(), sizeof(int) * ok.size(), hipMemcpyHostToDevice));
        int nbad = 0; for (int v : ok) nbad += !v;
        printf("plan row_align=%d stagger=%d rr=%d nnzb=%d ring=%d: %d blocks, %d runs of <=%d blocks, %d runs not ring-able, %d window restarts\n", g_row_align, g_stagger, R, nnzb, ring, nblk, wgs, bpw, nbad, nrestart);
        *P = dP; *OK = dOK; *wgs_out = wgs; *bpw_out = bpw;
        g_last_plan_nblk = R ? (int)nslots : nblk;
    };
    std::vector<Variant> vars;
    auto grid8 = [](int nblk) { return dim3(kNXCD * ((nblk + kNXCD - 1) / kNXCD)); };
    {
        CsrView V = view(T2k);
        vars.push_back({"stream<2048>", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream<2048>), grid8(V.nblk), dim3(kWG), 0, s, V, d_x, d_y); }});
    }
    {
        CsrView V = view(T1k);
        vars.push_back({"stream<1024>", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream<1024>), grid8(V.nblk), dim3(kWG), 0, s, V, d_x, d_y); }});
    }
    {
        CsrView V = view(T2k);
        vars.push_back({"rowpar", [=](hipStream_t s) { hipLaunchKernelGGL(spmv_csr_rowpar, dim3((n + kWG - 1) / kWG), dim3(kWG), 0, s, V, d_x, d_y); }});
    }
    add_experimental_variants(vars, n, d_ptrow, d_indcol, d_coef, d_x, d_y, view(T1k), view(T2k), view(T4k), T1k.d_meta, T2k.d_meta, T4k.d_meta, make_plan, &g_stagger, &g_row_align);

    if (filter) {
        std::vector<Variant> keep;
        std::string f(filter);
        for (auto& v : vars) {
            size_t pos = 0;
            bool hit = false;
            while (pos <= f.size()) {
                size_t q = f.find(',', pos);
                if (q == std::string::npos) q = f.size();
                const std::string tok = f.substr(pos, q - pos);
                if (!tok.empty() && v.name.find(tok) != std::string::npos) hit = true;
                pos = q + 1;
            }
            if (hit) keep.push_back(v);
        }
        vars.swap(keep);
    }
    // calibration streams
    const size_t copy_bytes = (size_t)1 << 30; // 1 GiB read + 1 GiB write
    double2 *d_a, *d_b;
    CK(hipMalloc(&d_a, copy_bytes));
    CK(hipMalloc(&d_b, copy_bytes));
    CK(hipMemset(d_a, 1, copy_bytes));
    Variant vcopy{"copy 1GiB->1GiB (double2)", [=](hipStream_t s) { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, s, d_a, d_b, copy_bytes / 16); }};
    Variant vastr{"A-stream read (coef+indcol, no gather)", [=](hipStream_t s) { hipLaunchKernelGGL(astream_kernel, dim3(256 * 8), dim3(256), 0, s, d_coef, d_indcol, (size_t)nnz, d_sink); }};

    std::vector<Variant> calib;
    for (int wgs : {256, 512, 1024, 2048, 8192})
        calib.push_back({"A-stream as " + std::to_string(wgs) + " contiguous runs x512 thr", [=](hipStream_t s) { hipLaunchKernelGGL(astream_runs_kernel, dim3(wgs), dim3(512), 0, s, d_coef, d_indcol, (size_t)nnz, d_sink); }});
    for (int wgs : {256, 512})
        calib.push_back({"A-stream as " + std::to_string(wgs) + " runs x1024 thr (u8)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_runs2_kernel<1024, 8>), dim3(wgs), dim3(1024), 0, s, d_coef, d_indcol, (size_t)nnz, d_sink); }});
    for (int wgs : {256, 512})
        calib.push_back({"A-stream as " + std::to_string(wgs) + " runs x512 thr (u16)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_runs2_kernel<512, 16>), dim3(wgs), dim3(512), 0, s, d_coef, d_indcol, (size_t)nnz, d_sink); }});
    {
        const unsigned* ucol = reinterpret_cast<const unsigned*>(d_indcol);
        double* ydum = d_y;
        double* d_a_as_y = reinterpret_cast<double*>(d_a); // 1 GiB scratch (the copy source) as a big write target
        calib.push_back({"pipe 256x512 PER8 D2", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 0>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D3", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 3, 0>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +clamp", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 1>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +store", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 2>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +nt store", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 8>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +store by last wave only", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 16>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +store 273 thr", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 32>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +store x4 every 4th", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 64>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +sc0sc1 store", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 128>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +store fresh lines (y-like)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 256>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +nt store fresh lines", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 512>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +sc0sc1nt store fresh", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 1024>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +fresh 16KB burst / 8 blocks", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 2048>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +fresh 64KB burst / 32 blocks", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 4096>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +fresh, 512 thr x 8B (4 KB/blk)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 16384>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, d_a_as_y); }});
        calib.push_back({"pipe 256x512 PER8 D2 +fresh, 512 thr x 64B (32 KB/blk)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 8192>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, d_a_as_y); }});
        calib.push_back({"front 256x512 PER8 D2 (one moving front)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_front_kernel<512, 8, 2, 0>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"front 256x512 PER8 D2 + y-like stores in front order", [=](hipStream_t s) { hipLaunchKernelGGL((astream_front_kernel<512, 8, 2, 1>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"front 512x512 PER8 D2 + y-like stores in front order", [=](hipStream_t s) { hipLaunchKernelGGL((astream_front_kernel<512, 8, 2, 1>), dim3(512), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"front 256x512 PER8 D3 + y-like stores in front order", [=](hipStream_t s) { hipLaunchKernelGGL((astream_front_kernel<512, 8, 3, 1>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        for (int R : {1, 2, 4, 8, 16, 72}) {
            calib.push_back({"rr 256x512 PER8 D2 R=" + std::to_string(R) + " no stores", [=](hipStream_t s) { hipLaunchKernelGGL((astream_rr_kernel<512, 8, 2, 0>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum, R); }});
            calib.push_back({"rr 256x512 PER8 D2 R=" + std::to_string(R) + " + y-like stores", [=](hipStream_t s) { hipLaunchKernelGGL((astream_rr_kernel<512, 8, 2, 1>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum, R); }});
        }
#define RRV(T_, PER_, D_, W_, R_) \
        calib.push_back({"rr W=" #W_ " T=" #T_ " PER" #PER_ " D" #D_ " R=" #R_ " no stores", [=](hipStream_t s) { hipLaunchKernelGGL((astream_rr_kernel<T_, PER_, D_, 0>), dim3(W_), dim3(T_), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum, R_); }}); \
        calib.push_back({"rr W=" #W_ " T=" #T_ " PER" #PER_ " D" #D_ " R=" #R_ " + stores", [=](hipStream_t s) { hipLaunchKernelGGL((astream_rr_kernel<T_, PER_, D_, 1>), dim3(W_), dim3(T_), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum, R_); }});
        RRV(512, 4, 2, 512, 8) RRV(512, 4, 2, 1024, 8) RRV(512, 4, 2, 1024, 1) RRV(512, 2, 2, 1024, 8) RRV(512, 2, 2, 2048, 8)
        RRV(256, 4, 2, 2048, 8) RRV(256, 8, 2, 1024, 8) RRV(256, 8, 1, 2048, 8) RRV(512, 4, 1, 1024, 8) RRV(512, 4, 1, 2048, 1)
        RRV(256, 4, 1, 4096, 1) RRV(256, 4, 1, 8192, 1)
        calib.push_back({"pipe 256x512 PER8 D3 +store", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 3, 2>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +ldsmeta+clamp", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 5>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 256x512 PER8 D2 +all", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<512, 8, 2, 7>), dim3(256), dim3(512), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
        calib.push_back({"pipe 512x256 PER8 D2 +all", [=](hipStream_t s) { hipLaunchKernelGGL((astream_pipe_kernel<256, 8, 2, 7>), dim3(512), dim3(256), 0, s, d_coef, ucol, (size_t)nnz, d_sink, ydum); }});
    }
    calib.push_back({"A-stream as 256 runs x512 thr (u32)", [=](hipStream_t s) { hipLaunchKernelGGL((astream_runs2_kernel<512, 32>), dim3(256), dim3(512), 0, s, d_coef, d_indcol, (size_t)nnz, d_sink); }});
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](Variant& v, int it) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < it; i++) v.launch(st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / it;
    };

    // correctness first
    std::vector<double> y(n);
    for (auto& v : vars) {
        CK(hipMemsetAsync(d_y, 0xFF, sizeof(double) * n, st));
        v.launch(st);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        CK(hipMemcpy(y.data(), d_y, sizeof(double) * n, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int i = 0; i < n; i++) bad += (memcmp(&y[i], &yref[i], 8) != 0);
        v.ok = (bad == 0);
        if (!v.ok) printf("  !! %s: %lld rows differ from the host fma chain\n", v.name.c_str(), bad);
    }
    // interleaved timing rounds
    for (auto& v : vars) time_it(v, 5);
    for (int r = 0; r < rounds; r++) {
        for (auto& v : vars) v.ms.push_back(time_it(v, iters));
        vcopy.ms.push_back(time_it(vcopy, 10));
        vastr.ms.push_back(time_it(vastr, 10));
        for (auto& v : calib) v.ms.push_back(time_it(v, 10));
    }
    // ---- interference experiment: the store-free skeleton (or any variant named by BGV) timed alone and
    // while a small kernel on another stream writes fresh lines
    if (const char* bgv = getenv("BGV")) {
        hipStream_t st2;
        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        double* d_bg = reinterpret_cast<double*>(d_b); // 1 GiB scratch
        for (auto& v : vars) {
            if (v.name.find(bgv) == std::string::npos) continue;
            float alone = time_it(v, iters);
            struct Cfg { int wg, sleep; };
            for (Cfg c : {Cfg{64, 1}, Cfg{64, 4}, Cfg{256, 4}, Cfg{256, 16}, Cfg{256, 64}}) {
                // throttled writers: 8 B per thread per ~sleep*1024 cycles; measure their own rate too
                hipEvent_t b0, b1;
                CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
                const size_t nwr = (size_t)c.wg * 256 * 600 / (c.sleep > 16 ? 8 : 1);
                CK(hipEventRecord(b0, st2));
                hipLaunchKernelGGL(bg_writer_kernel, dim3(c.wg), dim3(256), 0, st2, d_bg, nwr, c.sleep, 1);
                CK(hipEventRecord(b1, st2));
                float with_bg = time_it(v, iters);
                CK(hipStreamSynchronize(st2));
                float bms; CK(hipEventElapsedTime(&bms, b0, b1));
                printf("BG %-36s alone %.1f us | %3d writer WGs sleep %2d: %.1f us   (writers: %.1f MB in %.0f us = %.2f TB/s, main ran %.0f us)\n",
                       v.name.c_str(), alone * 1e3, c.wg, c.sleep, with_bg * 1e3, nwr * 8 / 1e6, bms * 1e3, nwr * 8 / (bms * 1e-3) / 1e12, with_bg * 1e3 * iters);
            }
        }
    }
    auto med = [](std::vector<float> a) { std::sort(a.begin(), a.end()); return a[a.size() / 2]; };
    auto mn = [](std::vector<float> a) { return *std::min_element(a.begin(), a.end()); };
    printf("%-44s %9s %9s %10s %8s %6s\n", "kernel", "med us", "min us", "GB/s(med)", "%8TB/s", "bits");
    for (auto& v : vars) {
        const double us = med(v.ms) * 1e3;
        printf("%-44s %9.1f %9.1f %10.1f %8.1f %6s\n", v.name.c_str(), us, mn(v.ms) * 1e3, B / us / 1e3, B / us / 1e3 / 80.0, v.ok ? "exact" : "WRONG");
    }
    {
        const double us = med(vcopy.ms) * 1e3;
        printf("%-44s %9.1f %9.1f %10.1f %8.1f\n", vcopy.name.c_str(), us, mn(vcopy.ms) * 1e3, 2.0 * copy_bytes / us / 1e3, 2.0 * copy_bytes / us / 1e3 / 80.0);
        const double us2 = med(vastr.ms) * 1e3;
        printf("%-44s %9.1f %9.1f %10.1f %8.1f\n", vastr.name.c_str(), us2, mn(vastr.ms) * 1e3, 12.0 * nnz / us2 / 1e3, 12.0 * nnz / us2 / 1e3 / 80.0);
    }
    if (g_prof_ptr) {
        unsigned long long h[16];
        CK(hipMemcpy(h, g_prof_ptr, sizeof h, hipMemcpyDeviceToHost));
        const char* names[5] = {"wait barrier1", "gather+stage", "issue loads", "wait barrier2", "ringwrite+reduce+store"};
        for (int w = 0; w < 2; w++) {
            double tot = 0;
            for (int i = 0; i < 5; i++) tot += (double)h[w * 8 + i];
            printf("ring5t phase shares, %s wave:", w == 0 ? "first (reduces rows)" : "last (no rows)");
            for (int i = 0; i < 5; i++) printf("  %s %.1f%%", names[i], 100.0 * h[w * 8 + i] / tot);
            printf("\n");
        }
    }
    if (g_prof_ptr2) {
        unsigned long long h[16];
        CK(hipMemcpy(h, g_prof_ptr2, sizeof h, hipMemcpyDeviceToHost));
        const char* names[6] = {"wait barrier1", "gather+stage", "issue loads", "wait barrier2", "ringwrite+reduce+store", "launch-to-loop"};
        for (int w = 0; w < 2; w++) {
            double tot = 0;
            for (int i = 0; i < 6; i++) tot += (double)h[w * 8 + i];
            printf("ring5t C16NT <256,2048> phase shares, %s wave:", w == 0 ? "first" : "last");
            for (int i = 0; i < 6; i++) printf("  %s %.1f%%", names[i], 100.0 * h[w * 8 + i] / tot);
            printf("   [clock64 ticks per workgroup-wave over all launches: %.0f]\n", tot / g_prof2_wgs);
        }
    }
    for (auto& v : calib) {
        const double us = med(v.ms) * 1e3;
        printf("%-44s %9.1f %9.1f %10.1f %8.1f\n", v.name.c_str(), us, mn(v.ms) * 1e3, 12.0 * nnz / us / 1e3, 12.0 * nnz / us / 1e3 / 80.0);
    }
    return 0;
}
