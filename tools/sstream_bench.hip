// tools/sstream_bench.hip — PROTOTYPE (round 4): the sliced-stream idea of spmv_bcsr_sell.hpp for SCALAR rows.
// y = A x for a banded CSR matrix with a lane per ROW PAIR: slices of 128 rows, step j of a slice = the j-th nonzero of each of its rows,
// stored as 16 bytes per lane ({a(row 2l, j), a(row 2l+1, j)}: one contiguous KiB per wave-instruction, non-temporal) + one u32 per lane
// (two 13-bit LDS ring slots + flags); x lives in an LDS ring indexed by column that slides with the rows (one workgroup of four waves
// per CU, one wave per SIMD, the four on neighbouring slices of one 512-row round); each row is ONE fma chain in CSR order.
// No staging phase, no row-chain phase, every lane busy — what the ring kernel's per-block pipeline (2.04 us per 2048 nonzeros and
// workgroup) is not.  Measures the kernel at C2 / C4 shape and checks every bit against the host's fma chain.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/sstream_bench tools/sstream_bench.hip && ./tools/sstream_bench [rows] [w] [nnz per row]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kRing = 8192;          // x entries in the LDS ring (64 KB)
constexpr int kRound = 512;          // rows per round of a workgroup: 4 waves x 128 rows
constexpr int kNewMax = 1024;        // new window columns per round (4 per thread, prefetched a round ahead)
constexpr unsigned kPad = 0x8000u;   // slot flag: padding place (not multiplied)
constexpr unsigned kFirst = 0x4000u; // (low half only) first step of a slice
constexpr int kPadSteps = 64;

struct SsView {
    const v2d* val;       // [steps + pad][64]
    const unsigned* slot; // [steps + pad][64]: low half = row 2l, high half = row 2l + 1
    const int* wptr;      // [nwg * 4 + 1] first step of each wave's stream (streams of one workgroup's waves lie one after the other)
    const int* rptr;      // [nwg + 1] first round of each workgroup
    const int2* win;      // [rounds] {first new column, count}: what the window takes in before the round (first round of a workgroup: its first fill)
    int nwg, n;
};

constexpr int kPark = 20; // slices of y parked per wave (4 x 20 KB beside the 64 KB ring)
template <int D, bool NT, int ABL = 0, bool PARK = false>
__global__ __launch_bounds__(256) void spmv_sstream(SsView S, const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double ring[kRing];
    __shared__ v2d s_park[PARK ? 4 * kPark * 64 : 1];
    v2d* park = s_park + (PARK ? (threadIdx.x >> 6) * kPark * 64 + (threadIdx.x & 63) : 0);
    int parked = 0, park_first = 0; // (wave-uniform) rounds park_first .. park_first + parked - 1 are parked
    const int g = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int r_begin = S.rptr[g], r_end = S.rptr[g + 1];
    if (r_begin >= r_end) return;
    const int t0 = __builtin_amdgcn_readfirstlane(S.wptr[g * 4 + wv]);
    const int t_end = __builtin_amdgcn_readfirstlane(S.wptr[g * 4 + wv + 1]);
    const int clast = S.n - 1;
    // first fill of the window
    {
        const int2 w = S.win[r_begin];
        for (int c = w.x + tid; c < w.x + w.y; c += 256) ring[c & (kRing - 1)] = x[c];
    }
    int r = r_begin; // the round this wave's current slice belongs to
    // the next round's new columns, a round ahead in registers
    double nx[kNewMax / 256];
    int2 wn = S.win[min(r + 1, r_end - 1)];
#pragma unroll
    for (int u = 0; u < kNewMax / 256; u++) nx[u] = x[min(wn.x + tid + 256 * u, clast)];
    const v2d* vb = S.val + lane;
    const unsigned* sb = S.slot + lane;
    v2d a[D];
    unsigned sl[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        a[d] = NT ? __builtin_nontemporal_load(vb + (size_t)(t0 + d) * 64) : vb[(size_t)(t0 + d) * 64];
        sl[d] = sb[(size_t)(t0 + d) * 64];
    }
    __syncthreads();
    double acc0 = 0.0, acc1 = 0.0;
    auto store = [&](int round, v2d v) {
        const int row0 = round * kRound + wv * 128 + 2 * lane;
        if ((ABL & 2) && v.x != 123.456) return;
        if (row0 + 1 < S.n) *reinterpret_cast<v2d*>(y + row0) = v;
        else if (row0 < S.n) y[row0] = v.x;
    };
    auto flush = [&]() {
        for (int j = 0; j < parked; j++) store(park_first + j, park[j * 64]);
        parked = 0;
    };
    auto emit = [&]() { // this wave's slice of round r is complete
        if (!PARK) { store(r, v2d{acc0, acc1}); return; }
        if (parked == 0) park_first = r;
        park[parked * 64] = v2d{acc0, acc1};
        if (++parked == kPark) flush();
    };
    for (int t = t0; t < t_end; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int i = t + d;
            if (i < t_end) { // (wave-uniform)
                const unsigned s = sl[d];
                if ((__builtin_amdgcn_readfirstlane(s) & kFirst) && i != t0) { // this wave's slice of round r is complete
                    emit();
                    acc0 = acc1 = 0.0;
                    r++;
                    __syncthreads(); // every wave is through with round r - 1: the ring entries about to be overwritten are dead
#pragma unroll
                    for (int u = 0; u < kNewMax / 256; u++) {
                        const int c = wn.x + tid + 256 * u;
                        if (c < wn.x + wn.y) ring[c & (kRing - 1)] = nx[u];
                    }
                    __syncthreads();
                    wn = S.win[min(r + 1, r_end - 1)];
#pragma unroll
                    for (int u = 0; u < kNewMax / 256; u++) nx[u] = x[min(wn.x + tid + 256 * u, clast)];
                }
                const double x0 = (ABL & 1) ? 1.0 + lane : ring[s & (kRing - 1)], x1 = (ABL & 1) ? 0.5 : ring[(s >> 16) & (kRing - 1)];
                const double n0 = fma(a[d].x, x0, acc0), n1 = fma(a[d].y, x1, acc1);
                acc0 = (s & kPad) ? acc0 : n0;
                acc1 = (s & (kPad << 16)) ? acc1 : n1;
            }
            a[d] = NT ? __builtin_nontemporal_load(vb + (size_t)(i + D) * 64) : vb[(size_t)(i + D) * 64];
            const unsigned* sp = sb + (size_t)(i + D) * 64;
            asm volatile("" ::"v"(sp));
            sl[d] = *sp;
        }
    }
    emit();
    if (PARK) flush();
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static inline uint64_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

template <int D, bool NT, int ABL, bool PARK = false>
static double run(const SsView& S, const double* x, double* y, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((spmv_sstream<D, NT, ABL, PARK>), dim3(S.nwg), dim3(256), 0, nullptr, S, x, y);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((spmv_sstream<D, NT, ABL, PARK>), dim3(S.nwg), dim3(256), 0, nullptr, S, x, y);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3 / reps;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, w = argc > 2 ? atoi(argv[2]) : 2000, per = argc > 3 ? atoi(argv[3]) : 15;
    // S15-like band: the diagonal + (per - 1) distinct columns in [i - w, i + w], ascending
    std::vector<int> ptrow(n + 1, 0), indcol;
    std::vector<double> coef;
    indcol.reserve((size_t)n * per);
    coef.reserve((size_t)n * per);
    std::vector<int> cols;
    for (int i = 0; i < n; i++) {
        cols.assign(1, i);
        while ((int)cols.size() < per) {
            const int c = i - w + (int)(rnd() % (2 * w + 1));
            if (c < 0 || c >= n) continue;
            if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
        }
        std::sort(cols.begin(), cols.end());
        for (int c : cols) {
            indcol.push_back(c);
            coef.push_back(c == i ? 1.0 : ((double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1) / per);
        }
        ptrow[i + 1] = (int)indcol.size();
    }
    const long long nnz = indcol.size();
    // plan: rounds of 512 rows dealt to 256 workgroups in contiguous ranges; per workgroup and wave one stream
    const int nwg = 256, rounds = (n + kRound - 1) / kRound;
    std::vector<int> rptr(nwg + 1);
    for (int g = 0; g <= nwg; g++) rptr[g] = (int)((long long)rounds * g / nwg);
    std::vector<int2> win(rounds);
    std::vector<int> wptr((size_t)nwg * 4 + 1, 0);
    std::vector<v2d> val;
    std::vector<unsigned> slot;
    bool eligible = true;
    long long pad_places = 0;
    for (int g = 0; g < nwg; g++) {
        // windows
        int whi = 0;
        std::vector<int> cmin(rptr[g + 1] - rptr[g]), cmax(rptr[g + 1] - rptr[g]);
        for (int r = rptr[g]; r < rptr[g + 1]; r++) {
            int lo = 0x7fffffff, hi = -1;
            for (int i = r * kRound; i < std::min(n, (r + 1) * kRound); i++)
                if (ptrow[i + 1] > ptrow[i]) { lo = std::min(lo, indcol[ptrow[i]]); hi = std::max(hi, indcol[ptrow[i + 1] - 1]); }
            cmin[r - rptr[g]] = lo; cmax[r - rptr[g]] = hi;
        }
        int allmin = 0x7fffffff;
        for (int v : cmin) allmin = std::min(allmin, v);
        for (int r = rptr[g]; r < rptr[g + 1]; r++) {
            const int k = r - rptr[g];
            const int nhi = std::max(whi, cmax[k] + 1);
            if (k == 0) {
                const int lo = std::max(std::max(0, nhi - kRing), std::min(allmin, nhi));
                win[r] = make_int2(lo, nhi - lo);
            } else {
                win[r] = make_int2(whi, nhi - whi);
                if (nhi - whi > kNewMax) eligible = false;
            }
            whi = nhi;
            if (cmin[k] != 0x7fffffff && cmin[k] < whi - kRing) eligible = false;
        }
        // streams
        for (int wv = 0; wv < 4; wv++) {
            wptr[(size_t)g * 4 + wv] = (int)(val.size() / 64);
            for (int r = rptr[g]; r < rptr[g + 1]; r++) {
                const int row0 = r * kRound + wv * 128;
                int L = 1;
                for (int i = row0; i < std::min(n, row0 + 128); i++) L = std::max(L, ptrow[i + 1] - ptrow[i]);
                for (int j = 0; j < L; j++)
                    for (int l = 0; l < 64; l++) {
                        v2d v = {0.0, 0.0};
                        unsigned s = 0;
                        for (int h = 0; h < 2; h++) {
                            const int i = row0 + 2 * l + h;
                            unsigned sh = kPad;
                            if (i < n && j < ptrow[i + 1] - ptrow[i]) {
                                const int k = ptrow[i] + j;
                                if (h == 0) v.x = coef[k]; else v.y = coef[k];
                                sh = (unsigned)(indcol[k] & (kRing - 1));
                            } else pad_places++;
                            s |= sh << (16 * h);
                        }
                        if (j == 0) s |= kFirst;
                        val.push_back(v);
                        slot.push_back(s);
                    }
            }
        }
    }
    wptr[(size_t)nwg * 4] = (int)(val.size() / 64);
    const long long steps = val.size() / 64;
    for (int k = 0; k < kPadSteps * 64; k++) { val.push_back(v2d{0.0, 0.0}); slot.push_back(kPad | (kPad << 16) | kFirst); }
    printf("n %d nnz %lld  steps %lld  padding places %.3f %%  eligible %d  stream bytes %.1f MB (%.2f B per nonzero)\n", n, nnz, steps, 100.0 * pad_places / nnz, (int)eligible,
           steps * 64 * 20.0 / 1e6, steps * 64 * 20.0 / nnz);
    if (!eligible) { printf("window does not fit: not eligible\n"); return 1; }
    std::vector<double> hx(n), href(n);
    for (int i = 0; i < n; i++) hx[i] = sin(0.001 * i);
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) s = fma(coef[k], hx[indcol[k]], s);
        href[i] = s;
    }
    v2d* d_val; unsigned* d_slot; int *d_wptr, *d_rptr; int2* d_win; double *d_x, *d_y;
    CK(hipMalloc(&d_val, sizeof(v2d) * val.size()));
    CK(hipMemcpy(d_val, val.data(), sizeof(v2d) * val.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_slot, sizeof(unsigned) * slot.size()));
    CK(hipMemcpy(d_slot, slot.data(), sizeof(unsigned) * slot.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_wptr, sizeof(int) * wptr.size()));
    CK(hipMemcpy(d_wptr, wptr.data(), sizeof(int) * wptr.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_rptr, sizeof(int) * rptr.size()));
    CK(hipMemcpy(d_rptr, rptr.data(), sizeof(int) * rptr.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_win, sizeof(int2) * win.size()));
    CK(hipMemcpy(d_win, win.data(), sizeof(int2) * win.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, sizeof(double) * n));
    CK(hipMemcpy(d_x, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, sizeof(double) * (n + 2)));
    SsView S{d_val, d_slot, d_wptr, d_rptr, d_win, nwg, n};
    const double B = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    auto line = [&](const char* name, double us) { printf("%-44s %8.2f us   %6.0f GB/s algorithmic  (%.3f of 8 TB/s)\n", name, us, B / us / 1e3, B / us / 1e3 / 8000); };
    auto check = [&](const char* name) {
        std::vector<double> hy(n);
        CK(hipMemcpy(hy.data(), d_y, sizeof(double) * n, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int i = 0; i < n; i++) bad += memcmp(&hy[i], &href[i], 8) != 0;
        printf("    %s: %lld of %d rows differ bitwise from the host's fma chain\n", name, bad, n);
    };
    const int R = 50;
    CK(hipMemset(d_y, 0xff, sizeof(double) * n));
    line("sliced stream  D=8 nt", run<8, true, 0>(S, d_x, d_y, R));
    check("D=8 nt");
    CK(hipMemset(d_y, 0xff, sizeof(double) * n));
    line("sliced stream  D=6 nt", run<6, true, 0>(S, d_x, d_y, R));
    check("D=6 nt");
    CK(hipMemset(d_y, 0xff, sizeof(double) * n));
    line("sliced stream, y parked in LDS  D=8 nt", run<8, true, 0, true>(S, d_x, d_y, R));
    check("parked D=8 nt");
    line("sliced stream, y parked in LDS  D=8 temporal", run<8, false, 0, true>(S, d_x, d_y, R));
    line("sliced stream, y parked in LDS  D=12 nt", run<12, true, 0, true>(S, d_x, d_y, R));
    line("sliced stream  D=12 nt", run<12, true, 0>(S, d_x, d_y, R));
    line("sliced stream  D=8 temporal", run<8, false, 0>(S, d_x, d_y, R));
    line("  no LDS gather (invalid)  D=8 nt", run<8, true, 1>(S, d_x, d_y, R));
    line("  no y stores (invalid)    D=8 nt", run<8, true, 2>(S, d_x, d_y, R));
    line("  neither (invalid)        D=8 nt", run<8, true, 3>(S, d_x, d_y, R));
    return 0;
}
