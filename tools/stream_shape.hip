// Dev tool: the ring kernel's MEMORY SKELETON at C4 — its loads and stores without the LDS work: 512 persistent workgroups, each walking a
// run of row blocks with four blocks of prefetch in registers (loads and waits written out, so every variant has the same pipeline).  Per
// block a workgroup reads 16 KB of values (non-temporal), 4 KB of 16-bit slots, 1 KB of row pointers and ~1 KB of x, and writes ~1 KB of y.
//   SEPARATE: values, slots and row pointers are three arrays (the library's layout; slots / row pointers loaded temporally)
//   PACKED:   one array, block after block [values | slots | row pointers] (21 504 B per block), everything non-temporal
//   y stores: none | plain | nt | sc1 | sc0 sc1 | 16-byte stores from half the lanes | four blocks' rows in one burst
// Warm (back to back) and cold (512 MiB fill + read sweep in front of every launch), three fresh sets of allocations.
// usage: stream_shape [blocks] [reps]      (profiles/r03_stream_shape.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int T = 256, D = 4, WGS = 512;
constexpr size_t kCoef = 16384, kSlots = 4096, kRows = 1024, kPacked = kCoef + kSlots + kRows, kVec = 1088;

// Loads and waits are written out (asm volatile) so that every variant has the SAME pipeline: the 7 loads of block lb + D are issued
// when block lb is consumed, and a block is consumed behind `s_waitcnt vmcnt(21 [+ 3 stores])` — its own loads done, three blocks'
// loads (and the last three y stores) still in flight.  The waited-for registers pass through the wait statement, so the compiler
// cannot use them in front of it.
#define LD16(dst, ptr, NTF) do { if (NTF) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(ptr) : "memory"); \
                                 else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory"); } while (0)
#define LD4(dst, ptr, NTF) do { if (NTF) asm volatile("global_load_dword %0, %1, off nt" : "=v"(dst) : "v"(ptr) : "memory"); \
                                else asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory"); } while (0)
#define LD8(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

template <bool PACKED, int STORE>
__global__ __launch_bounds__(T) void walk(const char* __restrict__ coef, const char* __restrict__ slots, const char* __restrict__ rows,
                                          const double* __restrict__ x, double* __restrict__ y, int nblk, int bpw, double* sink)
{
    const int bid = blockIdx.x, gw = (bid & 7) * (WGS / 8) + (bid >> 3), tid = threadIdx.x;
    const int b0 = min(nblk, gw * bpw), b1 = min(nblk, (gw + 1) * bpw), nb = b1 - b0;
    if (nb <= 0) return;
    d2 c[D][4];
    u4 sl[D];
    int pr[D];
    double xr[D];
    auto issue = [&](int lb, int s) {
        const size_t b = (size_t)min(b0 + lb, nblk - 1);
        const d2* cb = reinterpret_cast<const d2*>(PACKED ? coef + b * kPacked : coef + b * kCoef) + tid;
#pragma unroll
        for (int i = 0; i < 4; i++) LD16(c[s][i], cb + i * T, true);
        const u4* sp = reinterpret_cast<const u4*>(PACKED ? coef + b * kPacked + kCoef : slots + b * kSlots) + tid;
        const int* rp = reinterpret_cast<const int*>(PACKED ? coef + b * kPacked + kCoef + kSlots : rows + b * kRows) + tid;
        LD16(sl[s], sp, PACKED);
        LD4(pr[s], rp, PACKED);
        const double* xp = x + b * (kVec / 8) + (tid < 136 ? tid : 135);
        LD8(xr[s], xp);
    };
#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    double acc = 0.0, vb[D] = {0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s;
            // stores issued in the three iterations since this stage's loads: one each (modes 1-5), or the burst of four at s == 3 (mode 6)
#define WAITN(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(c[s][0]), "+v"(c[s][1]), "+v"(c[s][2]), "+v"(c[s][3]), "+v"(sl[s]), "+v"(pr[s]), "+v"(xr[s]) : : "memory")
            if (STORE == 0) WAITN(21);
            else if (STORE == 6) { if (s == 3) WAITN(21); else WAITN(25); }
            else WAITN(24);
            double v = xr[s] + (double)pr[s] + (double)(sl[s].x ^ sl[s].y ^ sl[s].z ^ sl[s].w);
#pragma unroll
            for (int i = 0; i < 4; i++) v += c[s][i].x + c[s][i].y;
            acc += v;
            vb[s] = v;
            // one store per lane and block, issued unconditionally (lanes past the rows write a scratch line behind y)
            double* yp = y + (size_t)(b0 + min(lb, nb - 1)) * (kVec / 8) + (tid < 136 ? tid : 136 + (tid & 7));
            if (STORE == 1) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(yp), "v"(v) : "memory");
            if (STORE == 2) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(yp), "v"(v) : "memory");
            if (STORE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(yp), "v"(v) : "memory");
            if (STORE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(yp), "v"(v) : "memory");
            if (STORE == 5) { // the same bytes as 16-byte stores from 68 lanes (the others a scratch line)
                const d2 vv = {v, v};
                double* yq = y + (size_t)(b0 + min(lb, nb - 1)) * (kVec / 8) + (tid < 68 ? 2 * tid : 136 + 2 * (tid & 3));
                asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(yq), "v"(vv) : "memory");
            }
            if (STORE == 6 && s == 3) { // four blocks' rows at once
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    double* yb = y + (size_t)(b0 + min(g + q, nb - 1)) * (kVec / 8) + (tid < 136 ? tid : 136 + (tid & 7));
                    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(yb), "v"(vb[q]) : "memory");
                }
            }
            issue(lb + D, s);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456) sink[0] = acc;
}

__global__ void sweep(const d2* p, size_t n16, double* sink)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) s += p[i].x;
    if (s == 123.456) sink[0] = s;
}

int main(int argc, char** argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 36765, reps = argc > 2 ? atoi(argv[2]) : 30;
    const int bpw = (nblk + WGS - 1) / WGS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* sink;
    CK(hipMalloc(&sink, 64));
    const size_t flush_bytes = 512u << 20;
    void* fl;
    CK(hipMalloc(&fl, flush_bytes));
    auto flush = [&]() {
        (void)hipMemsetAsync(fl, 1, flush_bytes, nullptr);
        hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, nullptr, (const d2*)fl, flush_bytes / 16, sink);
    };
    const size_t pad = 1 << 20;
    for (int draw = 0; draw < 3; draw++) {
        char *coef, *slots, *rows, *packed;
        double *x, *y;
        CK(hipMalloc(&coef, nblk * kCoef + pad));
        CK(hipMalloc(&slots, nblk * kSlots + pad));
        CK(hipMalloc(&rows, nblk * kRows + pad));
        CK(hipMalloc(&packed, nblk * kPacked + pad));
        CK(hipMalloc(&x, nblk * kVec + pad));
        CK(hipMalloc(&y, nblk * kVec + pad));
        CK(hipMemset(coef, 0, nblk * kCoef + pad));
        CK(hipMemset(slots, 0, nblk * kSlots + pad));
        CK(hipMemset(rows, 0, nblk * kRows + pad));
        CK(hipMemset(packed, 0, nblk * kPacked + pad));
        CK(hipMemset(x, 0, nblk * kVec + pad));
        CK(hipMemset(y, 0, nblk * kVec + pad));
        const char* snames[7] = {"reads", "+y plain", "+y nt", "+y sc1", "+y sc0sc1", "+y x4", "+y burst4"};
        auto run = [&](int mode, int st) {
#define L(P_, S_) hipLaunchKernelGGL((walk<P_, S_>), dim3(WGS), dim3(T), 0, nullptr, P_ ? packed : coef, slots, rows, x, y, nblk, bpw, sink)
            if (mode == 1) { if (st) L(true, 1); else L(true, 0); return; }
            switch (st) {
            case 0: L(false, 0); break;
            case 1: L(false, 1); break;
            case 2: L(false, 2); break;
            case 3: L(false, 3); break;
            case 4: L(false, 4); break;
            case 5: L(false, 5); break;
            default: L(false, 6); break;
            }
        };
        for (int mode = 0; mode < 2; mode++)
            for (int store = 0; store < (mode ? 2 : 7); store++) {
                for (int w = 0; w < 5; w++) run(mode, store);
                float ms;
                CK(hipEventRecord(e0, nullptr));
                for (int r = 0; r < reps; r++) run(mode, store);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                const float warm = ms * 1e3f / reps;
                std::vector<float> cold;
                for (int r = 0; r < 7; r++) {
                    flush();
                    CK(hipEventRecord(e0, nullptr));
                    run(mode, store);
                    CK(hipEventRecord(e1, nullptr));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    cold.push_back(ms * 1e3f);
                }
                std::sort(cold.begin(), cold.end());
                const double bytes = (double)nblk * (kPacked + kVec + (store ? kVec : 0));
                printf("SHAPE draw %d %-8s %-10s warm %6.1f us (%5.2f TB/s)  cold median %6.1f us (%5.2f TB/s)\n", draw, mode ? "PACKED" : "SEPARATE",
                       snames[store], warm, bytes / warm * 1e-6, cold[3], bytes / cold[3] * 1e-6);
                fflush(stdout);
            }
        CK(hipFree(coef)); CK(hipFree(slots)); CK(hipFree(rows)); CK(hipFree(packed)); CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
