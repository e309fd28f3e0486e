// membw.hip — streaming-bandwidth ceilings of this MI355X for the access shapes the CG kernels use (16 B per lane).
// Not part of the product: explains the in-run ceilings bench.py reports.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 tools/membw.hip -o tools/build/membw && tools/build/membw [MiB per vector]
// Variants: grid size (workgroups per CU, or one pass with no grid-stride loop), unroll (16-B accesses in flight per lane), plain /
// non-temporal stores and loads.  Output: one line per variant, TB/s of algorithmic bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NTL> __device__ __forceinline__ dbl2 ld(const dbl2* p) {
    if (NTL) return __builtin_nontemporal_load(p);
    return *p;
}
template <int NTS> __device__ __forceinline__ void st(dbl2* p, dbl2 v) {
    if (NTS) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// OP: 0 copy (1R 1W), 1 read (1R), 2 write (1W), 3 triad (2R 1W), 4 axpy in place (2R 1W, one of them RMW)
template <int OP, int U, int NTL, int NTS>
__global__ __launch_bounds__(256) void k_stream(const dbl2* __restrict__ a, const dbl2* __restrict__ b, dbl2* __restrict__ c, long long n2, double* sink) {
    const long long stride = (long long)gridDim.x * 256;
    double acc = 0.0;
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; e + (U - 1) * stride < n2; e += U * stride) {
        dbl2 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (OP != 2) va[u] = ld<NTL>(a + e + u * stride);
            if (OP == 3 || OP == 4) vb[u] = ld<NTL>((OP == 4 ? (const dbl2*)c : b) + e + u * stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (OP == 0) st<NTS>(c + e + u * stride, va[u]);
            else if (OP == 1) acc += va[u].x + va[u].y;
            else if (OP == 2) st<NTS>(c + e + u * stride, dbl2{1.5, 2.5});
            else st<NTS>(c + e + u * stride, dbl2{va[u].x + 0.5 * vb[u].x, va[u].y + 0.5 * vb[u].y});
        }
    }
    for (; e < n2; e += stride) {
        if (OP == 0) st<NTS>(c + e, ld<NTL>(a + e));
        else if (OP == 1) { dbl2 v = ld<NTL>(a + e); acc += v.x + v.y; }
        else if (OP == 2) st<NTS>(c + e, dbl2{1.5, 2.5});
        else { dbl2 v = ld<NTL>(a + e), w = ld<NTL>((OP == 4 ? (const dbl2*)c : b) + e); st<NTS>(c + e, dbl2{v.x + 0.5 * w.x, v.y + 0.5 * w.y}); }
    }
    if (OP == 1 && acc == 1.2345e300) sink[0] = acc;
}

struct Buf { dbl2 *a, *b, *c; double* sink; long long n2; };

template <int OP, int U, int NTL, int NTS>
double run(const Buf& B, int grid, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_stream<OP, U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, B.a, B.b, B.c, B.n2, B.sink);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<OP, U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, B.a, B.b, B.c, B.n2, B.sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reps;
}

template <int OP, int U>
void sweep_nt(const Buf& B, const char* name, int nvec, int grid, const char* gname) {
    const double bytes = (double)nvec * B.n2 * 16.0;
    const double t00 = run<OP, U, 0, 0>(B, grid, 10), t01 = run<OP, U, 0, 1>(B, grid, 10), t10 = run<OP, U, 1, 0>(B, grid, 10), t11 = run<OP, U, 1, 1>(B, grid, 10);
    printf("%-8s U=%d grid=%-9s  plain %6.3f ms %5.2f TB/s | nt-store %6.3f ms %5.2f | nt-load %6.3f ms %5.2f | nt both %6.3f ms %5.2f\n", name, U, gname, t00,
           bytes / t00 / 1e9, t01, bytes / t01 / 1e9, t10, bytes / t10 / 1e9, t11, bytes / t11 / 1e9);
    fflush(stdout);
}

template <int OP>
void sweep(const Buf& B, const char* name, int nvec, int ncu) {
    const long long full = (B.n2 + 255) / 256;
    struct { int g; const char* n; } grids[] = {{2 * ncu, "2/CU"}, {4 * ncu, "4/CU"}, {8 * ncu, "8/CU"}, {16 * ncu, "16/CU"}, {32 * ncu, "32/CU"}};
    for (auto& g : grids) {
        sweep_nt<OP, 1>(B, name, nvec, g.g, g.n);
        sweep_nt<OP, 2>(B, name, nvec, g.g, g.n);
        sweep_nt<OP, 4>(B, name, nvec, g.g, g.n);
    }
    // one element per thread (no loop): as many workgroups as elements / 256 (U = 1) or / 1024 (U = 4)
    if (full < 0x7fffffffLL) {
        sweep_nt<OP, 1>(B, name, nvec, (int)full, "full");
        sweep_nt<OP, 4>(B, name, nvec, (int)((full + 3) / 4), "full/4");
    }
}

int main(int argc, char** argv) {
    const long long mib = argc > 1 ? atoll(argv[1]) : 1059;        // 513^2 x 528 doubles = 1.11 GB: one CG vector of config 3
    Buf B;
    B.n2 = mib * (1LL << 20) / 16;
    CK(hipMalloc(&B.a, B.n2 * 16));
    CK(hipMalloc(&B.b, B.n2 * 16));
    CK(hipMalloc(&B.c, B.n2 * 16));
    CK(hipMalloc(&B.sink, 64));
    CK(hipMemset(B.a, 0, B.n2 * 16));
    CK(hipMemset(B.b, 0, B.n2 * 16));
    CK(hipMemset(B.c, 0, B.n2 * 16));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, %lld MiB per vector\n", prop.name, ncu, mib);
    // hipMemcpyAsync D2D as a further reference point
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipMemcpyAsync(B.c, B.a, B.n2 * 16, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) CK(hipMemcpyAsync(B.c, B.a, B.n2 * 16, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpyAsync D2D: %.3f ms  %.2f TB/s (1R 1W)\n", ms / 10, 2.0 * B.n2 * 16 / (ms / 10) / 1e9);
    }
    sweep<0>(B, "copy", 2, ncu);
    sweep<1>(B, "read", 1, ncu);
    sweep<2>(B, "write", 1, ncu);
    sweep<3>(B, "triad", 3, ncu);
    sweep<4>(B, "axpy", 3, ncu);
    return 0;
}
