// Probe of the elasticity PCG's update pass at config 5's size (257 x 65 x 65 nodes, 3 components, vectors of 27.6 MB: everything inside the
// Infinity Cache): r -= alpha q; z = D^-1 r; x += alpha p_old; p_new = z + beta p_old; partial r.z, r.r  — 7 vector passes, 193 MB.
// Thread mappings and load schedules side by side, each alternating with a stand-in for the apply (q = 0.999 p_new) so that the cache
// state is the loop's.  build: hipcc -O3 --offload-arch=gfx950 tools/el_update_probe.hip -o gpurun_out/el_update_probe
// run:   rocprofv3 --kernel-trace --stats -d gpurun_out/upd_probe -- gpurun_out/el_update_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Geo { int NX, NY, NZ, PX; long long PP, stride; };
struct Args {
    Geo g;
    const double* q; double* r; const double* p_old; double* p_new; double* x;
    const double* dinv; const double* scal; double* part; double* part2;
};

__device__ __forceinline__ int axis_class(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 2 : 1); }
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* s) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s[w];
    return t;
}
__device__ __forceinline__ void scalars(const double* sc, double* alpha, double* beta) {
    const double pq = sc[0], rz = sc[1], qmq = sc[3];
    *alpha = (pq > 0.0 && rz > 0.0) ? rz / pq : 0.0;
    const double rzn = *alpha * *alpha * qmq - rz;
    *beta = (*alpha > 0.0 && rzn > 0.0) ? rzn / rz : 0.0;
}

// V0: the library's form — grid-stride over node pairs, incremental (i2, j, k), all 12 loads of a pair before its 9 stores
__global__ __launch_bounds__(256) void upd_v0(Args a) {
    __shared__ double sRed[4];
    __shared__ double sD[81];
    if (threadIdx.x < 81) sD[threadIdx.x] = a.dinv[threadIdx.x];
    __syncthreads();
    double alpha, beta;
    scalars(a.scal, &alpha, &beta);
    const Geo& g = a.g;
    const long long pairs = (long long)g.NZ * g.PP / 2;
    const int PX2 = g.PX / 2;
    const long long step = (long long)gridDim.x * 256;
    const int di = (int)(step % PX2);
    const long long drow = step / PX2;
    const int dj = (int)(drow % g.NY), dk = (int)(drow / g.NY);
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    int i2 = (int)(e % PX2), j = (int)((e / PX2) % g.NY), k = (int)((e / PX2) / g.NY);
    double arz = 0.0, arr = 0.0;
    for (; e < pairs; e += step) {
        const long long idx = 2 * e;
        const int i = 2 * i2;
        const int cyz = 3 * axis_class(j, g.NY) + 9 * axis_class(k, g.NZ);
        const int c0 = axis_class(i < g.NX ? i : 1, g.NX) + cyz, c1 = axis_class(i + 1 < g.NX ? i + 1 : 1, g.NX) + cyz;
        i2 += di; j += dj; k += dk;
        if (i2 >= PX2) { i2 -= PX2; ++j; }
        if (j >= g.NY) { j -= g.NY; ++k; }
        double2 po[3], rv[3], qv[3], xv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            po[c] = *(const double2*)(a.p_old + c * g.stride + idx);
            rv[c] = *(const double2*)(a.r + c * g.stride + idx);
            qv[c] = *(const double2*)(a.q + c * g.stride + idx);
            xv[c] = *(const double2*)(a.x + c * g.stride + idx);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rv[c].x -= alpha * qv[c].x; rv[c].y -= alpha * qv[c].y;
            *(double2*)(a.r + c * g.stride + idx) = rv[c];
            xv[c].x += alpha * po[c].x; xv[c].y += alpha * po[c].y;
            *(double2*)(a.x + c * g.stride + idx) = xv[c];
            const double2 zv = make_double2(sD[c0 * 3 + c] * rv[c].x, sD[c1 * 3 + c] * rv[c].y);
            arz += rv[c].x * zv.x + rv[c].y * zv.y;
            arr += rv[c].x * rv[c].x + rv[c].y * rv[c].y;
            *(double2*)(a.p_new + c * g.stride + idx) = make_double2(zv.x + beta * po[c].x, zv.y + beta * po[c].y);
        }
    }
    const double t1 = block_sum<4>(arz, sRed), t2 = block_sum<4>(arr, sRed);
    if (threadIdx.x == 0) { a.part[blockIdx.x] = t1; a.part2[blockIdx.x] = t2; }
}

// V1: V0 with the loads of the next pair requested before the stores of the current one
__global__ __launch_bounds__(256) void upd_v1(Args a) {
    __shared__ double sRed[4];
    __shared__ double sD[81];
    const Geo& g = a.g;
    const long long pairs = (long long)g.NZ * g.PP / 2;
    const int PX2 = g.PX / 2;
    const long long step = (long long)gridDim.x * 256;
    const int di = (int)(step % PX2);
    const long long drow = step / PX2;
    const int dj = (int)(drow % g.NY), dk = (int)(drow / g.NY);
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    int i2 = (int)(e % PX2), j = (int)((e / PX2) % g.NY), k = (int)((e / PX2) / g.NY);
    double2 po[3], rv[3], qv[3], xv[3];
    auto load = [&](long long idx, double2* P, double2* R, double2* Q, double2* X) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            P[c] = *(const double2*)(a.p_old + c * g.stride + idx);
            R[c] = *(const double2*)(a.r + c * g.stride + idx);
            Q[c] = *(const double2*)(a.q + c * g.stride + idx);
            X[c] = *(const double2*)(a.x + c * g.stride + idx);
        }
    };
    load(e < pairs ? 2 * e : 0, po, rv, qv, xv);
    asm volatile("" ::: "memory");
    if (threadIdx.x < 81) sD[threadIdx.x] = a.dinv[threadIdx.x];
    __syncthreads();
    double alpha, beta;
    scalars(a.scal, &alpha, &beta);
    double arz = 0.0, arr = 0.0;
    while (e < pairs) {
        const long long idx = 2 * e;
        const int i = 2 * i2;
        const int cyz = 3 * axis_class(j, g.NY) + 9 * axis_class(k, g.NZ);
        const int c0 = axis_class(i < g.NX ? i : 1, g.NX) + cyz, c1 = axis_class(i + 1 < g.NX ? i + 1 : 1, g.NX) + cyz;
        i2 += di; j += dj; k += dk;
        if (i2 >= PX2) { i2 -= PX2; ++j; }
        if (j >= g.NY) { j -= g.NY; ++k; }
        e += step;
        double2 po2[3], rv2[3], qv2[3], xv2[3];
        load(e < pairs ? 2 * e : 0, po2, rv2, qv2, xv2);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rv[c].x -= alpha * qv[c].x; rv[c].y -= alpha * qv[c].y;
            *(double2*)(a.r + c * g.stride + idx) = rv[c];
            xv[c].x += alpha * po[c].x; xv[c].y += alpha * po[c].y;
            *(double2*)(a.x + c * g.stride + idx) = xv[c];
            const double2 zv = make_double2(sD[c0 * 3 + c] * rv[c].x, sD[c1 * 3 + c] * rv[c].y);
            arz += rv[c].x * zv.x + rv[c].y * zv.y;
            arr += rv[c].x * rv[c].x + rv[c].y * rv[c].y;
            *(double2*)(a.p_new + c * g.stride + idx) = make_double2(zv.x + beta * po[c].x, zv.y + beta * po[c].y);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { po[c] = po2[c]; rv[c] = rv2[c]; qv[c] = qv2[c]; xv[c] = xv2[c]; }
    }
    const double t1 = block_sum<4>(arz, sRed), t2 = block_sum<4>(arr, sRed);
    if (threadIdx.x == 0) { a.part[blockIdx.x] = t1; a.part2[blockIdx.x] = t2; }
}

// V2: PPT node pairs per thread, no grid stride: workgroup b takes the pairs [b * 256 * PPT, ...), thread t the pairs t, t + 256, ...; 32-bit index
// arithmetic; all loads of the thread's pairs before its first store; COMPS = 3: the three components of a pair in one thread, 1: blockIdx.y = component
template <int PPT, int COMPS, int NTH>
__global__ __launch_bounds__(NTH) void upd_v2(Args a) {
    __shared__ double sRed[NTH / 64];
    __shared__ double sD[81];
    const Geo& g = a.g;
    const unsigned pairs = (unsigned)((long long)g.NZ * g.PP / 2);
    const unsigned PX2 = (unsigned)g.PX / 2;
    const int cb = COMPS == 3 ? 0 : (int)blockIdx.y;
    unsigned e[PPT];
    double2 po[PPT][COMPS], rv[PPT][COMPS], qv[PPT][COMPS], xv[PPT][COMPS];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        e[u] = (blockIdx.x * PPT + u) * NTH + threadIdx.x;
        const long long idx = e[u] < pairs ? 2LL * e[u] : 0;
#pragma unroll
        for (int c = 0; c < COMPS; ++c) {
            po[u][c] = *(const double2*)(a.p_old + (c + cb) * g.stride + idx);
            rv[u][c] = *(const double2*)(a.r + (c + cb) * g.stride + idx);
            qv[u][c] = *(const double2*)(a.q + (c + cb) * g.stride + idx);
            xv[u][c] = *(const double2*)(a.x + (c + cb) * g.stride + idx);
        }
    }
    asm volatile("" ::: "memory");
    if (threadIdx.x < 81) sD[threadIdx.x] = a.dinv[threadIdx.x];
    __syncthreads();
    double alpha, beta;
    scalars(a.scal, &alpha, &beta);
    double arz = 0.0, arr = 0.0;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        if (e[u] < pairs) {
            const unsigned row = e[u] / PX2, i2 = e[u] - row * PX2;
            const unsigned k = row / (unsigned)g.NY, j = row - k * (unsigned)g.NY;
            const int i = 2 * (int)i2;
            const int cyz = 3 * axis_class((int)j, g.NY) + 9 * axis_class((int)k, g.NZ);
            const int c0 = axis_class(i < g.NX ? i : 1, g.NX) + cyz, c1 = axis_class(i + 1 < g.NX ? i + 1 : 1, g.NX) + cyz;
            const long long idx = 2LL * e[u];
#pragma unroll
            for (int c = 0; c < COMPS; ++c) {
                double2 R = rv[u][c], X = xv[u][c];
                const double2 P = po[u][c], Q = qv[u][c];
                R.x -= alpha * Q.x; R.y -= alpha * Q.y;
                *(double2*)(a.r + (c + cb) * g.stride + idx) = R;
                X.x += alpha * P.x; X.y += alpha * P.y;
                *(double2*)(a.x + (c + cb) * g.stride + idx) = X;
                const double2 zv = make_double2(sD[c0 * 3 + c + cb] * R.x, sD[c1 * 3 + c + cb] * R.y);
                arz += R.x * zv.x + R.y * zv.y;
                arr += R.x * R.x + R.y * R.y;
                *(double2*)(a.p_new + (c + cb) * g.stride + idx) = make_double2(zv.x + beta * P.x, zv.y + beta * P.y);
            }
        }
    }
    const double t1 = block_sum<NTH / 64>(arz, sRed), t2 = block_sum<NTH / 64>(arr, sRed);
    if (threadIdx.x == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x;
        a.part[b] = t1; a.part2[b] = t2;
    }
}

// stand-in for the apply between two update passes: reads p_new, writes q (same bytes as the apply's algorithmic traffic)
__global__ __launch_bounds__(256) void fake_apply(const double* p, double* q, long long n2) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n2; e += (long long)gridDim.x * 256) {
        const double2 v = *(const double2*)(p + 2 * e);
        *(double2*)(q + 2 * e) = make_double2(0.999 * v.x, 0.999 * v.y);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const int nzp = argc > 2 ? atoi(argv[2]) : 65;            // node planes (65 = config 5; fewer: a smaller working set)
    const bool inplace = argc > 3 && atoi(argv[3]) != 0;      // p updated in place (one p vector instead of the loop's two)
    Geo g;
    g.NX = 257; g.NY = 65; g.NZ = nzp; g.PX = 272;
    g.PP = (long long)g.PX * g.NY;
    g.stride = g.PP * g.NZ + 4096;
    const size_t vec = (size_t)3 * g.stride;
    double *x, *r, *p0, *p1, *q, *dinv, *scal, *part, *part2;
    for (double** v : {&x, &r, &p0, &p1, &q}) {
        CK(hipMalloc(v, (vec + 4096) * sizeof(double)));
        std::vector<double> h(vec + 4096);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000);
        CK(hipMemcpy(*v, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&dinv, 81 * sizeof(double)));
    CK(hipMalloc(&scal, 8 * sizeof(double)));
    CK(hipMalloc(&part, 65536 * sizeof(double)));
    CK(hipMalloc(&part2, 65536 * sizeof(double)));
    {
        std::vector<double> d(81, 0.5);
        CK(hipMemcpy(dinv, d.data(), 81 * sizeof(double), hipMemcpyHostToDevice));
        const double sc[8] = {2.0, 1e-3, 1.0, 1e3, 0, 0, 0, 0};      // alpha = 5e-4, beta small: values stay finite over the repetitions
        CK(hipMemcpy(scal, sc, sizeof(sc), hipMemcpyHostToDevice));
    }
    const long long pairs = (long long)g.NZ * g.PP / 2;
    const long long n2 = (long long)vec / 2;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int cur = 0;
    auto args = [&]() {
        Args a;
        a.g = g; a.q = q; a.r = r; a.p_old = cur ? p1 : p0; a.p_new = inplace ? (cur ? p1 : p0) : (cur ? p0 : p1); a.x = x;
        a.dinv = dinv; a.scal = scal; a.part = part; a.part2 = part2;
        return a;
    };
    auto run = [&](const char* name, auto launch) {
        float only_apply = 0.f, both = 0.f;
        for (int phase = 0; phase < 2; ++phase) {
            for (int w = 0; w < 10; ++w) {
                if (phase) { launch(args()); if (!inplace) cur ^= 1; }
                hipLaunchKernelGGL(fake_apply, dim3(2048), dim3(256), 0, st, cur ? p1 : p0, q, n2);
            }
            CK(hipEventRecord(e0, st));
            for (int w = 0; w < reps; ++w) {
                if (phase) { launch(args()); if (!inplace) cur ^= 1; }
                hipLaunchKernelGGL(fake_apply, dim3(2048), dim3(256), 0, st, cur ? p1 : p0, q, n2);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(phase ? &both : &only_apply, e0, e1));
        }
        const double us = (both - only_apply) * 1e3 / reps;
        printf("%-46s %7.2f us per update pass (+ stand-in apply %.2f us)  = %.2f TB/s of the 7 vector passes\n", name, us, only_apply * 1e3 / reps,
               7.0 * 3 * g.PP * g.NZ * 8 / (us * 1e-6) / 1e12);
        fflush(stdout);
    };
    printf("%d node planes: vectors of %.1f MB, %s\n", nzp, 3.0 * g.PP * g.NZ * 8 / 1e6, inplace ? "p in place (5 vectors)" : "two p vectors (6 vectors)");
    for (int mult : {4}) {
        char nm[96];
        snprintf(nm, sizeof nm, "v0 grid-stride, %d workgroups", mult * 256);
        run(nm, [&](Args a) { hipLaunchKernelGGL(upd_v0, dim3(mult * 256), dim3(256), 0, st, a); });
        snprintf(nm, sizeof nm, "v1 grid-stride + prefetch, %d workgroups", mult * 256);
        run(nm, [&](Args a) { hipLaunchKernelGGL(upd_v1, dim3(mult * 256), dim3(256), 0, st, a); });
    }
    auto nb = [&](int ppt, int nth) { return (unsigned)((pairs + (long long)ppt * nth - 1) / ((long long)ppt * nth)); };
    run("v2 1 pair x 3 comps per thread, 256 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<1, 3, 256>), dim3(nb(1, 256)), dim3(256), 0, st, a); });
    run("v2 2 pairs x 3 comps per thread, 256 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<2, 3, 256>), dim3(nb(2, 256)), dim3(256), 0, st, a); });
    run("v2 1 pair x 1 comp per thread, 256 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<1, 1, 256>), dim3(nb(1, 256), 3), dim3(256), 0, st, a); });
    run("v2 2 pairs x 1 comp per thread, 256 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<2, 1, 256>), dim3(nb(2, 256), 3), dim3(256), 0, st, a); });
    run("v2 4 pairs x 1 comp per thread, 256 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<4, 1, 256>), dim3(nb(4, 256), 3), dim3(256), 0, st, a); });
    run("v2 1 pair x 3 comps per thread, 512 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<1, 3, 512>), dim3(nb(1, 512)), dim3(512), 0, st, a); });
    run("v2 1 pair x 3 comps per thread, 1024 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<1, 3, 1024>), dim3(nb(1, 1024)), dim3(1024), 0, st, a); });
    run("v2 2 pairs x 1 comp per thread, 1024 thr", [&](Args a) { hipLaunchKernelGGL((upd_v2<2, 1, 1024>), dim3(nb(2, 1024), 3), dim3(1024), 0, st, a); });
    return 0;
}
