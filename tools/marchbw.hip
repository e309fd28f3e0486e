// marchbw.hip — the memory access pattern of the elasticity marching kernel without its arithmetic: which layout / tile shape moves
// bytes fastest on this MI355X?  Not part of the product.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 tools/marchbw.hip -o tools/build/marchbw && tools/build/marchbw
// A wavefront owns XT columns x RY rows and walks ZLEN planes; per plane it loads its RY rows of 3 components and stores RY rows of 3
// components (1 read + 1 write per node and component = the algorithmic traffic of the apply; halo rows are L2 hits in the real kernel and
// are left out).  Layouts: SoA = three component arrays, 8 B per lane and load (what k_el_fast does); AoS = one array of [node][3], 24 B per
// lane as 16 + 8.  Tiles: 62 columns (window of 64 lanes, 496-byte stride: unaligned), 64 (aligned; no x-halo lanes — not a correct stencil
// tiling, the upper bound of alignment) and 48 (aligned, 3/4 of the lanes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Geo { int NX, NY, NZ, PX; long long PP, stride; int ntx, nty, nzc, zlen, xt; };

// W16: SoA with TWO adjacent nodes per lane (16-byte loads and stores; the tile is 2 x xt columns, xt = 62 / 64 / 48 as above)
template <int RY, bool NTS>
__global__ __launch_bounds__(256) void k_march16(Geo g, const double* __restrict__ in, double* __restrict__ out) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bm = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) bm = (bm & 7) * (G >> 3) + (bm >> 3);
    const int nyg = (g.nty + 3) / 4;
    const int xt = bm % g.ntx, yt = ((bm / g.ntx) % nyg) * 4 + wv, zc = bm / (g.ntx * nyg);
    if (yt >= g.nty || zc >= g.nzc) return;
    const int i = xt * 2 * g.xt + 2 * lane;
    const bool own = lane < g.xt && i < g.NX;
    const int ic = i + 1 < g.PX ? i : g.PX - 2;
    const int j0 = yt * RY;
    const int k0 = zc * g.zlen, k1 = min(k0 + g.zlen, g.NZ);
    d2 v[RY][3], nx[RY][3];
    auto load = [&](int k, d2 (*d)[3]) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long node = (long long)k * g.PP + (long long)min(j0 + r, g.NY - 1) * g.PX + ic;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[r][c] = *reinterpret_cast<const d2*>(in + c * g.stride + node);
        }
    };
    load(k0, v);
    for (int k = k0; k < k1; ++k) {
        load(min(k + 1, k1 - 1), nx);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long node = (long long)k * g.PP + (long long)min(j0 + r, g.NY - 1) * g.PX + ic;
            if (own && j0 + r < g.NY) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const d2 o = {v[r][c].x + 1.0, v[r][c].y + 1.0};
                    if (NTS) __builtin_nontemporal_store(o, reinterpret_cast<d2*>(out + c * g.stride + node));
                    else *reinterpret_cast<d2*>(out + c * g.stride + node) = o;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) v[r][c] = nx[r][c];
        }
    }
}

__device__ __forceinline__ double dpp_from_upper(double v) {      // lane l <- lane l+1 (row-local quad permute: pairs stay inside a quad)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// ST: 0 plain 8-byte stores, 1 non-temporal 8-byte stores, 2 even lanes store their node and their right neighbour's as one non-temporal 16 bytes
template <int RY, int ST>
__global__ __launch_bounds__(256) void k_march8(Geo g, const double* __restrict__ in, double* __restrict__ out) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bm = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) bm = (bm & 7) * (G >> 3) + (bm >> 3);
    const int nyg = (g.nty + 3) / 4;
    const int xt = bm % g.ntx, yt = ((bm / g.ntx) % nyg) * 4 + wv, zc = bm / (g.ntx * nyg);
    if (yt >= g.nty || zc >= g.nzc) return;
    const int i = xt * g.xt + lane;
    const bool own = lane < g.xt && i < g.NX;
    const int ic = i < g.PX ? i : g.PX - 1;
    const int j0 = yt * RY;
    const int k0 = zc * g.zlen, k1 = min(k0 + g.zlen, g.NZ);
    double v[RY][3], nx[RY][3];
    auto load = [&](int k, double (*d)[3]) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long node = (long long)k * g.PP + (long long)min(j0 + r, g.NY - 1) * g.PX + ic;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[r][c] = in[c * g.stride + node];
        }
    };
    load(k0, v);
    for (int k = k0; k < k1; ++k) {
        load(min(k + 1, k1 - 1), nx);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long node = (long long)k * g.PP + (long long)min(j0 + r, g.NY - 1) * g.PX + ic;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double o = v[r][c] + 1.0;
                if (ST == 2) {
                    const double o2 = dpp_from_upper(o);                  // every lane active
                    if (own && j0 + r < g.NY && !(lane & 1)) {
                        const d2 t = {o, o2};
                        __builtin_nontemporal_store(t, reinterpret_cast<d2*>(out + c * g.stride + node));      // (tile starts are even here)
                    }
                } else if (own && j0 + r < g.NY) {
                    if (ST == 1) __builtin_nontemporal_store(o, out + c * g.stride + node);
                    else out[c * g.stride + node] = o;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) v[r][c] = nx[r][c];
        }
    }
}

template <bool AOS, int RY>
__global__ __launch_bounds__(256) void k_march(Geo g, const double* __restrict__ in, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bm = blockIdx.x;
    const int G = gridDim.x;
    if ((G & 7) == 0) bm = (bm & 7) * (G >> 3) + (bm >> 3);          // XCD-contiguous tile ranges, as the real kernel
    const int nyg = (g.nty + 3) / 4;
    const int xt = bm % g.ntx, yt = ((bm / g.ntx) % nyg) * 4 + wv, zc = bm / (g.ntx * nyg);
    if (yt >= g.nty || zc >= g.nzc) return;
    const int i = xt * g.xt + lane;
    const bool own = lane < g.xt && i < g.NX;
    const int ic = i < g.PX ? i : g.PX - 1;
    const int j0 = yt * RY;
    const int k0 = zc * g.zlen, k1 = min(k0 + g.zlen, g.NZ);
    double v[RY][3], nx[RY][3];
    auto load = [&](int k, double (*d)[3]) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long node = (long long)k * g.PP + (long long)min(j0 + r, g.NY - 1) * g.PX + ic;
            if (AOS) {
                const double2 a = *reinterpret_cast<const double2*>(in + node * 3);
                d[r][0] = a.x; d[r][1] = a.y; d[r][2] = in[node * 3 + 2];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) d[r][c] = in[c * g.stride + node];
            }
        }
    };
    load(k0, v);
    for (int k = k0; k < k1; ++k) {
        load(min(k + 1, k1 - 1), nx);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const long long node = (long long)k * g.PP + (long long)min(j0 + r, g.NY - 1) * g.PX + ic;
            if (own && j0 + r < g.NY) {
                if (AOS) {
                    *reinterpret_cast<double2*>(out + node * 3) = make_double2(v[r][0] + 1.0, v[r][1] + 1.0);
                    out[node * 3 + 2] = v[r][2] + 1.0;
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) out[c * g.stride + node] = v[r][c] + 1.0;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) v[r][c] = nx[r][c];
        }
    }
}

template <bool AOS, int RY>
double run(Geo g, const double* in, double* out, int reps) {
    const int nyg = (g.nty + 3) / 4;
    int nb = g.ntx * nyg * g.nzc;
    nb = (nb + 7) / 8 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_march<AOS, RY>), dim3(nb), dim3(256), 0, 0, g, in, out);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_march<AOS, RY>), dim3(nb), dim3(256), 0, 0, g, in, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

template <int RY, int ST>
double run8(Geo g, const double* in, double* out, int reps) {
    const int nyg = (g.nty + 3) / 4;
    int nb = g.ntx * nyg * g.nzc;
    nb = (nb + 7) / 8 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_march8<RY, ST>), dim3(nb), dim3(256), 0, 0, g, in, out);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_march8<RY, ST>), dim3(nb), dim3(256), 0, 0, g, in, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

template <int RY, bool NTS>
double run16(Geo g, const double* in, double* out, int reps) {
    const int nyg = (g.nty + 3) / 4;
    int nb = g.ntx * nyg * g.nzc;
    nb = (nb + 7) / 8 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_march16<RY, NTS>), dim3(nb), dim3(256), 0, 0, g, in, out);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_march16<RY, NTS>), dim3(nb), dim3(256), 0, 0, g, in, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    Geo g;
    g.NX = argc > 3 ? atoi(argv[1]) : 513; g.NY = argc > 3 ? atoi(argv[2]) : 129; g.NZ = argc > 3 ? atoi(argv[3]) : 129;
    g.PX = (g.NX + 15) / 16 * 16;
    g.PP = (long long)g.PX * g.NY;
    g.stride = g.PP * (g.NZ + 2);
    double *in, *out;
    CK(hipMalloc(&in, g.stride * 3 * 8 + (1 << 20)));
    CK(hipMalloc(&out, g.stride * 3 * 8 + (1 << 20)));
    CK(hipMemset(in, 0, g.stride * 3 * 8));
    CK(hipMemset(out, 0, g.stride * 3 * 8));
    const double alg = 2.0 * 3 * 8 * (double)g.NX * g.NY * g.NZ;          // 48 B per node
    printf("grid %d x %d x %d nodes, %.0f MB algorithmic per pass\n", g.NX, g.NY, g.NZ, alg / 1e6);
    for (int zlen : {16, 32, 129}) {
        for (int xt : {62, 64, 48}) {
            g.xt = xt; g.zlen = zlen;
            g.ntx = (g.NX + xt - 1) / xt;
            g.nzc = (g.NZ + zlen - 1) / zlen;
            double t[4];
            g.nty = (g.NY + 1) / 2;
            t[0] = run<false, 2>(g, in, out, 20);
            t[1] = run<true, 2>(g, in, out, 20);
            g.nty = (g.NY + 3) / 4;
            t[2] = run<false, 4>(g, in, out, 20);
            t[3] = run<true, 4>(g, in, out, 20);
            printf("zlen %3d  tile %2d:  SoA RY=2 %6.1f us %5.2f TB/s | AoS RY=2 %6.1f us %5.2f TB/s | SoA RY=4 %6.1f us %5.2f TB/s | AoS RY=4 %6.1f us %5.2f TB/s\n",
                   zlen, xt, t[0] * 1e3, alg / t[0] / 1e9, t[1] * 1e3, alg / t[1] / 1e9, t[2] * 1e3, alg / t[2] / 1e9, t[3] * 1e3, alg / t[3] / 1e9);
        }
    }
    printf("one node per lane (8-byte loads), SoA, RY = 2, by store form:\n");
    for (int zlen : {16, 32}) {
        for (int xt : {62, 64}) {
            g.xt = xt; g.zlen = zlen;
            g.ntx = (g.NX + xt - 1) / xt;
            g.nzc = (g.NZ + zlen - 1) / zlen;
            g.nty = (g.NY + 1) / 2;
            const double t0 = run8<2, 0>(g, in, out, 20), t1 = run8<2, 1>(g, in, out, 20), t2 = run8<2, 2>(g, in, out, 20);
            printf("zlen %3d  tile %2d:  plain 8 B %6.1f us %5.2f TB/s | nt 8 B %6.1f us %5.2f TB/s | paired nt 16 B %6.1f us %5.2f TB/s\n",
                   zlen, xt, t0 * 1e3, alg / t0 / 1e9, t1 * 1e3, alg / t1 / 1e9, t2 * 1e3, alg / t2 / 1e9);
        }
    }
    printf("two nodes per lane (16-byte accesses), SoA:\n");
    for (int zlen : {16, 32, 64}) {
        for (int xt : {62, 64, 48}) {
            g.xt = xt; g.zlen = zlen;
            g.ntx = (g.NX + 2 * xt - 1) / (2 * xt);
            g.nzc = (g.NZ + zlen - 1) / zlen;
            double t[4];
            g.nty = g.NY;
            t[0] = run16<1, false>(g, in, out, 20);
            t[1] = run16<1, true>(g, in, out, 20);
            g.nty = (g.NY + 1) / 2;
            t[2] = run16<2, false>(g, in, out, 20);
            t[3] = run16<2, true>(g, in, out, 20);
            printf("zlen %3d  tile %3d:  RY=1 %6.1f us %5.2f TB/s | RY=1 nt-store %6.1f us %5.2f TB/s | RY=2 %6.1f us %5.2f TB/s | RY=2 nt-store %6.1f us %5.2f TB/s\n",
                   zlen, 2 * xt, t[0] * 1e3, alg / t[0] / 1e9, t[1] * 1e3, alg / t[1] / 1e9, t[2] * 1e3, alg / t[2] / 1e9, t[3] * 1e3, alg / t[3] / 1e9);
        }
    }
    return 0;
}
