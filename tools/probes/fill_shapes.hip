// Diagnostic: what separates a plain fill's write rate from the rate at which an episode launch's rows leave?
// Every variant writes the same N bytes with 16-byte stores, whole 128-byte lines, to a buffer far larger
// than the caches; only who writes what, and when, differs.
//   0  256-thread workgroups, one 16-byte store per thread (4 KB per workgroup, N / 4 KB workgroups): a library fill
//   1  one-wave workgroups, one store per lane (1 KB per workgroup)
//   2  one-wave workgroups, each writes CH KB contiguous (one step's rows of 64 worlds) and ends
//   3  one-wave workgroups that live for T planes: CH KB into each of T planes, plane stride N / T (an episode launch)
//   4  as 3, but the wave's T chunks are adjacent (one contiguous region per wave)
//   5  as 0, persistent: 2048 workgroups of 256 threads stride over the buffer
//   6  as 3 with the waves' chunks of one plane dealt so that the waves of one XCD write one contiguous eighth
//   7  256-thread workgroups, 4 * CH KB each and out, the four waves interleaved (store i covers 4 KB)      8  each wave its own CH KB
//   9  as 7 but the workgroup lives for T planes                                                             10  as 8, T planes
//  11-15  as 3, each wave holding at most 0 / 1 / 2 / 4 / 8 stores in flight
//  16-19  as 0 with 2 / 4 / 8 / 16 stores per thread (interleaved over the workgroup) before the workgroup ends
//  20-27  persistent waves that write only the 4 KB blocks with index (XCD + C) mod 8, C = 0..7
//  28-30  as 0 with workgroups of 128 / 512 / 1024 threads        31  as 0, each thread a second store half the buffer away
//  32  as 3 with 8-byte stores      33 / 34  two-wave workgroups writing 64 KB per plane with 8- / 16-byte stores (a wave-per-world launch)
// hipcc --offload-arch=gfx950 -O3 -o fill_shapes fill_shapes.hip && ./fill_shapes [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CH = 11;      // KB a wave writes per step (simple_spread 3/3: 10.5 KB of observations + rewards)
constexpr int T = 25;

__global__ __launch_bounds__(256) void v0(float4* p) { p[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); }
__global__ __launch_bounds__(64) void v1(float4* p) { p[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); }
__global__ __launch_bounds__(64) void v2(float4* p) {
    float4* q = p + (size_t)blockIdx.x * (CH * 64) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < CH; ++i) q[i * 64] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <int MAP>
__global__ __launch_bounds__(64) void v3(float4* p, size_t plane_vecs) {
    unsigned w = blockIdx.x;
    if (MAP == 1) { const unsigned G = gridDim.x, xcd = w & 7u; w = xcd * (G >> 3) + (w >> 3); }
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int t = 0; t < T; ++t) {
        float4* q = p + (size_t)t * plane_vecs + (size_t)w * (CH * 64) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < CH; ++i) q[i * 64] = v;
        v.x += 1.f;
    }
}
__global__ __launch_bounds__(64) void v4(float4* p) {
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    float4* q = p + (size_t)blockIdx.x * (T * CH * 64) + threadIdx.x;
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < CH; ++i) q[(t * CH + i) * 64] = v;
        v.x += 1.f;
    }
}
__global__ __launch_bounds__(256) void v5(float4* p, size_t n_vecs) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vecs; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// 256-thread workgroups writing 4 * CH KB each.  COOP: store i of the workgroup's four waves covers one contiguous
// 4 KB (the waves interleave); otherwise each wave writes its own contiguous CH KB.  PERSIST: the workgroup lives
// for T planes (an episode launch with four-wave workgroups); otherwise one chunk and out.
template <bool COOP, bool PERSIST>
__global__ __launch_bounds__(256) void v7(float4* p, size_t plane_vecs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int t = 0; t < (PERSIST ? T : 1); ++t) {
        float4* q = p + (size_t)t * plane_vecs + (size_t)blockIdx.x * (4 * CH * 64);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (COOP) q[i * 256 + tid] = v;
            else q[(wave * CH + i) * 64 + lane] = v;
        }
        v.x += 1.f;
    }
}

// as 3 (an episode launch), each wave holding at most K stores in flight
template <int K>
__global__ __launch_bounds__(64) void v11(float4* p, size_t plane_vecs) {
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int t = 0; t < T; ++t) {
        float4* q = p + (size_t)t * plane_vecs + (size_t)blockIdx.x * (CH * 64) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            q[i * 64] = v;
            if (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        v.x += 1.f;
    }
}
// 256-thread workgroups, K stores per thread (K * 4 KB per workgroup, interleaved) and out
template <int K>
__global__ __launch_bounds__(256) void v12(float4* p) {
    float4* q = p + (size_t)blockIdx.x * (K * 256) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < K; ++i) q[i * 256] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// Persistent one-wave workgroups (T steps, 12 KB per step as three 4 KB blocks) where the wave on XCD x (workgroups
// are dealt round-robin over the 8 XCDs) only ever writes 4 KB blocks whose index is (x + C) mod 8.
template <int C>
__global__ __launch_bounds__(64) void v20(float4* p) {
    const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3;
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t block = ((size_t)(i * T + t) * 3 + k) * 8 + ((x + C) & 7u);
            float4* q = p + block * 256 + threadIdx.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j * 64] = v;
        }
        v.x += 1.f;
    }
}

// one store per thread and out, workgroups of NTH threads
template <int NTH>
__global__ __launch_bounds__(NTH) void v28(float4* p) { p[(size_t)blockIdx.x * NTH + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); }
// two stores per thread and out, the second a whole plane (half the buffer) away
__global__ __launch_bounds__(256) void v31(float4* p, size_t half) {
    float4* q = p + (size_t)blockIdx.x * 256 + threadIdx.x;
    q[0] = make_float4(1.f, 2.f, 3.f, 4.f);
    q[half] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// as 3 with 8-byte stores (twice the instructions, 512 bytes = four lines per wave and instruction)
__global__ __launch_bounds__(64) void v32(float2* p, size_t plane_vecs) {
    float2 v = make_float2(1.f, 2.f);
    for (int t = 0; t < T; ++t) {
        float2* q = p + (size_t)t * plane_vecs + (size_t)blockIdx.x * (CH * 128) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 2 * CH; ++i) q[i * 64] = v;
        v.x += 1.f;
    }
}
// a wave-per-world launch's shape: two-wave workgroups, each living T planes and writing 66 KB per plane, the two
// waves interleaved; VB = bytes per lane and store
template <typename V>
__global__ __launch_bounds__(128) void v33(V* p, size_t plane_vecs, int vecs_per_wg) {
    V v{};
    for (int t = 0; t < T; ++t) {
        V* q = p + (size_t)t * plane_vecs + (size_t)blockIdx.x * vecs_per_wg + threadIdx.x;
#pragma unroll 4
        for (int i = 0; i < vecs_per_wg; i += 128) q[i] = v;
    }
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    // waves of the persistent variants: a whole number of (T * CH KB) regions
    const size_t waves = (size_t)(gib * 1073741824.0 / (T * CH * 1024.0));
    const size_t n_bytes = waves * T * CH * 1024, n_vecs = n_bytes / 16;
    const unsigned w20 = (unsigned)((n_bytes / (T * 3 * 4096)) & ~(size_t)7);       // waves of variants 20-27
    const unsigned wg33 = (unsigned)(n_bytes / ((size_t)T * 65536));        // workgroups of variants 33-34 (64 KB per plane each)
    float4* p;
    CK(hipMalloc(&p, n_bytes));
    CK(hipMemset(p, 0, n_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%.2f GiB, %zu waves in the persistent variants\n", n_bytes / 1073741824.0, waves);
    for (int variant = 0; variant < 35; ++variant) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            switch (variant) {
                case 0: hipLaunchKernelGGL(v0, dim3((unsigned)(n_vecs / 256)), dim3(256), 0, 0, p); break;
                case 1: hipLaunchKernelGGL(v1, dim3((unsigned)(n_vecs / 64)), dim3(64), 0, 0, p); break;
                case 2: hipLaunchKernelGGL(v2, dim3((unsigned)(waves * T)), dim3(64), 0, 0, p); break;
                case 3: hipLaunchKernelGGL(v3<0>, dim3((unsigned)waves), dim3(64), 0, 0, p, n_vecs / T); break;
                case 4: hipLaunchKernelGGL(v4, dim3((unsigned)waves), dim3(64), 0, 0, p); break;
                case 5: hipLaunchKernelGGL(v5, dim3(2048), dim3(256), 0, 0, p, n_vecs); break;
                case 7: hipLaunchKernelGGL((v7<true, false>), dim3((unsigned)(waves * T / 4)), dim3(256), 0, 0, p, (size_t)0); break;
                case 8: hipLaunchKernelGGL((v7<false, false>), dim3((unsigned)(waves * T / 4)), dim3(256), 0, 0, p, (size_t)0); break;
                case 9: hipLaunchKernelGGL((v7<true, true>), dim3((unsigned)(waves / 4)), dim3(256), 0, 0, p, n_vecs / T); break;
                case 10: hipLaunchKernelGGL((v7<false, true>), dim3((unsigned)(waves / 4)), dim3(256), 0, 0, p, n_vecs / T); break;
                case 11: hipLaunchKernelGGL(v11<0>, dim3((unsigned)waves), dim3(64), 0, 0, p, n_vecs / T); break;
                case 12: hipLaunchKernelGGL(v11<1>, dim3((unsigned)waves), dim3(64), 0, 0, p, n_vecs / T); break;
                case 13: hipLaunchKernelGGL(v11<2>, dim3((unsigned)waves), dim3(64), 0, 0, p, n_vecs / T); break;
                case 14: hipLaunchKernelGGL(v11<4>, dim3((unsigned)waves), dim3(64), 0, 0, p, n_vecs / T); break;
                case 15: hipLaunchKernelGGL(v11<8>, dim3((unsigned)waves), dim3(64), 0, 0, p, n_vecs / T); break;
                case 16: hipLaunchKernelGGL(v12<2>, dim3((unsigned)(n_vecs / 512)), dim3(256), 0, 0, p); break;
                case 17: hipLaunchKernelGGL(v12<4>, dim3((unsigned)(n_vecs / 1024)), dim3(256), 0, 0, p); break;
                case 18: hipLaunchKernelGGL(v12<8>, dim3((unsigned)(n_vecs / 2048)), dim3(256), 0, 0, p); break;
                case 19: hipLaunchKernelGGL(v12<16>, dim3((unsigned)(n_vecs / 4096)), dim3(256), 0, 0, p); break;
                case 20: hipLaunchKernelGGL(v20<0>, dim3(w20), dim3(64), 0, 0, p); break;
                case 21: hipLaunchKernelGGL(v20<1>, dim3(w20), dim3(64), 0, 0, p); break;
                case 22: hipLaunchKernelGGL(v20<2>, dim3(w20), dim3(64), 0, 0, p); break;
                case 23: hipLaunchKernelGGL(v20<3>, dim3(w20), dim3(64), 0, 0, p); break;
                case 24: hipLaunchKernelGGL(v20<4>, dim3(w20), dim3(64), 0, 0, p); break;
                case 25: hipLaunchKernelGGL(v20<5>, dim3(w20), dim3(64), 0, 0, p); break;
                case 26: hipLaunchKernelGGL(v20<6>, dim3(w20), dim3(64), 0, 0, p); break;
                case 27: hipLaunchKernelGGL(v20<7>, dim3(w20), dim3(64), 0, 0, p); break;
                case 28: hipLaunchKernelGGL(v28<128>, dim3((unsigned)(n_vecs / 128)), dim3(128), 0, 0, p); break;
                case 29: hipLaunchKernelGGL(v28<512>, dim3((unsigned)(n_vecs / 512)), dim3(512), 0, 0, p); break;
                case 30: hipLaunchKernelGGL(v28<1024>, dim3((unsigned)(n_vecs / 1024)), dim3(1024), 0, 0, p); break;
                case 31: hipLaunchKernelGGL(v31, dim3((unsigned)(n_vecs / 512)), dim3(256), 0, 0, p, n_vecs / 2); break;
                case 32: hipLaunchKernelGGL(v32, dim3((unsigned)waves), dim3(64), 0, 0, (float2*)p, n_vecs * 2 / T); break;
                case 33: hipLaunchKernelGGL(v33<float2>, dim3(wg33), dim3(128), 0, 0, (float2*)p, (size_t)wg33 * 8192, 8192); break;
                case 34: hipLaunchKernelGGL(v33<float4>, dim3(wg33), dim3(128), 0, 0, p, (size_t)wg33 * 4096, 4096); break;
                case 6: hipLaunchKernelGGL(v3<1>, dim3((unsigned)(waves & ~(size_t)7)), dim3(64), 0, 0, p, n_vecs / T); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = variant >= 33 ? (double)wg33 * T * 65536 : (variant >= 20 && variant < 28 ? (double)w20 * T * 3 * 4096 : (double)n_bytes);
        printf("variant %d  %.3f ms  %.0f GB/s\n", variant, best, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
