// Diagnostic: one wave per SIMD that alternates arithmetic and a burst of stores, against the same wave with the
// stores of the previous step spread between slices of the arithmetic.
// Per step and wave: `work` rounds of dependent arithmetic and 10.5 KiB of rows (eleven 16-byte-per-lane stores, the
// last one half a wave) to a [T][B] array, as an episode launch of simple_spread at 65 536 worlds writes them.
//   hipcc --offload-arch=gfx950 -O3 -o interleave interleave.hip && ./interleave <work>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SLICES>      // 1: arithmetic, then all stores; 11: one store after every eleventh of the arithmetic; 3: four stores after every third
__global__ __launch_bounds__(64) void k(float4* obs, long long B, int T, int work) {
    const int lane = threadIdx.x;
    const long long env0 = (long long)blockIdx.x * 64;
    float4 v = make_float4(lane, 1.f, 2.f, 3.f), prev = v;
    const int per = work / SLICES;
    for (int t = 0; t <= T; ++t) {
        float4* o = obs + ((long long)(t - 1) * B + env0) * 42 / 4;     // rows of the previous step
        if (SLICES == 1) {
            for (int i = 0; i < work; ++i) v.y = v.y * 1.0001f + v.z;
            if (t > 0) {
#pragma unroll
                for (int it = 0; it < 10; ++it) o[it * 64 + lane] = prev;
                if (lane < 32) o[640 + lane] = prev;
            }
        } else if (SLICES == 3) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (t < T) for (int i = 0; i < per; ++i) v.y = v.y * 1.0001f + v.z;
                if (t > 0) {
#pragma unroll
                    for (int it = 4 * s; it < 4 * s + 4 && it < 11; ++it)
                        if (it < 10 || lane < 32) o[it * 64 + lane] = prev;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 11; ++s) {
                if (t < T) for (int i = 0; i < per; ++i) v.y = v.y * 1.0001f + v.z;
                if (t > 0 && (s < 10 || lane < 32)) o[s * 64 + lane] = prev;
            }
        }
        prev = v;
        v.x += 1.f;
    }
}

int main(int argc, char** argv) {
    const long long B = argc > 2 ? atoll(argv[2]) : 65536;
    const int T = 24, work = argc > 1 ? atoi(argv[1]) : 99;
    float4* obs;
    CK(hipMalloc(&obs, (size_t)T * B * 168));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            const int N = 20;
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k<1>, dim3(B / 64), dim3(64), 0, 0, obs, B, T, work);
                else if (variant == 1) hipLaunchKernelGGL(k<11>, dim3(B / 64), dim3(64), 0, 0, obs, B, T, work);
                else hipLaunchKernelGGL(k<3>, dim3(B / 64), dim3(64), 0, 0, obs, B, T, work);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / N < best) best = ms / N;
        }
        printf("B=%lld work=%d %s  %.1f us/launch  %.0f GB/s\n", B, work, variant == 0 ? "arithmetic, then the stores" : (variant == 1 ? "one store after every eleventh of the arithmetic" : "four stores after every third of the arithmetic"),
               best * 1e3, (double)T * B * 168 / (best * 1e-3) / 1e9);
    }
    return 0;
}
