// Diagnostic: ceiling of the rollout kernel's store pattern without any compute.
// Each 64-lane wave owns 64 envs; per step it writes 10752 B (obs), 768 B (rew) and 64 B (done)
// to three [T][B]-major arrays, T steps per launch.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(64) void k(float4* obs, float4* rew, unsigned char* done, long long B, int T) {
    const int lane = threadIdx.x;
    const long long env0 = (long long)blockIdx.x * 64;
    float4 v = make_float4(lane, 1.f, 2.f, 3.f);
    for (int t = 0; t < T; ++t) {
        const long long row0 = (long long)t * B + env0;
        float4* o = obs + row0 * 42 / 4;
#pragma unroll
        for (int it = 0; it < 10; ++it) o[it * 64 + lane] = v;
        if (lane < 32) o[640 + lane] = v;
        if (VARIANT >= 1) { if (lane < 48) (rew + row0 * 3 / 4)[lane] = v; }
        if (VARIANT >= 2) done[row0 + lane] = 0;
        v.x += 1.f;
    }
}

int main(int argc, char** argv) {
    long long B = argc > 1 ? atoll(argv[1]) : 65536;
    int T = 25;
    float4 *obs, *rew; unsigned char* done;
    CK(hipMalloc(&obs, (size_t)T * B * 168)); CK(hipMalloc(&rew, (size_t)T * B * 12)); CK(hipMalloc(&done, (size_t)T * B));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            const int N = 20;
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(B / 64), dim3(64), 0, 0, obs, rew, done, B, T);
                if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(B / 64), dim3(64), 0, 0, obs, rew, done, B, T);
                if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(B / 64), dim3(64), 0, 0, obs, rew, done, B, T);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = (double)T * B * (168 + (variant >= 1 ? 12 : 0) + (variant >= 2 ? 1 : 0));
            if (rep == 2) printf("B=%lld variant=%d  %.1f us/launch  %.0f GB/s\n", B, variant, ms * 1e3 / N, bytes / (ms * 1e-3 / N) / 1e9);
        }
    }
    return 0;
}
