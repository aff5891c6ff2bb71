// Diagnostic: what the reward rows of an episode launch cost by the shape of their store.
// usage: rew_shapes <worlds> [dependent instructions between steps, default 0]
// Each 64-lane wave owns 64 envs; per step it writes 10752 B of observations (16-byte vectors, whole lines) and its
// 768 B of rewards ([B][3] floats) in one of several shapes.   hipcc --offload-arch=gfx950 -O3
//   0 no rewards                         1 48 lanes x 16 B (one masked instruction, as the kernels do)
//   2 64 lanes x 12 B (dwordx3)          3 three instructions of 64 lanes x 4 B
//   4 four steps' rewards together as three full 16-byte instructions (rows of four planes)
//   5 as 1, issued before the observation rows instead of after them
//   6 ONE output stream: rows of 46 floats (42 observation values, 3 rewards, the done flag as a float), 11 776 B per
//     wave-step, contiguous 16-byte vectors -- the same bytes + 3 per env-step, but no second and third stream.
//     (Here, with arithmetic between the steps, it costs nothing: 61.8 us without rewards, 69.5 us as the kernels write
//     them, 62.3 us as packed rows.  The real episode kernel did not follow: built with packed rows it took 64.1-64.8 us
//     against 60.4-61.5 us for the three arrays on the same box, so the kernels keep the three arrays.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct F3 { float x, y, z; };

template <int VARIANT>
__global__ __launch_bounds__(64) void k(float4* obs, float* rew, long long B, int T, int work) {
    const int lane = threadIdx.x;
    const long long env0 = (long long)blockIdx.x * 64;
    float4 v = make_float4(lane, 1.f, 2.f, 3.f);
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < work; ++i) v.y = v.y * 1.0001f + v.z;      // a step's arithmetic: `work` dependent instructions
        const long long row0 = (long long)t * B + env0;
        float4* o = obs + row0 * 42 / 4;
        float* r = rew + row0 * 3;
        if (VARIANT == 6) {
            float4* p = obs + row0 * 46 / 4;
#pragma unroll
            for (int it = 0; it < 11; ++it) p[it * 64 + lane] = v;
            if (lane < 32) p[704 + lane] = v;
            v.x += 1.f;
            continue;
        }
        if (VARIANT == 5) { if (lane < 48) reinterpret_cast<float4*>(r)[lane] = v; }
#pragma unroll
        for (int it = 0; it < 10; ++it) o[it * 64 + lane] = v;
        if (lane < 32) o[640 + lane] = v;
        if (VARIANT == 1) { if (lane < 48) reinterpret_cast<float4*>(r)[lane] = v; }
        if (VARIANT == 2) { F3 w{v.x, v.y, v.z}; reinterpret_cast<F3*>(r)[lane] = w; }
        if (VARIANT == 3) { r[lane] = v.x; r[64 + lane] = v.y; r[128 + lane] = v.z; }
        if (VARIANT == 4 && (t & 3) == 3) {
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int g = it * 64 + lane, plane = g / 48, within = g - plane * 48;       // 192 vectors: 4 planes x 48
                reinterpret_cast<float4*>(rew + ((long long)(t - 3 + plane) * B + env0) * 3)[within] = v;
            }
        }
        v.x += 1.f;
    }
}

int main(int argc, char** argv) {
    long long B = argc > 1 ? atoll(argv[1]) : 65536;
    int T = 24;
    int work = argc > 2 ? atoi(argv[2]) : 0;      // dependent instructions between two steps' stores
    float4* obs; float* rew;
    CK(hipMalloc(&obs, (size_t)T * B * 184)); CK(hipMalloc(&rew, (size_t)T * B * 12));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 7; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            const int N = 20;
            for (int i = 0; i < N; ++i) {
                const dim3 g(B / 64), b(64);
                if (variant == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, obs, rew, B, T, work);
                if (variant == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, obs, rew, B, T, work);
                if (variant == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, obs, rew, B, T, work);
                if (variant == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, obs, rew, B, T, work);
                if (variant == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, obs, rew, B, T, work);
                if (variant == 5) hipLaunchKernelGGL(k<5>, g, b, 0, 0, obs, rew, B, T, work);
                if (variant == 6) hipLaunchKernelGGL(k<6>, g, b, 0, 0, obs, rew, B, T, work);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = (double)T * B * (variant == 6 ? 184 : 168 + (variant >= 1 ? 12 : 0));
            if (rep == 2) printf("B=%lld work=%d variant=%d  %.1f us/launch  %.0f GB/s\n", B, work, variant, ms * 1e3 / N, bytes / (ms * 1e-3 / N) / 1e9);
        }
    }
    return 0;
}
