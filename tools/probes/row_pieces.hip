// Diagnostic: cost of writing an env-major observation block in per-agent column pieces instead of whole lines.
// Shape of simple_predator_prey: [T][B][315] float32, one 64-lane wave per 64 worlds, 7 pieces of 45 values per row.
//   hipcc --offload-arch=gfx950 -O3 -o row_pieces row_pieces.hip && ./row_pieces 131072
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int OT = 315, W = 45;
// V0: one row piece (45 lanes) per store instruction.  V1: 64 consecutive values of a piece-major tile per
// instruction.  V2: the wave's whole 64 x 315 block as contiguous 16-byte vectors.
template <int V>
__global__ __launch_bounds__(64) void k(float* obs, long long B, int T) {
    const int lane = threadIdx.x;
    const long long env0 = (long long)blockIdx.x * 64;
    float v = (float)lane;
    for (int t = 0; t < T; ++t) {
        float* o = obs + ((long long)t * B + env0) * OT;
        if (V == 0) {
#pragma unroll
            for (int c = 0; c < 7; ++c)
                if (lane < W) {
#pragma unroll
                    for (int r = 0; r < 64; ++r) o[r * OT + c * W + lane] = v;
                }
        } else if (V == 1) {
#pragma unroll
            for (int c = 0; c < 7; ++c)
#pragma unroll
                for (int it = 0; it < W; ++it) {
                    const int g = it * 64 + lane, r = g / W;
                    o[r * OT + c * W + (g - r * W)] = v;
                }
        } else {
            f4v* o4 = (f4v*)o;
            const f4v v4 = {v, v, v, v};
#pragma unroll
            for (int it = 0; it < 64 * OT / 4 / 64; ++it) o4[it * 64 + lane] = v4;
            if (lane < (64 * OT / 4) % 64) o4[(64 * OT / 4 / 64) * 64 + lane] = v4;
        }
        v += 1.f;
    }
}
template <int V> void go(const char* name, float* obs, long long B, int T) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 4.0 * OT * T * B;
    for (int rep = 0; rep < 3; ++rep) {
        const int N = 5;
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<V>, dim3(B / 64), dim3(64), 0, 0, obs, B, T);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("B=%lld %-44s %9.1f us/launch  %.0f GB/s\n", B, name, ms * 1e3 / N, bytes / (ms * 1e-3 / N) / 1e9);
    }
}
int main(int argc, char** argv) {
    const long long B = argc > 1 ? atoll(argv[1]) : 131072;
    if (B <= 0 || B % 64 != 0) { printf("B must be a positive multiple of 64\n"); return 1; }
    const int T = 25;
    float* obs;
    CK(hipMalloc(&obs, (size_t)T * B * OT * 4));
    go<0>("per-agent pieces, row piece per store", obs, B, T);
    go<1>("per-agent pieces, 64 values per store", obs, B, T);
    go<2>("contiguous 16-byte vectors", obs, B, T);
    return 0;
}
