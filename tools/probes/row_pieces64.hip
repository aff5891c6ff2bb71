// Diagnostic: does it pay to cut per-agent row pieces at 64-byte boundaries?  Shape of collector: [T][B][684] float32,
// one 64-lane wave per 64 worlds, 8 pieces per row (6 x 86 + 2 x 84 values).
//   V1: every piece as it is (starts and ends anywhere: partial 64-byte sectors at both ends, completed by a later pass)
//   V3: piece c of a row covers [floor16(start_c), floor16(start_{c+1})): what a pass has not rounded off waits for the next
//       (the row's first piece starts at the row start, its last ends at the row end)
//   V4: piece c of a row covers [floor4(start_c), floor4(start_{c+1})) and leaves as 16-byte vectors, 22 vector slots per row
//   V2: the wave's whole block as contiguous 16-byte vectors (no pieces)
//   hipcc --offload-arch=gfx950 -O3 -o row_pieces64 row_pieces64.hip && ./row_pieces64 65536
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int OT = 684, NP = 8;
__host__ __device__ constexpr int off(int c) { return c <= 6 ? 86 * c : 516 + 84 * (c - 6); }     // off(8) == 684
template <int V>
__global__ __launch_bounds__(64) void k(float* obs, long long B, int T, int spin) {
    const int lane = threadIdx.x;
    const long long env0 = (long long)blockIdx.x * 64;
    float v = (float)lane;
    for (int t = 0; t < T; ++t) {
        float* o = obs + ((long long)t * B + env0) * OT;
        if (V == 2) {
            f4v* o4 = (f4v*)o;
            const f4v v4 = {v, v, v, v};
            for (int c = 0; c < NP; ++c) {
                for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f;      // the arithmetic between two passes
                const int n4 = 64 * OT / 4 / NP;                             // an eighth of the block per pass
                for (int g = lane; g < n4; g += 64) o4[c * n4 + g] = v4;
            }
        } else if (V == 4) {
            f4v* o4 = (f4v*)o;
            const f4v v4 = {v, v, v, v};
            for (int c = 0; c < NP; ++c) {
                for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f;
                for (int g = lane; g < 64 * 22; g += 64) {
                    const int r = g / 22, j = g - r * 22;
                    const int a = (r * OT + off(c)) & ~3, b = (r * OT + off(c + 1)) & ~3;
                    if (a + 4 * j < b) o4[(a >> 2) + j] = v4;
                }
            }
        } else {
            for (int c = 0; c < NP; ++c) {
                for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f;
                for (int r = 0; r < 64; ++r) {
                    int a = r * OT + off(c), b = r * OT + off(c + 1);
                    if (V == 3) {
                        if (c > 0) a &= ~15;
                        if (c < NP - 1) b &= ~15;
                    }
                    for (int g = a + lane; g < b; g += 64) o[g] = v;
                }
            }
        }
        v += 1.f;
    }
}
template <int V> void go(const char* name, float* obs, long long B, int T, int spin) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 4.0 * OT * T * B;
    for (int rep = 0; rep < 3; ++rep) {
        const int N = 5;
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<V>, dim3(B / 64), dim3(64), 0, 0, obs, B, T, spin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("B=%lld spin=%d %-44s %9.1f us/launch  %.0f GB/s\n", B, spin, name, ms * 1e3 / N, bytes / (ms * 1e-3 / N) / 1e9);
    }
}
int main(int argc, char** argv) {
    const long long B = argc > 1 ? atoll(argv[1]) : 65536;
    if (B <= 0 || B % 64 != 0) { printf("B must be a positive multiple of 64\n"); return 1; }
    const int T = 25;
    float* obs;
    CK(hipMalloc(&obs, (size_t)T * B * OT * 4));
    for (int spin : {0, 400, 1500}) {
        go<1>("pieces as they are", obs, B, T, spin);
        go<3>("pieces cut at 64-byte boundaries", obs, B, T, spin);
        go<4>("pieces cut at 16 bytes, 16-byte vectors", obs, B, T, spin);
        go<2>("contiguous 16-byte vectors", obs, B, T, spin);
    }
    return 0;
}
