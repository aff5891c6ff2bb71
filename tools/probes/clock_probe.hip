// Diagnostic: shader clock and per-instruction latency seen by ONE wave per SIMD (the occupancy of
// the B = 65536 rollout launch).  Reports cycles per dependent v_fma, per dependent transcendental,
// and the clock frequency from s_memtime (shader cycles) vs s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k(float* out, long long* stamps, int iters, int mode) {
    float x = threadIdx.x * 1e-3f + 1.0f, y = 0.5f;
    long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, 0.999f, y);
        } else if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 64; ++j) x = __builtin_amdgcn_sqrtf(x + 1.0f);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {  // 4 independent chains
                x = __builtin_fmaf(x, 0.999f, y); y = __builtin_fmaf(y, 0.999f, 0.1f);
                x = __builtin_fmaf(x, 0.999f, 0.2f); y = __builtin_fmaf(y, 0.999f, 0.3f);
            }
        }
    }
    long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = x + y;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 1024;
    float* out; long long* st;
    CK(hipMalloc(&out, blocks * 64 * 4)); CK(hipMalloc(&st, blocks * 16));
    long long* h = (long long*)malloc(blocks * 16);
    for (int mode = 0; mode < 3; ++mode) for (int iters : {100, 2000}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, st, iters, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
        cyc /= blocks; rt /= blocks;
        printf("blocks=%d mode=%d iters=%d: %.2f shader cycles/instr, %.0f MHz shader clock, %.1f us per wave\n", blocks, mode, iters,
               cyc / (iters * 64.0), cyc / (rt / 100.0), rt / 100.0);
    }
    return 0;
}
