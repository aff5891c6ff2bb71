// Diagnostic: can the observation block of a 64-agent / 64-landmark simple_spread world (66 048 B per world-step) be
// written by a second kernel of one-store workgroups, from a per-step record of positions and velocities, at the
// rate of a plain fill?  Built as a shared library (the driver is obs_expand.py):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libobsx.so obs_expand.hip
// A world-step's block is 4 128 16-byte quads; two world-steps are 129 waves, i.e. 43 workgroups of 192 threads, each
// thread storing one quad (two coordinate pairs, each `first - subtrahend` of the world's 193-entry table of pairs:
// 128 entity positions, 64 velocities, one zero).  `tab` holds the four table indices of every quad of a world-step.
//   variant 0  the world's record staged in LDS, gathered from there     1  gathered from global memory directly
//   variant 2  no loads at all (a fill in the same workgroup shape)       3  as 0 with 256-thread workgroups, flat index
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int QW = 4128, REC = 192;

template <int VARIANT>
__global__ __launch_bounds__(192) void obsx(const float2* __restrict__ rec, const uint32_t* __restrict__ tab, float4* __restrict__ obs) {
    __shared__ float2 S[2][REC + 1];
    const int tid = threadIdx.x;
    const long long pair = blockIdx.y;
    const int q0 = blockIdx.x * 192, q = q0 + tid;
    const int w_lo = q0 >= QW, w_hi = (q0 + 191) >= QW;
    const int w = q >= QW, qw = q - w * QW;
    float4 out;
    if (VARIANT == 2) {
        out = make_float4((float)q, 1.f, 2.f, 3.f);
    } else {
        const uint32_t e = tab[qw];
        if (VARIANT == 0) {
            S[0][tid] = rec[(pair * 2 + w_lo) * REC + tid];
            if (w_hi != w_lo) S[1][tid] = rec[(pair * 2 + w_hi) * REC + tid];
            if (tid == 0) { S[0][REC] = make_float2(0.f, 0.f); S[1][REC] = make_float2(0.f, 0.f); }
            __syncthreads();
            const float2* s = S[w - w_lo];
            const float2 a0 = s[e & 255u], b0 = s[(e >> 8) & 255u], a1 = s[(e >> 16) & 255u], b1 = s[e >> 24];
            out = make_float4(a0.x - b0.x, a0.y - b0.y, a1.x - b1.x, a1.y - b1.y);
        } else {
            // the record carries its zero pair at index 192 of a 193-entry stride?  no: index REC reads the next
            // world's first pair, so subtract a masked value instead
            const float2* s = rec + (pair * 2 + w) * REC;
            const unsigned i0 = e & 255u, j0 = (e >> 8) & 255u, i1 = (e >> 16) & 255u, j1 = e >> 24;
            const float2 a0 = s[i0], b0 = s[j0 == REC ? 0 : j0], a1 = s[i1], b1 = s[j1 == REC ? 0 : j1];
            const float m0 = j0 == REC ? 0.f : 1.f, m1 = j1 == REC ? 0.f : 1.f;
            out = make_float4(a0.x - m0 * b0.x, a0.y - m0 * b0.y, a1.x - m1 * b1.x, a1.y - m1 * b1.y);
        }
    }
    obs[pair * (2 * QW) + q] = out;
}

__global__ __launch_bounds__(256) void obsx_flat(const float2* __restrict__ rec, const uint32_t* __restrict__ tab, float4* __restrict__ obs,
                                                 long long n_quads) {
    __shared__ float2 S[2][REC + 1];
    const int tid = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * 256, g = g0 + tid;
    const long long ws_lo = g0 / QW, ws_hi = (g0 + 255) / QW;
    if (tid < REC) {
        S[0][tid] = rec[ws_lo * REC + tid];
        if (ws_hi != ws_lo) S[1][tid] = rec[ws_hi * REC + tid];
    }
    if (tid == 0) { S[0][REC] = make_float2(0.f, 0.f); S[1][REC] = make_float2(0.f, 0.f); }
    __syncthreads();
    if (g >= n_quads) return;
    const long long ws = g / QW;
    const int qw = (int)(g - ws * QW);
    const uint32_t e = tab[qw];
    const float2* s = S[ws - ws_lo];
    const float2 a0 = s[e & 255u], b0 = s[(e >> 8) & 255u], a1 = s[(e >> 16) & 255u], b1 = s[e >> 24];
    obs[g] = make_float4(a0.x - b0.x, a0.y - b0.y, a1.x - b1.x, a1.y - b1.y);
}

extern "C" int obsx_launch(int variant, const void* rec, const void* tab, void* obs, long long n_ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(43, (unsigned)(n_ws / 2)), block(192);
    const float2* r = (const float2*)rec;
    const uint32_t* t = (const uint32_t*)tab;
    float4* o = (float4*)obs;
    if (n_ws % 2) return -1;
    if (variant == 0) hipLaunchKernelGGL(obsx<0>, grid, block, 0, st, r, t, o);
    else if (variant == 1) hipLaunchKernelGGL(obsx<1>, grid, block, 0, st, r, t, o);
    else if (variant == 2) hipLaunchKernelGGL(obsx<2>, grid, block, 0, st, r, t, o);
    else {
        const long long n_quads = n_ws * QW;
        hipLaunchKernelGGL(obsx_flat, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, st, r, t, o, n_quads);
    }
    return (int)hipGetLastError();
}
