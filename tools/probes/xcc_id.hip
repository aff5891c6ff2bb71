// Diagnostic: which XCD does workgroup i of a launch run on?  Every workgroup stores the hardware's XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o variants/libxccid.so tools/probes/xcc_id.hip      (driver: xcc_id.py)
#include <hip/hip_runtime.h>
__global__ void xcc_kernel(int* out) {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits 3:0: simm16 = (size - 1) << 11 | offset << 6 | id
    const unsigned id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(id & 15u);
}
extern "C" int xcc_probe(int* out, int n_wg, int threads, void* stream) {
    hipLaunchKernelGGL(xcc_kernel, dim3(n_wg), dim3(threads), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
