// Probe of ds_read_b64_tr_b16 on gfx950: which LDS 16-bit element lands in (lane, j)?
// LDS holds lds[i] = i. mode 0: lane l supplies address 4*l elements (8 contiguous bytes per lane).
// mode 1: lane l supplies address (l&15)*64 + (l>>4)*4 elements.   Prints the element index per (lane, j).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(short* out, int mode) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int lane = threadIdx.x;
    int addr_elems = mode == 0 ? lane * 4 : (lane & 15) * 64 + (lane >> 4) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + addr_elems));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    return 0;
}
