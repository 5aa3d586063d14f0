#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int src = lane * 10;
    int old = -1;
    out[lane] = __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(old, src, 0x101, 0xf, 0xf, false);
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("row_shr:1 (0x111): "); for (int i = 0; i < 20; ++i) printf("%d ", h[i]); printf("\n");
    printf("row_shl:1 (0x101): "); for (int i = 0; i < 20; ++i) printf("%d ", h[64 + i]); printf("\n");
}
