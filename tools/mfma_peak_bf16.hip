// Calibration: what v_mfma_f32_32x32x16_bf16 rate does this MI355X sustain in a bare register loop (8 accumulators per
// wave, like the conv kernel's wave tile) at 1 and 2 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(0.001f * (threadIdx.x + j));
        b[j] = (__bf16)(0.5f + 0.01f * j);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * NACC * 32768.0;
    printf("%-28s blocks/CU=%d  %.2f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 2; ++b) run<8>("bf16 32x32x16, 8 acc", b);
    for (int b = 1; b <= 2; ++b) run<4>("bf16 32x32x16, 4 acc", b);
    run<8>("bf16 32x32x16, 8 acc (again, warm)", 2);
    return 0;
}
