// Calibration: what fp32 MFMA rate does this MI355X sustain (a) in a bare register loop, (b) with the operands
// re-read from LDS every step like the conv kernel does.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.01f, b = 1.0f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            a = sh[(it * 64 + lane) & 4095];
            b = sh[(it * 64 + lane + 2048) & 4095];
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + i, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("%-28s blocks/CU=%d  %.2f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 4; ++b) run<4, false>("regs, 4 acc", b);
    for (int b = 1; b <= 4; ++b) run<4, true>("lds operands, 4 acc", b);
    run<8, true>("lds operands, 8 acc", 2);
    run<1, false>("regs, 1 acc", 4);
    return 0;
}
