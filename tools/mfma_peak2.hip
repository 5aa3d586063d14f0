// Calibration 2: does keeping the fp32-MFMA accumulators in AGPRs ("+a") instead of VGPRs help an LDS-fed loop?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: builtin (VGPR acc), 1: inline asm AGPR acc
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float sh[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sh[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const float4* a4 = reinterpret_cast<const float4*>(sh);
    float4 a = a4[lane];
    float b = sh[4096 + lane];
    for (int it = 0; it < iters; ++it) {
        const float4 an = a4[((it + 1) * 64 + lane) & 1023];
        const float bn = sh[4096 + (((it + 1) * 64 + lane) & 4095)];
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc[3], 0, 0, 0);
        } else {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %8, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %1, %5, %8, %1\n\t"
                         "v_mfma_f32_32x32x2_f32 %2, %6, %8, %2\n\t"
                         "v_mfma_f32_32x32x2_f32 %3, %7, %8, %3"
                         : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3])
                         : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b));
        }
        a = an;
        b = bn;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int bpc) {
    float* out;
    const int blocks = 256 * bpc;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 4 * 4096.0;
    printf("%-32s blocks/CU=%d  %.2f ms  %.1f TFLOP/s\n", name, bpc, ms, fl / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int b = 1; b <= 4; ++b) run<0>("LDS-fed prefetch, VGPR acc", b);
    for (int b = 1; b <= 4; ++b) run<1>("LDS-fed prefetch, AGPR acc", b);
    return 0;
}
