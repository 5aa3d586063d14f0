// Does the fp32 MFMA (v_mfma_f32_16x16x4_f32) share its datapath with fp32 VALU work?  One or two waves per SIMD run a loop
// of independent MFMAs with NV independent v_fma_f32 per MFMA interleaved; if the two overlapped, the time would stay at
// the MFMA-only time until the VALU issue slots run out.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool MF, int KIND = 0>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.01f, b = 1.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk[5];
    for (int i = 0; i < 5; ++i) pk[i] = f2{v[i], v[i + 1]};
    __shared__ f32x4 sh[1024];
    sh[threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f32x4 ld[2] = {};
    unsigned ldsaddr = (threadIdx.x & 63) * 48;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q + i) & 7]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[(q + i) & 3]) : "v"(pk[4]));
                if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(v[(q + i) & 7]) : "v"(a));
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(q + i) & 1]) : "v"(ldsaddr));
            }
        }
    }
    float s = 0;
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + pk[i & 3][0] + pk[i & 3][1] + ld[i & 1][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool MF, int KIND = 0>
void run(int threads) {
    float* out;
    const int blocks = 256;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, MF, KIND>), dim3(blocks), dim3(threads), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, MF, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 8;            // MFMAs per wave
    printf("kind=%d waves/SIMD=%d  ops per MFMA=%d  mfma=%d : %.3f ms  = %.1f ns per (MFMA + %d VALU) per wave\n", KIND, threads / 256, NV, (int)MF, ms,
           ms * 1e6 / nm, NV);
    hipFree(out);
}

int main() {
    for (int t = 256; t <= 512; t += 256) {
        run<0, true>(t);
        run<1, true>(t);
        run<2, true>(t);
        run<4, true>(t);
        run<6, true>(t);
        run<8, true>(t);
        run<4, false>(t);
        run<8, false>(t);
        run<4, true, 1>(t);
        run<4, false, 1>(t);
        run<4, true, 2>(t);
        run<2, true, 3>(t);
        run<2, false, 3>(t);
    }
    return 0;
}
