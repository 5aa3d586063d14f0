// What does the operand feed of conv_wino45x cost the fp32 matrix pipe?  (round 4 calibration)
// The kernel's MFMA group: 3 ds_read_b128 (A of the 16 output channels + B of two 16-unit segments: 4 phases each) feed 8
// v_mfma_f32_16x16x4_f32; two operand register sets (the reads of group g + 1 are issued before the MFMAs of group g); 8 waves
// per workgroup, one workgroup per CU (two waves per SIMD).  This program runs that loop and nothing else, in variants:
//   0  MFMAs only (operands in registers, no LDS)                                  -> the pipe's own ceiling
//   1  the kernel's pattern (3 reads clustered before the 8 MFMAs of the previous group)
//   2  reads spread: one ds_read after every second MFMA
//   3  pattern 1 + s_setprio 1 around the MFMAs
//   4  pattern 1 with the accumulators in AGPRs
//   5  pattern 1, ONE wave per SIMD (256 threads)
//   6  pattern 1 with ds_read_b64 pairs instead of b128
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_feed.hip -o tools/bin/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* sm = reinterpret_cast<f32x4*>(smem_f);
    for (int i = threadIdx.x; i < 9216; i += NT) sm[i] = f32x4{0.001f * i, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int aoff = 3072 + wave * 192 + (lk * 16 + l15) * 3;      // A: wave-private ring part
    const int boff = (lk * 32 + l15) * 3;                          // B: shared X
    f32x4 acc[2][12];
    for (int i = 0; i < 2; ++i)
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av[2], bv[2][2];
    av[0] = sm[aoff];
    bv[0][0] = sm[boff];
    bv[0][1] = sm[boff + 48];
    av[1] = av[0];
    bv[1][0] = bv[0][0];
    bv[1][1] = bv[0][1];
#define MF(c, pg, i, s) acc[s][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][s][i], acc[s][4 * (pg) + i], 0, 0, 0);
#define FENCE __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int c = g & 1, n = c ^ 1, pg = g % 3, sl = g / 3;
            const int ao = aoff + sl * 1536 + ((g + 1) % 3), bo = boff + sl * 384 + ((g + 1) % 3);
            if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {
                av[n] = sm[ao];
                bv[n][0] = sm[bo];
                bv[n][1] = sm[bo + 48];
                FENCE
            }
            if (MODE == 6) {
                const f32x2* s2 = reinterpret_cast<const f32x2*>(sm);
                f32x2 t0 = s2[2 * ao], t1 = s2[2 * ao + 1], t2 = s2[2 * bo], t3 = s2[2 * bo + 1], t4 = s2[2 * bo + 96], t5 = s2[2 * bo + 97];
                av[n] = f32x4{t0[0], t0[1], t1[0], t1[1]};
                bv[n][0] = f32x4{t2[0], t2[1], t3[0], t3[1]};
                bv[n][1] = f32x4{t4[0], t4[1], t5[0], t5[1]};
                FENCE
            }
            if (MODE == 3) __builtin_amdgcn_s_setprio(1);
            if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[0][4 * pg + i]) : "v"(av[c][i]), "v"(bv[c][0][i]));
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[1][4 * pg + i]) : "v"(av[c][i]), "v"(bv[c][1][i]));
                }
            } else if (MODE == 2) {
                MF(c, pg, 0, 0) MF(c, pg, 0, 1)
                FENCE
                av[n] = sm[ao];
                FENCE
                MF(c, pg, 1, 0) MF(c, pg, 1, 1)
                FENCE
                bv[n][0] = sm[bo];
                FENCE
                MF(c, pg, 2, 0) MF(c, pg, 2, 1)
                FENCE
                bv[n][1] = sm[bo + 48];
                FENCE
                MF(c, pg, 3, 0) MF(c, pg, 3, 1)
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    MF(c, pg, i, 0)
                    MF(c, pg, i, 1)
                }
            }
            if (MODE == 3) __builtin_amdgcn_s_setprio(0);
            FENCE
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int p = 0; p < 12; ++p) s += acc[i][p][0] + acc[i][p][1] + acc[i][p][2] + acc[i][p][3];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int MODE, int NT>
void run(const char* name) {
    float* out;
    const int blocks = 256;
    (void)hipMalloc(&out, blocks * NT * sizeof(float));
    const size_t lds = 9216 * 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(NT), lds, 0, out, 200);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(NT), lds, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double fl = (double)blocks * (NT / 64) * iters * 96.0 * 2048.0;
    printf("%-64s %7.3f ms  %6.1f TFLOP/s = %.3f of 157.3\n", name, best, fl / best / 1e9, fl / best / 1e9 / 157.3);
    (void)hipFree(out);
}

int main() {
    run<0, 512>("0 MFMAs only, 2 waves/SIMD");
    run<1, 512>("1 kernel pattern (3 b128 reads, then 8 MFMAs of previous set)");
    run<2, 512>("2 reads spread between the MFMAs");
    run<3, 512>("3 pattern 1 + s_setprio around the MFMAs");
    run<4, 512>("4 pattern 1, accumulators in AGPRs");
    run<5, 256>("5 pattern 1, one wave per SIMD");
    run<0, 256>("0 MFMAs only, one wave per SIMD");
    run<6, 512>("6 pattern 1 with ds_read_b64 pairs");
    return 0;
}
