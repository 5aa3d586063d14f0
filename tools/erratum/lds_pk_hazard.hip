// Reduction of the co-residency finding, step 3: is "LDS read -> s_waitcnt lgkmcnt(0) -> packed-fp32 consumer" enough?
// Hand-written sequence (inline asm), run alone and beside the library's 64-channel bf16 conv; LDS word i holds float(i % 1021 + 1),
// the consumer multiplies by exact constants, so the expected result is known exactly.
//   consumer 0: two v_mul_f32                     (control)
//   consumer 1: v_pk_mul_f32 d, x, v[a:a+1]
//   consumer 2: v_pk_mul_f32 d, x, v[a:a+1] op_sel:[0,1]      (the victim's first consumer: both halves times the HIGH word)
//   consumer 3: consumer 2 after s_nop 7
//   consumer 4: v_pk_fma_f32 d, x, v[a:a+1], y op_sel_hi:[1,0,1]
// address mode: per lane, or uniform over the wave (the victim's taps).
// Build: hipcc --offload-arch=gfx950 -O3 tools/erratum/lds_pk_hazard.hip -o tools/bin/lds_pk_hazard -Lbabe_amd -lbabe_hip -Wl,-rpath,$PWD/babe_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../include/babe_hip.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V, bool UNI>
__global__ __launch_bounds__(256) void hz(unsigned* bad, int iters, int units) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < units * 4; i += 256) sm[i] = (float)(i % 1021 + 1);
    __syncthreads();
    unsigned nb = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned u = ((UNI ? 0u : threadIdx.x * 37u) + it * 101u + blockIdx.x) % units;
        const unsigned addr = u * 16;
        const float e0 = (float)((4 * u) % 1021 + 1), e1 = (float)((4 * u + 1) % 1021 + 1);
        f32x2 d, x = {2.f, 4.f}, y = {1.f, 3.f};
        float x0, x1;
#define RD "ds_read_b128 v[100:103], %[a]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define CL "memory", "v100", "v101", "v102", "v103"
        if (V == 0) {
            asm volatile(RD "v_mul_f32 %[x0], 2.0, v100\n\tv_mul_f32 %[x1], 4.0, v100" : [x0] "=&v"(x0), [x1] "=&v"(x1) : [a] "v"(addr) : CL);
            nb += (x0 != 2.f * e0) + (x1 != 4.f * e0);
        } else if (V == 1) {
            asm volatile(RD "v_pk_mul_f32 %[d], %[x], v[100:101]" : [d] "=&v"(d) : [a] "v"(addr), [x] "v"(x) : CL);
            nb += (d[0] != 2.f * e0) + (d[1] != 4.f * e1);
        } else if (V == 2) {
            asm volatile(RD "v_pk_mul_f32 %[d], %[x], v[100:101] op_sel:[0,1]" : [d] "=&v"(d) : [a] "v"(addr), [x] "v"(x) : CL);
            nb += (d[0] != 2.f * e1) + (d[1] != 4.f * e1);
        } else if (V == 3) {
            asm volatile(RD "s_nop 7\n\tv_pk_mul_f32 %[d], %[x], v[100:101] op_sel:[0,1]" : [d] "=&v"(d) : [a] "v"(addr), [x] "v"(x) : CL);
            nb += (d[0] != 2.f * e1) + (d[1] != 4.f * e1);
        } else {
            asm volatile(RD "v_pk_fma_f32 %[d], %[x], v[100:101], %[y] op_sel_hi:[1,0,1]" : [d] "=&v"(d) : [a] "v"(addr), [x] "v"(x), [y] "v"(y) : CL);
            nb += (d[0] != 2.f * e0 + 1.f) + (d[1] != 4.f * e0 + 3.f);
        }
    }
    if (nb) atomicAdd(bad, nb);
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 2; } } while (0)
template <int V, bool UNI>
void launch(unsigned* bad, hipStream_t st) { hipLaunchKernelGGL((hz<V, UNI>), dim3(2048), dim3(256), 20480, st, bad, 400, 1280); }

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 20;
    unsigned* bad;
    CK(hipMalloc(&bad, 4));
    const int AC = 64, AB = 2, AF = 64, AT = 4096;
    const size_t an = (size_t)AB * AC * AF * AT;
    float *ax, *aw, *ao; void* awp;
    CK(hipMalloc(&ax, an * 4)); CK(hipMalloc(&ao, an * 4)); CK(hipMalloc(&aw, (size_t)AC * AC * 15 * 4));
    CK(hipMemset(ax, 0x3c, an * 4)); CK(hipMemset(aw, 0x3c, (size_t)AC * AC * 15 * 4));
    CK(hipMalloc(&awp, (size_t)babe_conv_packed_size_bf16(AC, AC, 5, 3, 0, 1) * 2));
    hipStream_t sA, sB;
    CK(hipStreamCreate(&sA)); CK(hipStreamCreate(&sB));
    if (babe_conv_pack_weights_bf16(aw, awp, AC, AC, 5, 3, 0, 1, sB)) { printf("pack: %s\n", babe_last_error()); return 2; }
    babe_conv_args a;
    memset(&a, 0, sizeof a);
    a.in = ax; a.in_bs = (long)AC * AF * AT; a.in_cs = (long)AF * AT; a.cin_split = AC;
    a.out = ao; a.out_bs = (long)AC * AF * AT; a.out_cs = (long)AF * AT; a.alpha = 1.f;
    a.B = AB; a.Cin = AC; a.Cout = AC; a.F = AF; a.T = AT; a.KH = 5; a.KW = 3; a.dil = 1;
    const char* names[] = {"0 v_mul_f32 x2 (control)", "1 v_pk_mul_f32", "2 v_pk_mul_f32 op_sel:[0,1]", "3 s_nop 7 + v_pk_mul_f32 op_sel:[0,1]", "4 v_pk_fma_f32 op_sel_hi:[1,0,1]"};
    void (*fn[])(unsigned*, hipStream_t) = {launch<0, false>, launch<1, false>, launch<2, false>, launch<3, false>, launch<4, false>,
                                            launch<0, true>, launch<1, true>, launch<2, true>, launch<3, true>, launch<4, true>};
    for (int v = 0; v < 10; ++v) {
        CK(hipMemset(bad, 0, 4));
        fn[v](bad, sA);
        CK(hipDeviceSynchronize());
        unsigned alone = 0, beside = 0;
        CK(hipMemcpy(&alone, bad, 4, hipMemcpyDeviceToHost));
        int bad_runs = 0;
        for (int i = 0; i < trials; ++i) {
            CK(hipMemset(bad, 0, 4));
            if (babe_conv2d_bf16(&a, awp, 1, sB)) { printf("aggressor: %s\n", babe_last_error()); return 2; }
            fn[v](bad, sA);
            if (babe_conv2d_bf16(&a, awp, 1, sB)) return 2;
            CK(hipDeviceSynchronize());
            unsigned hb;
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            beside += hb; bad_runs += hb > 0;
        }
        printf("%-40s %-8s: alone %u wrong; beside the bf16 conv %d of %d runs wrong (%u wrong values)\n", names[v % 5], v < 5 ? "per-lane" : "uniform", alone, bad_runs, trials, beside);
    }
    return 0;
}
