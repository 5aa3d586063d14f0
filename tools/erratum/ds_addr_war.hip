// Is overwriting the ADDRESS register of an LDS read before s_waitcnt lgkmcnt(0) safe on gfx950?
// hipcc emits exactly that in the packed-fp32 build of tools/erratum/coresidency_repro.hip's victim:
//     ds_read_b96 v[30:32], v22 ; s_waitcnt vmcnt(0) ; ... ; v_mov_b32 v22, v7 ; s_waitcnt lgkmcnt(0)
// This program runs that sequence by hand (inline asm), alone and beside the library's 64-channel bf16 conv, and checks the data
// the read returned against the LDS contents (word i holds i).  Variants:
//   0 control : address register untouched until after the wait
//   1 b96     : ds_read_b96, address overwritten (with address 0) right after the read is issued
//   2 b96+vm  : the victim's shape: a global load in flight, s_waitcnt vmcnt(0) between the read and the overwrite
//   3 b128    : variant 1 with ds_read_b128
//   4 b32     : variant 1 with ds_read_b32
// Build: hipcc --offload-arch=gfx950 -O3 tools/erratum/ds_addr_war.hip -o tools/bin/ds_addr_war -Lbabe_amd -lbabe_hip -Wl,-rpath,$PWD/babe_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../include/babe_hip.h"
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

template <int V>
__global__ __launch_bounds__(256) void war(const unsigned* __restrict__ g, unsigned* bad, int iters, int units) {
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < units * 4; i += 256) sm[i] = i;
    __syncthreads();
    unsigned nb = 0;
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const unsigned u = (threadIdx.x * 37u + it * 101u) % units;
        unsigned addr = u * 16, zero = 0, gv = 0;
        u32x4 r = {0, 0, 0, 0};
        if (V == 0) {
            u32x3 t;
            asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %1, %2" : "=&v"(t), "+v"(addr) : "v"(zero) : "memory");
            r = u32x4{t[0], t[1], t[2], 4 * u + 3};
        } else if (V == 1) {
            u32x3 t;
            asm volatile("ds_read_b96 %0, %1\n\tv_mov_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t), "+v"(addr) : "v"(zero) : "memory");
            r = u32x4{t[0], t[1], t[2], 4 * u + 3};
        } else if (V == 2) {
            u32x3 t;
            const unsigned* gp = g + ((tid * 64u + it * 4099u) & 0xfffffu);
            asm volatile("global_load_dword %3, %4, off\n\tds_read_b96 %0, %1\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t), "+v"(addr), "+v"(zero), "=&v"(gv) : "v"(gp) : "memory");
            r = u32x4{t[0], t[1], t[2], 4 * u + 3 + gv};
        } else if (V == 3) {
            asm volatile("ds_read_b128 %0, %1\n\tv_mov_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r), "+v"(addr) : "v"(zero) : "memory");
        } else {
            unsigned t;
            asm volatile("ds_read_b32 %0, %1\n\tv_mov_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t), "+v"(addr) : "v"(zero) : "memory");
            r = u32x4{t, 4 * u + 1, 4 * u + 2, 4 * u + 3};
        }
        nb += (r[0] != 4 * u) + (r[1] != 4 * u + 1) + (r[2] != 4 * u + 2) + (r[3] != 4 * u + 3) + (addr != 0);
    }
    if (nb) atomicAdd(bad, nb);
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 2; } } while (0)

template <int V>
void launch(const unsigned* g, unsigned* bad, hipStream_t st) { hipLaunchKernelGGL(war<V>, dim3(2048), dim3(256), 20480, st, g, bad, 400, 1280); }

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 20;
    unsigned *g, *bad;
    CK(hipMalloc(&g, (1 << 20) * 4 + 64)); CK(hipMemset(g, 0, (1 << 20) * 4 + 64)); CK(hipMalloc(&bad, 4));
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
    const char* names[] = {"0 control (overwrite after the wait)", "1 ds_read_b96, address overwritten at once", "2 ds_read_b96, vmcnt(0), overwrite (the victim)",
                           "3 ds_read_b128, overwritten at once", "4 ds_read_b32, overwritten at once"};
    void (*fn[])(const unsigned*, unsigned*, hipStream_t) = {launch<0>, launch<1>, launch<2>, launch<3>, launch<4>};
    for (int v = 0; v < 5; ++v) {
        CK(hipMemset(bad, 0, 4));
        fn[v](g, bad, sA);
        CK(hipDeviceSynchronize());
        unsigned alone = 0, beside = 0;
        CK(hipMemcpy(&alone, bad, 4, hipMemcpyDeviceToHost));
        int bad_runs = 0;
        for (int i = 0; i < trials; ++i) {
            CK(hipMemset(bad, 0, 4));
            if (babe_conv2d_bf16(&a, awp, 1, sB)) { printf("aggressor: %s\n", babe_last_error()); return 2; }
            fn[v](g, bad, sA);
            if (babe_conv2d_bf16(&a, awp, 1, sB)) return 2;
            CK(hipDeviceSynchronize());
            unsigned hb;
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            beside += hb; bad_runs += hb > 0;
        }
        printf("%-52s: alone %u wrong words; beside the bf16 conv %d of %d runs wrong (%u wrong words)\n", names[v], alone, bad_runs, trials, beside);
    }
    return 0;
}
