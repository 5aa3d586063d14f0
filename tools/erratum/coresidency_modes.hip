// Where does the co-residency corruption come from?  Three reduced victims beside the library's 64-channel bf16 conv
// (babe_conv2d_bf16, conv_bf16p G = 2: 256 threads, 57 KB LDS, two workgroups per CU), see tools/erratum/coresidency_repro.hip:
//   lds_watch  : every workgroup writes a pattern to 20 KB of LDS and re-reads it for a while: mismatches = somebody else wrote
//                into this workgroup's LDS
//   pk_alu     : registers only, v_pk_fma_f32 chains (no LDS, no memory in the loop): wrong = the arithmetic itself is disturbed
//   conv_nolds : the repro victim with its taps read from global memory (scalar loads) instead of LDS
// Build: hipcc --offload-arch=gfx950 -O3 tools/erratum/coresidency_modes.hip -o tools/bin/coresidency_modes -Lbabe_amd -lbabe_hip -Wl,-rpath,$PWD/babe_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/babe_hip.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void lds_watch(unsigned* bad, int iters, int words) {
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < words; i += 256) sm[i] = 0x9e3779b9u * (i + 1) ^ blockIdx.x;
    __syncthreads();
    unsigned nb = 0;
    for (int it = 0; it < iters; ++it)
        for (int i = threadIdx.x; i < words; i += 256) nb += (sm[i] != (0x9e3779b9u * (i + 1) ^ blockIdx.x));
    if (nb) atomicAdd(bad, nb);
}

__global__ __launch_bounds__(256) void pk_alu(float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[8], m = {0.999f, 1.001f}, k = {1e-3f * (tid & 31), -1e-3f};
    for (int j = 0; j < 8; ++j) acc[j] = f32x2{(float)tid * 1e-4f + j, (float)j - 3.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_elementwise_fma(acc[j], m, k);
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1];
    out[tid] = s;
}

// VAR (LDS taps only): 0 as is (ds_read_b96); 1 the 4th component is used (ds_read_b128); 2 wait + 16 idle cycles after the read;
// 3 the three taps are copied through v_mov_b32 before use; 4 fill with scalar ds_write_b32
template <bool LDS, int VAR = 0>
__global__ __launch_bounds__(256) void conv_victim(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                   int C, int F, int T) {
    extern __shared__ f32x4 wl[];
    if (LDS) {
        if (VAR == 4) {
            float* wf = reinterpret_cast<float*>(wl);
            for (int i = threadIdx.x; i < C * 20; i += 256) wf[i] = (i & 3) == 3 ? 0.f : w[(i >> 2) * 3 + (i & 3)];
        } else
            for (int i = threadIdx.x; i < C * 5; i += 256) wl[i] = f32x4{w[i * 3], w[i * 3 + 1], w[i * 3 + 2], 0.f};
        __syncthreads();
    }
    const int q = blockIdx.x * 256 + threadIdx.x, q4 = T / 4;
    if (q >= F * q4) return;
    const int f = q / q4, t = (q % q4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c)
        for (int kh = 0; kh < 5; ++kh) {
            const int fr = f + kh - 2;
            if (fr < 0 || fr >= F) continue;
            const float* row = x + ((long)c * F + fr) * T + t;
            const f32x4 xc = *reinterpret_cast<const f32x4*>(row);
            const float xl = t > 0 ? row[-1] : 0.f, xr = t + 4 < T ? row[4] : 0.f;
            f32x4 wv = LDS ? wl[c * 5 + kh] : f32x4{w[(c * 5 + kh) * 3], w[(c * 5 + kh) * 3 + 1], w[(c * 5 + kh) * 3 + 2], 0.f};
            if (VAR == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]));
            if (VAR == 3) {
                float a0, a1, a2;
                asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(a0), "=&v"(a1), "=&v"(a2) : "v"(wv[0]), "v"(wv[1]), "v"(wv[2]));
                wv = f32x4{a0, a1, a2, 0.f};
            }
            const f32x4 left = {xl, xc[0], xc[1], xc[2]}, right = {xc[1], xc[2], xc[3], xr};
            acc += wv[0] * left + wv[1] * xc + wv[2] * right;
            if (VAR == 1) acc += wv[3] * xc;
        }
    *reinterpret_cast<f32x4*>(y + (long)f * T + t) = acc;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 30;
    const int C = 256, F = 448, T = 64;
    std::vector<float> hx((size_t)C * F * T), hw((size_t)C * 15);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    const size_t NO = 1 << 20;
    float *x, *w, *y, *yref; unsigned* bad;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&y, NO * 4)); CK(hipMalloc(&yref, NO * 4)); CK(hipMalloc(&bad, 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
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
    const char* names[] = {"lds_watch (20 KB)", "lds_watch (40 KB)", "pk_alu", "conv victim, taps in LDS", "conv victim, taps from global",
                           "conv victim, LDS, b128 reads", "conv victim, LDS, wait + 16 idle", "conv victim, LDS, taps copied", "conv victim, LDS, b32 fill"};
    for (int mode = 0; mode < 9; ++mode) {
        auto run_victim = [&](float* out, hipStream_t st) {
            if (mode == 0) hipLaunchKernelGGL(lds_watch, dim3(1024), dim3(256), 20480, st, bad, 200, 5120);
            else if (mode == 1) hipLaunchKernelGGL(lds_watch, dim3(1024), dim3(256), 40960, st, bad, 100, 10240);
            else if (mode == 2) hipLaunchKernelGGL(pk_alu, dim3(1024), dim3(256), 0, st, out, 20000);
            else if (mode == 3) hipLaunchKernelGGL(conv_victim<true>, dim3((F * T / 4 + 255) / 256), dim3(256), C * 5 * 16, st, x, w, out, C, F, T);
            else if (mode == 4) hipLaunchKernelGGL(conv_victim<false>, dim3((F * T / 4 + 255) / 256), dim3(256), 0, st, x, w, out, C, F, T);
            else if (mode == 5) hipLaunchKernelGGL((conv_victim<true, 1>), dim3((F * T / 4 + 255) / 256), dim3(256), C * 5 * 16, st, x, w, out, C, F, T);
            else if (mode == 6) hipLaunchKernelGGL((conv_victim<true, 2>), dim3((F * T / 4 + 255) / 256), dim3(256), C * 5 * 16, st, x, w, out, C, F, T);
            else if (mode == 7) hipLaunchKernelGGL((conv_victim<true, 3>), dim3((F * T / 4 + 255) / 256), dim3(256), C * 5 * 16, st, x, w, out, C, F, T);
            else hipLaunchKernelGGL((conv_victim<true, 4>), dim3((F * T / 4 + 255) / 256), dim3(256), C * 5 * 16, st, x, w, out, C, F, T);
        };
        const size_t n = mode == 2 ? 1024 * 256 : (size_t)F * T;
        CK(hipMemset(bad, 0, 4));
        run_victim(yref, sA);
        CK(hipDeviceSynchronize());
        unsigned hb0 = 0;
        CK(hipMemcpy(&hb0, bad, 4, hipMemcpyDeviceToHost));
        std::vector<float> href(n), hy(n);
        CK(hipMemcpy(href.data(), yref, n * 4, hipMemcpyDeviceToHost));
        int bad_runs = 0; long bad_elems = 0;
        for (int i = 0; i < trials; ++i) {
            CK(hipMemset(bad, 0, 4));
            if (babe_conv2d_bf16(&a, awp, 1, sB)) { printf("aggressor: %s\n", babe_last_error()); return 2; }
            run_victim(y, sA);
            if (babe_conv2d_bf16(&a, awp, 1, sB)) return 2;
            CK(hipDeviceSynchronize());
            long nb = 0;
            if (mode < 2) { unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); nb = hb; }
            else {
                CK(hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost));
                for (size_t k = 0; k < n; ++k) nb += memcmp(&hy[k], &href[k], 4) != 0;
            }
            bad_runs += nb > 0; bad_elems += nb;
        }
        printf("%-32s: alone %u mismatches; beside the bf16 conv %d of %d runs wrong (%ld elements / mismatches)\n", names[mode], hb0, bad_runs, trials, bad_elems);
    }
    return 0;
}
