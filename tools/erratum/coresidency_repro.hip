// Reduced reproducer of the round-3 co-residency finding: a kernel containing packed-fp32 instructions returns wrong sums when
// it runs on a second stream beside the library's bf16 conv (v_mfma_f32_32x32x16_bf16, csrc/conv_bf16p.hip).
// VICTIM (this file, ~35 lines): a (5,3) conv with one output channel on the vector ALU, 4 time steps per thread as float4
// arithmetic - what csrc/conv_fewco.hip does.  Built with hipcc's defaults the SLP vectoriser / packed-fp32 selection turn
// the float4 arithmetic into v_pk_fma_f32 / v_pk_mul_f32 fed by v_pk_mov_b32 ... op_sel; built with -DNO_PK flags (see below)
// the same source has no packed instruction.  AGGRESSOR: the product library's babe_conv2d_bf16 through its C-ABI.
// Protocol: victim alone -> reference; then N times { aggressor on stream B, victim on stream A, aggressor on stream B },
// victim outputs compared bit for bit with the reference.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 tools/erratum/coresidency_repro.hip -o tools/bin/coresidency_repro -Lbabe_amd -lbabe_hip -Wl,-rpath,$PWD/babe_amd
//   control: add  -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops  -o tools/bin/coresidency_repro_nopk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/babe_hip.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                              int C, int F, int T) {
    extern __shared__ f32x4 wl[];                                  // [C][5] taps {w0, w1, w2, 0}
    for (int i = threadIdx.x; i < C * 5; i += 256) wl[i] = f32x4{w[i * 3], w[i * 3 + 1], w[i * 3 + 2], 0.f};
    __syncthreads();
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
            const f32x4 wv = wl[c * 5 + kh];
            const f32x4 left = {xl, xc[0], xc[1], xc[2]}, right = {xc[1], xc[2], xc[3], xr};
            acc += wv[0] * left + wv[1] * xc + wv[2] * right;
        }
    *reinterpret_cast<f32x4*>(y + (long)f * T + t) = acc;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 30;
    // victim problem: the deepest pyramid VJP shape of the UNet (256 channels, 448 x 64)
    const int C = 256, F = 448, T = 64;
    std::vector<float> hx((size_t)C * F * T), hw((size_t)C * 15);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    float *x, *w, *y, *yref;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&y, (size_t)F * T * 4)); CK(hipMalloc(&yref, (size_t)F * T * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    // aggressor: babe_conv2d_bf16, AC -> AC channels, (5,3), B = 2; default = the 64-channel layer of the UNet (64 x 4096, dil 1);
    // argv: trials AC AF AT dil
    const int AC = argc > 2 ? atoi(argv[2]) : 64, AB = 2, AF = argc > 3 ? atoi(argv[3]) : 64, AT = argc > 4 ? atoi(argv[4]) : 4096;
    const int adil = argc > 5 ? atoi(argv[5]) : 1;
    float *ax, *aw, *ao; void* awp;
    const size_t an = (size_t)AB * AC * AF * AT;
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
    a.B = AB; a.Cin = AC; a.Cout = AC; a.F = AF; a.T = AT; a.KH = 5; a.KW = 3; a.dil = adil;
    auto run_victim = [&](float* out, hipStream_t st) {
        hipLaunchKernelGGL(victim, dim3((F * T / 4 + 255) / 256), dim3(256), C * 5 * 16, st, x, w, out, C, F, T);
    };
    run_victim(yref, sA);
    CK(hipDeviceSynchronize());
    std::vector<float> href((size_t)F * T), hy((size_t)F * T);
    CK(hipMemcpy(href.data(), yref, href.size() * 4, hipMemcpyDeviceToHost));
    int bad_runs = 0; long bad_elems = 0; double worst = 0;
    for (int i = 0; i < trials; ++i) {
        if (babe_conv2d_bf16(&a, awp, 1, sB)) { printf("aggressor: %s\n", babe_last_error()); return 2; }
        run_victim(y, sA);
        if (babe_conv2d_bf16(&a, awp, 1, sB)) return 2;
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
        long nb = 0;
        for (size_t k = 0; k < hy.size(); ++k)
            if (memcmp(&hy[k], &href[k], 4)) { ++nb; double d = fabs((double)hy[k] - href[k]); if (d > worst) worst = d; }
        bad_runs += nb > 0; bad_elems += nb;
    }
    printf("victim beside babe_conv2d_bf16 (%d ch, %d x %d, dil %d): %d of %d runs differ from the victim alone (%ld elements in total, worst |diff| %.3e)\n",
           AC, AF, AT, adil, bad_runs, trials, bad_elems, worst);
    return 0;
}
