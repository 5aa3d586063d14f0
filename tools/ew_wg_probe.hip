// How many workgroups should an HBM-bound element-wise pass use when it runs BESIDE the other lane's conv kernel?
// The nested conv workgroup owns its CU (2 x 248 registers per SIMD, 144 KB LDS): nothing co-resides, so CU-time is additive
// and an element-wise kernel that spreads thin over all 256 CUs takes 256 x t of it.  This probe measures
//   (1) streaming bandwidth of a 12 B/element pass (2 reads + 1 write, the GroupNorm-VJP shape) vs number of workgroups,
//       threads per workgroup and loads in flight per thread, alone on the GPU;
//   (2) the makespan of a "hog" kernel (512 threads, 248 registers, 144 KB LDS, ~50 us of MFMAs per workgroup, 1024
//       workgroups = 4 rounds) on one stream with the element-wise pass repeated on a second stream, for thin and fat grids.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ew_wg_probe.hip -o tools/bin/ew_wg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(1024) void ew_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ y, f32x4* __restrict__ o, long nv) {
    // contiguous range per workgroup, U independent 16-byte loads per tensor in flight per thread
    const long per = (nv + gridDim.x - 1) / gridDim.x;
    const long beg = (long)blockIdx.x * per;
    long end = beg + per;
    if (end > nv) end = nv;
    const long st = blockDim.x;
    long i = beg + threadIdx.x;
    for (; i + (U - 1) * st < end; i += U * st) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = x[i + u * st];
            b[u] = y[i + u * st];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) o[i + u * st] = a[u] * 0.5f + b[u];
    }
    for (; i < end; i += st) o[i] = x[i] * 0.5f + y[i];
}

__global__ __launch_bounds__(512, 1) void hog_kernel(float* out, int iters) {
    extern __shared__ float sm[];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 0.01f, b = 1.0f;
    asm volatile("v_mov_b32 v247, 0" ::: "v247");          // the wave holds 248 registers like the conv kernel
    sm[threadIdx.x] = a;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = sm[(threadIdx.x + 1) & 511];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

template <int U>
static void launch_ew(int wgs, int th, const f32x4* x, const f32x4* y, f32x4* o, long nv, hipStream_t s) {
    hipLaunchKernelGGL((ew_kernel<U>), dim3(wgs), dim3(th), 0, s, x, y, o, nv);
}
static void launch_ew_u(int U, int wgs, int th, const f32x4* x, const f32x4* y, f32x4* o, long nv, hipStream_t s) {
    if (U == 1) launch_ew<1>(wgs, th, x, y, o, nv, s);
    else if (U == 2) launch_ew<2>(wgs, th, x, y, o, nv, s);
    else if (U == 4) launch_ew<4>(wgs, th, x, y, o, nv, s);
    else launch_ew<8>(wgs, th, x, y, o, nv, s);
}

int main() {
    const long n = 25L << 20;                      // 25 M elements = 100 MB per tensor (the 96-channel level)
    const long nv = n / 4;
    f32x4 *x, *y, *o;
    float* hout;
    hipMalloc(&x, n * 4);
    hipMalloc(&y, n * 4);
    hipMalloc(&o, n * 4);
    hipMalloc(&hout, 4096 * 512 * 4);
    hipMemset(x, 0, n * 4);
    hipMemset(y, 0, n * 4);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, e2, e3;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventCreate(&e2);
    hipEventCreate(&e3);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);

    printf("== (1) 12 B/element pass alone: %ld MB moved per launch\n", 3 * n * 4 >> 20);
    const int wgl[] = {32, 64, 96, 128, 192, 256, 512, 1024, 4096, 16384};
    for (int th : {256, 512, 1024})
        for (int U : {1, 4, 8})
            for (int wgs : wgl) {
                for (int r = 0; r < 2; ++r) launch_ew_u(U, wgs, th, x, y, o, nv, sa);
                hipEventRecord(e0, sa);
                for (int r = 0; r < 5; ++r) launch_ew_u(U, wgs, th, x, y, o, nv, sa);
                hipEventRecord(e1, sa);
                hipEventSynchronize(e1);
                const float ms = elapsed(e0, e1) / 5;
                printf("threads %4d  U %d  wgs %5d : %7.1f us  %5.2f TB/s\n", th, U, wgs, ms * 1e3, 3.0 * n * 4 / ms * 1e-9);
            }

    // hog calibration: iterations for ~50 us per workgroup
    int iters = 600;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, sa);
        hipLaunchKernelGGL(hog_kernel, dim3(256), dim3(512), 144 * 1024, sa, hout, iters);
        hipEventRecord(e1, sa);
        hipEventSynchronize(e1);
        const float us = elapsed(e0, e1) * 1e3f;
        if (r < 2) iters = (int)(iters * 50.f / us);
        else printf("== hog: %d iterations = %.1f us per round of 256 workgroups\n", iters, us);
    }
    const int HOGS = 8, HWG = 1024;                 // 8 launches x 4 rounds x 50 us = 1.6 ms of conv-like work
    auto hog_run = [&]() {
        for (int r = 0; r < HOGS; ++r) hipLaunchKernelGGL(hog_kernel, dim3(HWG), dim3(512), 144 * 1024, sa, hout, iters);
    };
    hog_run();
    hipDeviceSynchronize();
    hipEventRecord(e0, sa);
    hog_run();
    hipEventRecord(e1, sa);
    hipEventSynchronize(e1);
    const float hog_ms = elapsed(e0, e1);
    printf("== (2) hog alone: %.3f ms\n", hog_ms);
    const int EW = 12;                              // element-wise launches beside it (12 x 300 MB)
    for (int th : {256, 1024})
        for (int U : {1, 4, 8})
            for (int wgs : {32, 64, 96, 128, 192, 256, 1024, 16384}) {
                // alone
                hipEventRecord(e0, sb);
                for (int r = 0; r < EW; ++r) launch_ew_u(U, wgs, th, x, y, o, nv, sb);
                hipEventRecord(e1, sb);
                hipEventSynchronize(e1);
                const float ew_ms = elapsed(e0, e1);
                // together
                hipDeviceSynchronize();
                hipEventRecord(e0, sa);
                hipStreamWaitEvent(sb, e0, 0);
                hog_run();
                for (int r = 0; r < EW; ++r) launch_ew_u(U, wgs, th, x, y, o, nv, sb);
                hipEventRecord(e1, sa);
                hipEventRecord(e2, sb);
                hipEventSynchronize(e1);
                hipEventSynchronize(e2);
                const float ta = elapsed(e0, e1), tb = elapsed(e0, e2);
                const float mk = ta > tb ? ta : tb;
                printf("threads %4d U %d wgs %5d : ew alone %.3f ms | together: hog done %.3f, ew done %.3f, makespan %.3f ms "
                       "(sum alone %.3f; overlap gain %.3f ms)\n", th, U, wgs, ew_ms, ta, tb, mk, hog_ms + ew_ms, hog_ms + ew_ms - mk);
            }
    return 0;
}
