// LDS canary (debugging tool, not part of the product): every workgroup fills `words` dwords of dynamic LDS with a pattern that
// encodes (workgroup, index), spins, re-reads, and reports every dword that changed: out[0] = count (atomic), then up to 1023
// records {workgroup, dword index, expected, found}.  Run beside another kernel on a second stream: foreign LDS writes show up here.
#include <hip/hip_runtime.h>
extern "C" __global__ void lds_canary_kernel(unsigned* out, int words, long spin) {
    extern __shared__ unsigned lds[];
    const unsigned wg = blockIdx.x;
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xC0000000u | (wg << 16) | (unsigned)(i & 0xffff);
    __syncthreads();
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const unsigned exp = 0xC0000000u | (wg << 16) | (unsigned)(i & 0xffff), got = lds[i];
        if (got != exp) {
            const unsigned k = atomicAdd(out, 1u);
            if (k < 1023) {
                out[4 + 4 * k] = wg;
                out[5 + 4 * k] = (unsigned)i;
                out[6 + 4 * k] = exp;
                out[7 + 4 * k] = got;
            }
        }
    }
}
extern "C" int lds_canary(unsigned* out, int blocks, int threads, int words, long spin, void* stream) {
    hipFuncSetAttribute((const void*)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, words * 4);
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(threads), words * 4, (hipStream_t)stream, out, words, spin);
    return (int)hipGetLastError();
}

// VALU canary: every lane runs the same long chain of FMAs in NR independent registers (values depend only on the register
// index and the iteration count, so every lane of every wave must end with identical numbers) and reports lanes whose result
// differs from lane 0 of workgroup 0's reference computed by the same code.  out[0] = mismatch count, then records
// {workgroup, thread, register, found bits}; out[2] receives the reference bits of register 0.
extern "C" __global__ void valu_canary_kernel(unsigned* out, int iters) {
    constexpr int NR = 48;
    float r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = 1.0f + 0.001f * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = __builtin_fmaf(r[i], 0.999f, 0.0011f * (float)((i + it) & 7));
    }
    // reference: recompute with the SAME code path in a way the compiler cannot merge (volatile seed)
    volatile float seed = 1.0f;
    float q[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) q[i] = seed + 0.001f * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) q[i] = __builtin_fmaf(q[i], 0.999f, 0.0011f * (float)((i + it) & 7));
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        if (__float_as_uint(r[i]) != __float_as_uint(q[i])) {
            const unsigned k = atomicAdd(out, 1u);
            if (k < 1023) {
                out[4 + 4 * k] = blockIdx.x;
                out[5 + 4 * k] = threadIdx.x;
                out[6 + 4 * k] = (unsigned)i;
                out[7 + 4 * k] = __float_as_uint(r[i]) ^ __float_as_uint(q[i]);
            }
        }
    }
}
extern "C" int valu_canary(unsigned* out, int blocks, int threads, int iters, void* stream) {
    hipLaunchKernelGGL(valu_canary_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}

// packed-fp32 canary: the same idea with v_pk_fma_f32 / v_pk_mul_f32 (plain and with op_sel swizzles, as hipcc emits them
// for float4 arithmetic): two identical chains per lane must agree bit for bit.
typedef float cf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf2 pk_step(cf2 r, cf2 c, cf2 d) {
    cf2 t, u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(r), "v"(c));                       // (r.x c.x, r.x c.y)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(u) : "v"(r), "v"(d), "v"(t));   // (r.x d.x + t.y, r.y d.y + t.x)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(u), "v"(c), "v"(r));         // (u.x c.x + r.x, u.y c.x + r.y)
    // the forms hipcc uses to build shifted pairs and to add with an inline constant
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(u) : "v"(t), "v"(r));                          // (t.y, r.x)
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(u));
    asm volatile("v_pk_add_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(t) : "v"(t));
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(c));
    return t;
}
extern "C" __global__ void pk_canary_kernel(unsigned* out, int iters) {
    constexpr int NR = 24;
    cf2 r[NR], q[NR];
    volatile float seed = 1.0f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        r[i] = cf2{1.0f + 0.001f * i, 0.5f + 0.002f * i};
        q[i] = cf2{seed + 0.001f * i, seed * 0.5f + 0.002f * i};
    }
    const cf2 c = {0.4993f, 0.0007f}, d = {0.4991f, 0.4989f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = pk_step(r[i], c, d);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) q[i] = pk_step(q[i], c, d);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const unsigned x0 = __float_as_uint(r[i].x) ^ __float_as_uint(q[i].x), x1 = __float_as_uint(r[i].y) ^ __float_as_uint(q[i].y);
        if (x0 | x1) {
            const unsigned k = atomicAdd(out, 1u);
            if (k < 1023) {
                out[4 + 4 * k] = blockIdx.x;
                out[5 + 4 * k] = threadIdx.x;
                out[6 + 4 * k] = (unsigned)i | (x0 ? 0x100u : 0u) | (x1 ? 0x200u : 0u);
                out[7 + 4 * k] = x0 | x1;
            }
        }
    }
}
extern "C" int pk_canary(unsigned* out, int blocks, int threads, int iters, void* stream) {
    hipLaunchKernelGGL(pk_canary_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
