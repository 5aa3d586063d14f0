// BiasFreeGroupNorm (no mean removal) + FiLM + exact GELU, forward and input-VJP.
// Reference: /root/reference/networks/cqtdiff+.py:147-163 (torch.std, unbiased) and :472-482.
//   fwd:  a = gelu( x / (std_g + eps) * gamma_c * (film_c + 1) )  =  gelu(x * scale[b][c])
//   vjp:  du = da * gelu'(x*scale) (formed on the fly in both passes, never stored);
//         gx = scale*du - (x-mean_g) * S_g / ((n-1) std (std+eps)^2),
//         S_g = sum_g( du * scale*(std+eps) * x )            
// All of these are HBM-bound streaming kernels: float4 loads, wave-shuffle + LDS tree reductions,
// double accumulation for the statistics.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include "gelu.h"
#include <cstdlib>

namespace {

using babe_gelu::gelu_f;
using babe_gelu::gelu_grad_f;

__device__ __forceinline__ void block_reduce2(double& s0, double& s1, double* sh) {
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[wave * 2] = s0;
        sh[wave * 2 + 1] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            a += sh[w * 2];
            b += sh[w * 2 + 1];
        }
        s0 = a;
        s1 = b;
    }
}

// grid: (S, B*G).  FUSED: the workgroup that finishes a (b, g) group LAST also does gn_finalize's work for it (same reduction
// order over the S partial sums, so the statistics are bit-identical to the two-launch form): one launch per GroupNorm instead
// of two.  "Last" = the S-th arrival at a per-group ticket (atomic add after a device-scope fence); the last workgroup resets the
// ticket to 0, so the buffer needs zeroing once, when it is allocated (one buffer per stream: babe_amd/ops.py).
template <bool FUSED>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                         long n, int S, int* __restrict__ ticket,
                                                         const float* __restrict__ gamma, const float* __restrict__ film,
                                                         long film_bs, float* __restrict__ stats, float* __restrict__ scale,
                                                         int C, int G, float eps) {
    __shared__ double sh[8];
    __shared__ int last_sh;
    __shared__ float r_sh;
    const int s = blockIdx.x;
    const long bg = blockIdx.y;
    const long chunk = ((n / 4 + S - 1) / S) * 4;       // multiple of 4 elements
    const long beg = (long)s * chunk;
    long end = beg + chunk;
    if (end > n) end = n;
    const float* p = x + bg * n;
    double s0 = 0, s1 = 0;
    if (beg < end) {
        const long nv = (end - beg) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(p + beg);
        // four independent 16-byte loads in flight per thread (one per iteration left the 512 workgroups of a launch at
        // 3.9 TB/s: not enough bytes in flight per CU)
        long i = threadIdx.x;
        for (; i + 3 * (long)blockDim.x < nv; i += 4 * (long)blockDim.x) {
            const float4 v0 = p4[i], v1 = p4[i + blockDim.x], v2 = p4[i + 2 * blockDim.x], v3 = p4[i + 3 * blockDim.x];
            s0 += ((double)v0.x + (double)v0.y + (double)v0.z + (double)v0.w) + ((double)v1.x + (double)v1.y + (double)v1.z + (double)v1.w) +
                  ((double)v2.x + (double)v2.y + (double)v2.z + (double)v2.w) + ((double)v3.x + (double)v3.y + (double)v3.z + (double)v3.w);
            s1 += ((double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w) +
                  ((double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w) +
                  ((double)v2.x * v2.x + (double)v2.y * v2.y + (double)v2.z * v2.z + (double)v2.w * v2.w) +
                  ((double)v3.x * v3.x + (double)v3.y * v3.y + (double)v3.z * v3.z + (double)v3.w * v3.w);
        }
        for (; i < nv; i += blockDim.x) {
            const float4 v = p4[i];
            s0 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            s1 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        for (long i = beg + nv * 4 + threadIdx.x; i < end; i += blockDim.x) {
            const double v = p[i];
            s0 += v;
            s1 += v * v;
        }
    }
    block_reduce2(s0, s1, sh);
    if (threadIdx.x == 0) {
        part[(bg * S + s) * 2] = s0;
        part[(bg * S + s) * 2 + 1] = s1;
    }
    if constexpr (FUSED) {
        if (threadIdx.x == 0) {
            __threadfence();                                   // the partial sums above are visible device-wide before the ticket
            const int old = atomicAdd(&ticket[bg], 1);
            last_sh = old == S - 1;
            if (old == S - 1) ticket[bg] = 0;                  // every workgroup of this group has arrived: ready for the next call
        }
        __syncthreads();
        if (!last_sh) return;
        __threadfence();
        // gn_finalize_kernel's arithmetic for this one group: lanes 0-31 add the S partial sums (lane s, s + 32, ...), xor-shuffle
        const int b = (int)(bg / G), g = (int)(bg % G);
        if (threadIdx.x < 32) {
            const volatile double* vp = part;                  // written by other workgroups of THIS launch: not through a stale L1 line
            double t0 = 0, t1 = 0;
            for (int q = threadIdx.x; q < S; q += 32) {
                t0 += vp[(bg * S + q) * 2];
                t1 += vp[(bg * S + q) * 2 + 1];
            }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {
                t0 += __shfl_xor(t0, o, 32);
                t1 += __shfl_xor(t1, o, 32);
            }
            if (threadIdx.x == 0) {
                const double mean = t0 / (double)n;
                double var = (t1 - (double)n * mean * mean) / (double)(n - 1);
                if (var < 0) var = 0;
                const float sd = (float)sqrt(var);
                const float r = 1.f / (sd + eps);
                stats[(b * G + g) * 3 + 0] = (float)mean;
                stats[(b * G + g) * 3 + 1] = sd;
                stats[(b * G + g) * 3 + 2] = r;
                r_sh = r;
            }
        }
        __syncthreads();
        const int cg = C / G;
        for (int j = threadIdx.x; j < cg; j += blockDim.x) {
            const int c = g * cg + j;
            scale[b * C + c] = gamma[c] * (film[(long)b * film_bs + c] + 1.f) * r_sh;
        }
    }
}

// one block per (b); threads over channels
__global__ void gn_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                   const float* __restrict__ film, long film_bs, float* __restrict__ stats,
                                   float* __restrict__ scale, int C, int G, long n, int S, float eps) {
    const int b = blockIdx.x;
    __shared__ float rstd_sh[64];
    // one 32-lane half-wave per group: the S partial sums are read in parallel and combined by shuffles (a serial loop of
    // S dependent-latency loads on 8 lanes made this tiny kernel take 8 us)
    const int sub = threadIdx.x & 31;
    for (int g = threadIdx.x >> 5; g < G; g += blockDim.x >> 5) {
        double s0 = 0, s1 = 0;
        for (int s = sub; s < S; s += 32) {
            s0 += part[((long)(b * G + g) * S + s) * 2];
            s1 += part[((long)(b * G + g) * S + s) * 2 + 1];
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            s0 += __shfl_xor(s0, o, 32);
            s1 += __shfl_xor(s1, o, 32);
        }
        if (sub != 0) continue;
        const double mean = s0 / (double)n;
        double var = (s1 - (double)n * mean * mean) / (double)(n - 1);
        if (var < 0) var = 0;
        const float sd = (float)sqrt(var);
        const float r = 1.f / (sd + eps);
        stats[(b * G + g) * 3 + 0] = (float)mean;
        stats[(b * G + g) * 3 + 1] = sd;
        stats[(b * G + g) * 3 + 2] = r;
        rstd_sh[g] = r;
    }
    __syncthreads();
    const int cg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        scale[b * C + c] = gamma[c] * (film[(long)b * film_bs + c] + 1.f) * rstd_sh[c / cg];
}

// grid: (blocks over hw/4, C, B)
__global__ __launch_bounds__(256) void scale_gelu_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         float* __restrict__ a, int C, long hw) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float sc = scale[b * C + c];
    const long base = ((long)b * C + c) * hw;
    const long nv = hw / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    float4* a4 = reinterpret_cast<float4*>(a + base);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        float4 v = x4[i];
        v.x = gelu_f(v.x * sc);
        v.y = gelu_f(v.y * sc);
        v.z = gelu_f(v.z * sc);
        v.w = gelu_f(v.w * sc);
        a4[i] = v;
    }
    if (blockIdx.x == 0)
        for (long i = nv * 4 + threadIdx.x; i < hw; i += blockDim.x) a[base + i] = gelu_f(x[base + i] * sc);
}

// scale_gelu with gn_finalize's work folded into its prologue (one launch less on every layer's critical path): each workgroup
// re-derives 1/(std + eps) of its channel's group from the S partial sums - the same reduction order as gn_finalize_kernel, so
// the scale is bit-identical - and the first workgroup of a channel also stores scale[b][c] (and the group's statistics) for the
// VJP.  No atomics, no fences: the partial sums come from the previous launch on the stream.
__global__ __launch_bounds__(256) void scale_gelu_fin_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                             const float* __restrict__ gamma, const float* __restrict__ film,
                                                             long film_bs, float* __restrict__ stats, float* __restrict__ scale,
                                                             float* __restrict__ a, int C, int G, long hw, long n, int S,
                                                             float eps) {
    __shared__ float sc_sh;
    __shared__ double red_sh[512];
    const int c = blockIdx.y, b = blockIdx.z;
    const int cg = C / G, g = c / cg;
    // All 256 threads fetch the S partial sums (a conv epilogue's fused sums come in hundreds to thousands of slots); thread q < 32 then adds the sums of threads q, q + 32, ...
    // in that order - for S <= 256 exactly the additions, in the order, of the 32-thread loop this replaces (and of gn_finalize).
    {
        double t0 = 0, t1 = 0;
        for (int q = threadIdx.x; q < S; q += 256) {
            t0 += part[((long)(b * G + g) * S + q) * 2];
            t1 += part[((long)(b * G + g) * S + q) * 2 + 1];
        }
        red_sh[threadIdx.x * 2] = t0;
        red_sh[threadIdx.x * 2 + 1] = t1;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            t0 += red_sh[(threadIdx.x + 32 * j) * 2];
            t1 += red_sh[(threadIdx.x + 32 * j) * 2 + 1];
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            t0 += __shfl_xor(t0, o, 32);
            t1 += __shfl_xor(t1, o, 32);
        }
        if (threadIdx.x == 0) {
            const double mean = t0 / (double)n;
            double var = (t1 - (double)n * mean * mean) / (double)(n - 1);
            if (var < 0) var = 0;
            const float sd = (float)sqrt(var);
            const float r = 1.f / (sd + eps);
            const float scv = gamma[c] * (film[(long)b * film_bs + c] + 1.f) * r;
            sc_sh = scv;
            if (blockIdx.x == 0) {
                scale[b * C + c] = scv;
                if (c == g * cg) {
                    stats[(b * G + g) * 3 + 0] = (float)mean;
                    stats[(b * G + g) * 3 + 1] = sd;
                    stats[(b * G + g) * 3 + 2] = r;
                }
            }
        }
    }
    __syncthreads();
    const float sc = sc_sh;
    const long base = ((long)b * C + c) * hw;
    const long nv = hw / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    float4* a4 = reinterpret_cast<float4*>(a + base);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        float4 v = x4[i];
        v.x = gelu_f(v.x * sc);
        v.y = gelu_f(v.y * sc);
        v.z = gelu_f(v.z * sc);
        v.w = gelu_f(v.w * sc);
        a4[i] = v;
    }
    if (blockIdx.x == 0)
        for (long i = nv * 4 + threadIdx.x; i < hw; i += blockDim.x) a[base + i] = gelu_f(x[base + i] * sc);
}

// scale*GELU written as bf16 "units" for the pipelined bf16 conv (conv_bf16p.hip, UNITS variant): a unit = 8 consecutive
// channels of one (f, t) as 8 bf16 = 16 bytes - exactly one lane's MFMA B-operand fragment.  Layout
// [B][C/8][F][4 planes][T/4 + 1] units, plane p entry j = time step 4j + p - 1 (so a conv tile's operand runs S(m) are
// contiguous in HBM and travel by LDS-DMA unchanged); t = -1 and t >= T are stored as zeros (the conv's time padding).
// One thread = 8 channels x 4 time steps: eight 16-byte loads, four 16-byte stores, all coalesced along t.
// grid: (blocks over F * T/4, C/8, B)
typedef __bf16 bf16x8_n __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void scale_gelu_units_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               bf16x8_n* __restrict__ au, int C, int F, int T) {
    const int g = blockIdx.y, b = blockIdx.z;
    const int Q = T >> 2, PI = Q + 1;
    const long hw = (long)F * T;
    float sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = scale[b * C + 8 * g + j];
    const float* xb = x + ((long)b * C + 8 * g) * hw;
    bf16x8_n* ab = au + ((long)b * (C >> 3) + g) * F * 4 * PI;
    const long nq = (long)F * Q;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i / Q), q = (int)(i - (long)f * Q);
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(xb + j * hw + (long)f * T + 4 * q);
        bf16x8_n u0, u1, u2, u3, z;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u0[j] = (__bf16)gelu_f(v[j].x * sc[j]);
            u1[j] = (__bf16)gelu_f(v[j].y * sc[j]);
            u2[j] = (__bf16)gelu_f(v[j].z * sc[j]);
            u3[j] = (__bf16)gelu_f(v[j].w * sc[j]);
            z[j] = (__bf16)0.f;
        }
        bf16x8_n* row = ab + (long)f * 4 * PI;
        row[1 * PI + q] = u0;          // t = 4q     -> plane 1, entry q
        row[2 * PI + q] = u1;          // t = 4q + 1 -> plane 2
        row[3 * PI + q] = u2;          // t = 4q + 2 -> plane 3
        row[q + 1] = u3;               // t = 4q + 3 -> plane 0, entry q + 1
        if (q == 0) row[0] = z;        // t = -1
        if (q == Q - 1) {              // t = T, T + 1, T + 2
            row[1 * PI + Q] = z;
            row[2 * PI + Q] = z;
            row[3 * PI + Q] = z;
        }
    }
}

// grid: (S, B*G).  A group is cg channels of hw elements, contiguous: n = cg*hw.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dadu,
                                                             const float* __restrict__ scale,
                                                             double* __restrict__ part, int C, int G, long hw, int S) {
    __shared__ double sh[8];
    const int s = blockIdx.x;
    const int bg = blockIdx.y;
    const int b = bg / G, g = bg % G;
    const int cg = C / G;
    const long n = (long)cg * hw;
    const long chunk = ((n / 4 + S - 1) / S) * 4;
    const long beg = (long)s * chunk;
    long end = beg + chunk;
    if (end > n) end = n;
    const long base = ((long)b * C + (long)g * cg) * hw;
    double s0 = 0, s1 = 0;
    // hw is a multiple of 4 (checked on the host) so a float4 never straddles two channels.  Channel by channel: the
    // scale is a scalar of the inner loop and there is no 64-bit division per element (there was: i / hw).
    if (beg < end) {
        const int c_lo = (int)(beg / hw), c_hi = (int)((end - 1) / hw);
        for (int cl = c_lo; cl <= c_hi; ++cl) {
            const long lo = beg > (long)cl * hw ? beg : (long)cl * hw;
            const long hi = end < (long)(cl + 1) * hw ? end : (long)(cl + 1) * hw;
            const float sc = scale[b * C + g * cg + cl];
            double sc_sum = 0;
            auto term = [&](const float4 xv, float4 dv) {
                dv.x *= gelu_grad_f(xv.x * sc);
                dv.y *= gelu_grad_f(xv.y * sc);
                dv.z *= gelu_grad_f(xv.z * sc);
                dv.w *= gelu_grad_f(xv.w * sc);
                return (double)dv.x * xv.x + (double)dv.y * xv.y + (double)dv.z * xv.z + (double)dv.w * xv.w;
            };
            const long st = (long)blockDim.x * 4;
            long i = lo + (long)threadIdx.x * 4;
            for (; i + st < hi; i += 2 * st) {           // two iterations' loads (4 x 16 bytes) in flight per thread
                const float4 x0 = *reinterpret_cast<const float4*>(x + base + i);
                const float4 d0 = *reinterpret_cast<const float4*>(dadu + base + i);
                const float4 x1 = *reinterpret_cast<const float4*>(x + base + i + st);
                const float4 d1 = *reinterpret_cast<const float4*>(dadu + base + i + st);
                sc_sum += term(x0, d0);
                sc_sum += term(x1, d1);
            }
            for (; i < hi; i += st)
                sc_sum += term(*reinterpret_cast<const float4*>(x + base + i), *reinterpret_cast<const float4*>(dadu + base + i));
            s0 += (double)sc * sc_sum;
        }
    }
    block_reduce2(s0, s1, sh);
    if (threadIdx.x == 0) part[(long)bg * S + s] = s0;
}

// grid: (blocks, C, B)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                           const float* __restrict__ gy,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ stats,
                                                           const double* __restrict__ part, float* __restrict__ gx,
                                                           float rbeta, int C, int G, long hw, int S, float eps,
                                                           const float* __restrict__ acc, float ca, float cb) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int cg = C / G;
    const int g = c / cg;
    const long n = (long)cg * hw;
    // the S partial sums: read in parallel by the first wave (a serial loop of S loads delayed every block's start)
    __shared__ double S1_sh;
    if (threadIdx.x < 64) {
        double v = 0;
        for (int s = threadIdx.x; s < S; s += 64) v += part[(long)(b * G + g) * S + s];
        v = wave_sum(v);
        if (threadIdx.x == 0) S1_sh = v;
    }
    __syncthreads();
    const double S1 = S1_sh;
    const float mean = stats[(b * G + g) * 3 + 0];
    const float sd = stats[(b * G + g) * 3 + 1];
    // S1 was accumulated with scale = k/(sd+eps); the formula needs sum(k*du*x) = S1*(sd+eps)
    const double se = (double)sd + eps;
    const float coef = (sd > 0.f) ? (float)(S1 * se / ((double)(n - 1) * sd * se * se)) : 0.f;
    const float sc = scale[b * C + c];
    const long base = ((long)b * C + c) * hw;
    const long nv = hw / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* d4 = reinterpret_cast<const float4*>(da + base);
    const float4* g4 = gy ? reinterpret_cast<const float4*>(gy + base) : nullptr;
    float4* o4 = reinterpret_cast<float4*>(gx + base);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        const float4 xv = x4[i];
        float4 dv = d4[i];
        dv.x *= gelu_grad_f(xv.x * sc);          // du, recomputed instead of round-tripping it through HBM
        dv.y *= gelu_grad_f(xv.y * sc);
        dv.z *= gelu_grad_f(xv.z * sc);
        dv.w *= gelu_grad_f(xv.w * sc);
        float4 o;
        o.x = sc * dv.x - (xv.x - mean) * coef;
        o.y = sc * dv.y - (xv.y - mean) * coef;
        o.z = sc * dv.z - (xv.z - mean) * coef;
        o.w = sc * dv.w - (xv.w - mean) * coef;
        if (g4) {
            const float4 gv = g4[i];
            o.x += rbeta * gv.x;
            o.y += rbeta * gv.y;
            o.z += rbeta * gv.z;
            o.w += rbeta * gv.w;
        }
        if (acc) {
            // merged block tail (babe_gn_bwd_apply_merge): out = ca*acc + cb*gx instead of storing gx and running axpby2 over it;
            // rounded as the two-pass form: fl(fl(ca acc) + fl(cb gx))
#pragma clang fp contract(off)
            const float4 av = reinterpret_cast<const float4*>(acc + base)[i];
            const float px = ca * av.x, py = ca * av.y, pz = ca * av.z, pw = ca * av.w;
            const float qx = cb * o.x, qy = cb * o.y, qz = cb * o.z, qw = cb * o.w;
            o.x = px + qx;
            o.y = py + qy;
            o.z = pz + qz;
            o.w = pw + qw;
        }
        o4[i] = o;
    }
}

}  // namespace

// (timing probe only, results wrong: BABE_ABL_GN bit 1 skips the forward statistics pass, bit 2 the VJP's partial-sum pass - the upper
// bound of what producing those sums inside the convolutions' epilogues could save)
static int abl_gn() {
    static const int v = [] { const char* e = getenv("BABE_ABL_GN"); return e ? atoi(e) : 0; }();
    return v;
}

extern "C" int babe_gn_partial(const float* x, double* part, int B, int G, long n, int S, void* stream) {
    BABE_CHECK_ARG(x && part && B > 0 && G > 0 && n > 1 && S > 0, "gn_partial: bad arguments");
    if (abl_gn() & 1) return BABE_OK;
    BABE_CHECK_ARG(n % 4 == 0, "gn_partial: group size %ld not a multiple of 4", n);
    BabeProfScope prof(BABE_SLOT_GN_STATS, 4.0 * B * G * (double)n, 0, 0, stream);
    hipLaunchKernelGGL(gn_partial_kernel<false>, dim3(S, B * G), dim3(256), 0, (hipStream_t)stream, x, part, n, S,
                       (int*)nullptr, (const float*)nullptr, (const float*)nullptr, 0L, (float*)nullptr, (float*)nullptr, 0, 1, 0.f);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

/* babe_gn_partial + babe_gn_finalize in ONE launch (same results bit for bit).  ticket: B*G ints, zero when first used and left
 * zero by every call; calls that may run concurrently (different streams) need different ticket buffers. */
extern "C" int babe_gn_stats(const float* x, double* part, int* ticket, const float* gamma, const float* film, long film_bs,
                             float* stats, float* scale, int B, int C, int G, long n, int S, float eps, void* stream) {
    BABE_CHECK_ARG(x && part && ticket && gamma && film && stats && scale && B > 0 && G > 0 && n > 1 && S > 0, "gn_stats: bad arguments");
    BABE_CHECK_ARG(n % 4 == 0 && G <= 64 && C % G == 0, "gn_stats: n=%ld C=%d G=%d unsupported", n, C, G);
    BabeProfScope prof(BABE_SLOT_GN_STATS, 4.0 * B * G * (double)n, 0, 0, stream);
    hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(S, B * G), dim3(256), 0, (hipStream_t)stream, x, part, n, S, ticket, gamma,
                       film, film_bs, stats, scale, C, G, eps);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_gn_finalize(const double* part, const float* gamma, const float* film, long film_bs, float* stats,
                                float* scale, int B, int C, int G, long n, int S, float eps, void* stream) {
    BABE_CHECK_ARG(part && gamma && film && stats && scale, "gn_finalize: null pointer");
    BABE_CHECK_ARG(G <= 64 && C % G == 0, "gn_finalize: C=%d G=%d unsupported", C, G);
    BabeProfScope prof(BABE_SLOT_GN_STATS, 0, 0, 0, stream);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, part, gamma, film, film_bs,
                       stats, scale, C, G, n, S, eps);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_scale_gelu(const float* x, const float* scale, float* a, int B, int C, long hw, void* stream) {
    BABE_CHECK_ARG(x && scale && a && B > 0 && C > 0 && hw > 0, "scale_gelu: bad arguments");
    BABE_CHECK_ARG(hw % 4 == 0, "scale_gelu: plane size %ld not a multiple of 4", hw);
    BabeProfScope prof(BABE_SLOT_SCALE_GELU, 8.0 * B * C * (double)hw, 0, 0, stream);
    int bx = cdiv(hw / 4, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(scale_gelu_kernel, dim3(bx, C, B), dim3(256), 0, (hipStream_t)stream, x, scale, a, C, hw);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

/* babe_gn_finalize + babe_scale_gelu in one launch: part from babe_gn_partial (same B, G, n, S); writes a = gelu(x * scale) and,
 * for the VJP, stats [B][G][3] and scale [B][C] - bit-identical to the two separate calls. */
extern "C" int babe_scale_gelu_fin(const float* x, const double* part, const float* gamma, const float* film, long film_bs,
                                   float* stats, float* scale, float* a, int B, int C, int G, long hw, int S, float eps,
                                   void* stream) {
    BABE_CHECK_ARG(x && part && gamma && film && stats && scale && a && B > 0 && C > 0 && hw > 0 && S > 0, "scale_gelu_fin: bad arguments");
    BABE_CHECK_ARG(hw % 4 == 0 && G <= 64 && C % G == 0, "scale_gelu_fin: hw=%ld C=%d G=%d unsupported", hw, C, G);
    BabeProfScope prof(BABE_SLOT_SCALE_GELU, 8.0 * B * C * (double)hw, 0, 0, stream);
    int bx = cdiv(hw / 4, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    const long n = (long)(C / G) * hw;
    hipLaunchKernelGGL(scale_gelu_fin_kernel, dim3(bx, C, B), dim3(256), 0, (hipStream_t)stream, x, part, gamma, film, film_bs,
                       stats, scale, a, C, G, hw, n, S, eps);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" long babe_units_size(int C, int F, int T) { return (long)(C / 8) * F * 4 * (T / 4 + 1); }     // 16-byte units per batch item

extern "C" int babe_scale_gelu_units(const float* x, const float* scale, void* au, int B, int C, int F, int T,
                                     void* stream) {
    BABE_CHECK_ARG(x && scale && au && B > 0 && C > 0 && F > 0 && T > 0, "scale_gelu_units: bad arguments");
    BABE_CHECK_ARG(T % 4 == 0 && C % 8 == 0, "scale_gelu_units: T=%d C=%d unsupported (need T %% 4 == 0, C %% 8 == 0)", T, C);
    BABE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)au & 15) == 0, "scale_gelu_units: unaligned pointer");
    BabeProfScope prof(BABE_SLOT_SCALE_GELU, 6.0 * B * C * (double)F * T, 0, 0, stream);
    int bx = cdiv((long)F * (T / 4), 256 * 2);
    if (bx < 1) bx = 1;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(scale_gelu_units_kernel, dim3(bx, C / 8, B), dim3(256), 0, (hipStream_t)stream, x, scale,
                       (bf16x8_n*)au, C, F, T);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_gn_bwd_partial(const float* x, const float* da, const float* scale, double* part, int B, int C,
                                   int G, long hw, int S, void* stream) {
    BABE_CHECK_ARG(x && da && scale && part, "gn_bwd_partial: null pointer");
    BABE_CHECK_ARG(hw % 4 == 0 && C % G == 0, "gn_bwd_partial: hw=%ld C=%d G=%d unsupported", hw, C, G);
    if (abl_gn() & 2) return BABE_OK;
    BabeProfScope prof(BABE_SLOT_GN_BWD_PARTIAL, 8.0 * B * C * (double)hw, 0, 0, stream);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(S, B * G), dim3(256), 0, (hipStream_t)stream, x, da, scale,
                       part, C, G, hw, S);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

static int gn_bwd_apply_launch(const float* acc, float ca, float cb, const float* x, const float* da, const float* gy, const float* scale,
                                 const float* stats, const double* part, float* gx, float rbeta, int B, int C, int G,
                                 long hw, int S, float eps, void* stream) {
    BABE_CHECK_ARG(x && da && scale && stats && part && gx, "gn_bwd_apply: null pointer");
    BABE_CHECK_ARG(hw % 4 == 0 && C % G == 0, "gn_bwd_apply: hw=%ld C=%d G=%d unsupported", hw, C, G);
    BabeProfScope prof(BABE_SLOT_GN_BWD_APPLY, ((gy ? 16.0 : 12.0) + (acc ? 4.0 : 0.0)) * B * C * (double)hw, 0, 0, stream);
    int bx = cdiv(hw / 4, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(bx, C, B), dim3(256), 0, (hipStream_t)stream, x, da, gy, scale, stats,
                       part, gx, rbeta, C, G, hw, S, eps, acc, ca, cb);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_gn_bwd_apply(const float* x, const float* da, const float* gy, const float* scale,
                                 const float* stats, const double* part, float* gx, float rbeta, int B, int C, int G,
                                 long hw, int S, float eps, void* stream) {
    return gn_bwd_apply_launch(nullptr, 0.f, 0.f, x, da, gy, scale, stats, part, gx, rbeta, B, C, G, hw, S, eps, stream);
}

/* the same pass with the block's tail merged in: out = ca*acc + cb*(the gx babe_gn_bwd_apply would store); acc: dense [B][C][hw] */
extern "C" int babe_gn_bwd_apply_merge(const float* x, const float* da, const float* gy, const float* scale,
                                 const float* stats, const double* part, float* gx, float rbeta, int B, int C, int G,
                                 long hw, int S, float eps, void* stream, const float* acc, float ca, float cb) {
    BABE_CHECK_ARG(acc && ((uintptr_t)acc & 15) == 0, "gn_bwd_apply_merge: acc must be a 16-byte aligned dense tensor");
    return gn_bwd_apply_launch(acc, ca, cb, x, da, gy, scale, stats, part, gx, rbeta, B, C, G, hw, S, eps, stream);
}

