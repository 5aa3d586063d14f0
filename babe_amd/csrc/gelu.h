// GELU and its derivative as the GroupNorm / FiLM / GELU passes compute them (csrc/norm.hip) - shared with the conv epilogues that
// form the same sums (csrc/conv_wino85.hip, round 6), so that both round identically.
#pragma once
namespace babe_gelu {
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

// Standard normal CDF and density with ONE exponential: Phi(u) = 1 - P/2 (u >= 0), P/2 (u < 0) with
// P = (a1 t + ... + a5 t^5) exp(-u^2/2), t = 1/(1 + p |u|/sqrt2)  (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 on erf,
// i.e. 7.5e-8 on Phi: fp32 round-off level).
struct PhiPdf { float Phi, E; };
__device__ __forceinline__ PhiPdf phi_pdf(float u) {
    const float ax = fabsf(u) * kInvSqrt2;
    const float E = __expf(-0.5f * u * u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float P = fmaf(1.061405429f, t, -1.453152027f);
    P = fmaf(P, t, 1.421413741f);
    P = fmaf(P, t, -0.284496736f);
    P = fmaf(P, t, 0.254829592f);
    P = 0.5f * P * t * E;
    return {u >= 0.f ? 1.f - P : P, E};
}
__device__ __forceinline__ float gelu_f(float u) { return u * phi_pdf(u).Phi; }
__device__ __forceinline__ float gelu_grad_f(float u) {
    const PhiPdf r = phi_pdf(u);
    return fmaf(u * kInvSqrt2Pi, r.E, r.Phi);
}
}  // namespace babe_gelu
