// Measurement hook shared by every C-ABI entry point (bench.py's roofline / hbm blocks): when enabled, each launch is
// bracketed by HIP events on ITS stream and tallied in a slot together with its algorithmic bytes / flops and the flops
// the matrix pipe actually executes.  Dispatch counters (which kernel a conv call really took) are always on.
// Single host thread (the Python host is single-threaded); not part of the data path.
#pragma once
#include "../../include/babe_hip.h"

enum BabeProfSlot {
    BABE_SLOT_CONV53_WINO4 = 0,   // (5,3) conv, Winograd F(4,3) kernel
    BABE_SLOT_CONV53_WINO2,       // (5,3) conv, Winograd F(2,3) kernel
    BABE_SLOT_CONV53_DIRECT,      // (5,3) conv, direct implicit GEMM
    BABE_SLOT_CONV11,             // (1,1) conv
    BABE_SLOT_CONV_BF16,          // any conv on the bf16 MFMA kernels
    BABE_SLOT_DFT_STAGE,          // dense DFT stages of the length-L real FFT (1x1 convs with DFT matrices)
    BABE_SLOT_GN_STATS,           // gn_partial + gn_finalize
    BABE_SLOT_SCALE_GELU,
    BABE_SLOT_GN_BWD_PARTIAL,
    BABE_SLOT_GN_BWD_APPLY,
    BABE_SLOT_RESAMPLE,
    BABE_SLOT_AXPBY,
    BABE_SLOT_FILM,               // rff + linear
    BABE_SLOT_CQT_ANALYSIS,       // band_analysis_kernel
    BABE_SLOT_CQT_SYNTHESIS,      // band_synthesis_kernel
    BABE_SLOT_CQT_GATHER,         // gather / spec_scale / twiddle-transpose
    BABE_SLOT_STFT_FWD,
    BABE_SLOT_ISTFT,              // spec_filter_istft + ola + residual_seed
    BABE_SLOT_MAG_STATS,
    BABE_SLOT_FILTER_FIT,         // design_filter + filter_fit
    BABE_SLOT_SAMPLER,            // lincomb3 / sumsq / score_direction / mask_blend / fir_same
    BABE_SLOT_DENOISER,           // denoiser pre-pass kernels
    BABE_SLOT_CONV53_FEWCO,       // (5,3) conv with <= 4 output channels on the vector ALU (conv_fewco.hip)
    BABE_SLOT_CONV_BF16P,         // pipelined bf16 (5,3) conv (conv_bf16p.hip)
    BABE_SLOT_CONV53_WINO45,      // (5,3) conv, nested Winograd F(2,5) x F(4,3) (conv_wino45.hip)
    BABE_SLOT_CONV53_WINO85,      // (5,3) conv, nested Winograd F(4,5) x F(4,3) (conv_wino85.hip)
    BABE_NSLOTS
};

extern "C" void babe_prof_begin(int slot, double bytes, double flops, double exec_flops, void* stream);
extern "C" void babe_prof_end(void* stream);

struct BabeProfScope {
    void* s;
    BabeProfScope(int slot, double bytes, double flops, double exec_flops, void* stream) : s(stream) {
        babe_prof_begin(slot, bytes, flops, exec_flops, stream);
    }
    ~BabeProfScope() { babe_prof_end(s); }
};

// ALGORITHMIC work of one conv launch: direct-convolution flops with the unpadded channel counts; bytes = input + output
// (+ residual) + weights, each touched once.
static inline double babe_conv_flops(const babe_conv_args& a) {
    return 2.0 * a.B * (double)a.Cout * a.Cin * a.KH * a.KW * (double)a.F * a.T;
}
static inline double babe_conv_bytes(const babe_conv_args& a) {
    const double px = (double)a.B * a.F * a.T;
    return 4.0 * (px * (a.Cin + a.Cout + (a.res ? a.Cout : 0)) + (double)a.Cout * a.Cin * a.KH * a.KW);
}
