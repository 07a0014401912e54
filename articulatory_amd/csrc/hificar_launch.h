// Host-side entry to the conv engine's kernel instantiations.  The kernels are templates (hificar_conv.hip.h); every (family, tile shape) that is
// built lives in ONE of the instantiation sets of hificar_conv_inst.hip (compiled once per set, in parallel: the Makefile), and the host code in
// hificar.hip reaches them only through the two functions below — it never names a kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace hificar {

struct MultiConvParams;
struct PairParams;

enum ConvFamily {
    kConvF32do = 0,     // conv_f32do_kernel<MI, WM, WN, NC16>       dense exact-fp32, direct output
    kConvBf16x3 = 1,    // conv_bf16x3_kernel<MI, WM, WN, NC16>      dense bf16x3, LDS out-buffer
    kConvBf16x3nb = 2,  // conv_bf16x3nb_kernel<MI, WM, WN, NC16>    dense bf16x3, two channel blocks per MFMA wave
    kConvSkF32 = 3,     // conv_sk_f32_kernel<MI, NC16>              split-K exact fp32 (wm = wn = 1)
    kConvSkBf16x3 = 4,  // conv_sk_bf16x3_kernel<MI, NC16>
    kPairF32 = 5,       // conv_pair_f32_kernel<MI, WM, WN, NC16>    fused ResBlock layer pair (params: PairParams)
    kPairBf16x3 = 6,    // conv_pair_bf16x3_kernel<MI, WM, WN, NC16>
};

struct ConvShape {
    int family, mi, wm, wn, nc16;
};

// Launch the instantiation `s` (params: MultiConvParams, or PairParams for the pair families).  hipErrorInvalidValue: that shape is not built.
hipError_t conv_launch(const ConvShape& s, const void* params, dim3 grid, size_t lds_bytes, hipStream_t stream);
// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KiB on every instantiation (once per process and device is enough; cheap to repeat).
hipError_t conv_set_lds_attributes();

// one pair of these per instantiation set (hificar_conv_inst.hip, -DHIFICAR_INST_SET=n); `handled` false: the shape belongs to another set
#define HIFICAR_N_INST_SETS 10
#define HIFICAR_DECL_SET(n)                                                                                                              \
    hipError_t conv_inst_launch_##n(const ConvShape& s, const void* params, dim3 grid, size_t lds, hipStream_t stream, bool* handled); \
    hipError_t conv_inst_attrs_##n();
HIFICAR_DECL_SET(0) HIFICAR_DECL_SET(1) HIFICAR_DECL_SET(2) HIFICAR_DECL_SET(3) HIFICAR_DECL_SET(4)
HIFICAR_DECL_SET(5) HIFICAR_DECL_SET(6) HIFICAR_DECL_SET(7) HIFICAR_DECL_SET(8) HIFICAR_DECL_SET(9)
#undef HIFICAR_DECL_SET

}  // namespace hificar
