// Instantiation sets of the conv engine's kernel templates (hificar_conv.hip.h).  This file is compiled once per set with
// -DHIFICAR_INST_SET=n (the Makefile builds the ten objects in parallel; one translation unit with all of them took four minutes):
//   0 / 1 / 2   conv_f32do_kernel, K chunks of 16 / 32 / 64 channels        (9 tile shapes each)
//   3 / 4 / 5   conv_bf16x3_kernel, the same
//   6           conv_bf16x3nb_kernel                                         (6 shapes)
//   7 / 8       conv_sk_f32_kernel / conv_sk_bf16x3_kernel                   (9 shapes each)
//   9           conv_pair_f32_kernel (3 shapes), conv_pair_bf16x3_kernel (2)
// Each set exports conv_inst_launch_<n> / conv_inst_attrs_<n> (hificar_launch.h); hificar.hip dispatches over them.
#include "hificar_conv.hip.h"
#include "hificar_launch.h"

#ifndef HIFICAR_INST_SET
#error "compile with -DHIFICAR_INST_SET=0..9"
#endif

namespace hificar {

// (MI, WM, WN) of the dense forms: four MFMA waves as 1 x 4, 2 x 2 or 4 x 1, each owning 4, 2 or 1 row blocks.
// (The body also supports 8 MFMA waves per workgroup — measured 3-5 % slower than the 4-wave shapes of the same tile in both arithmetics.)
#define HIFICAR_TILES9(X, nc) X(4, 1, 4, nc) X(4, 2, 2, nc) X(4, 4, 1, nc) X(2, 1, 4, nc) X(2, 2, 2, nc) X(2, 4, 1, nc) X(1, 1, 4, nc) X(1, 2, 2, nc) X(1, 4, 1, nc)

#if HIFICAR_INST_SET == 0
#define FAMILY kConvF32do
#define KERNEL conv_f32do_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) HIFICAR_TILES9(X, 1)
#elif HIFICAR_INST_SET == 1
#define FAMILY kConvF32do
#define KERNEL conv_f32do_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) HIFICAR_TILES9(X, 2)
#elif HIFICAR_INST_SET == 2
#define FAMILY kConvF32do
#define KERNEL conv_f32do_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) HIFICAR_TILES9(X, 4)
#elif HIFICAR_INST_SET == 3
#define FAMILY kConvBf16x3
#define KERNEL conv_bf16x3_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) HIFICAR_TILES9(X, 1)
#elif HIFICAR_INST_SET == 4
#define FAMILY kConvBf16x3
#define KERNEL conv_bf16x3_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) HIFICAR_TILES9(X, 2)
#elif HIFICAR_INST_SET == 5
#define FAMILY kConvBf16x3
#define KERNEL conv_bf16x3_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) HIFICAR_TILES9(X, 4)
#elif HIFICAR_INST_SET == 6
#define FAMILY kConvBf16x3nb
#define KERNEL conv_bf16x3nb_kernel
#define PARAMS MultiConvParams
#define SHAPES(X) X(2, 2, 2, 2) X(2, 4, 1, 2) X(2, 1, 4, 2) X(2, 2, 2, 4) X(2, 4, 1, 4) X(2, 1, 4, 4)
#elif HIFICAR_INST_SET == 7 || HIFICAR_INST_SET == 8
// split-K: conv_sk_*_kernel<MI, NC16>, a 512-thread workgroup whatever MI (wm = wn = 1 in the ConvShape)
#if HIFICAR_INST_SET == 7
#define FAMILY kConvSkF32
#define KERNEL_PTR(mi, wm, wn, nc) (conv_sk_f32_kernel<mi, nc>)
#else
#define FAMILY kConvSkBf16x3
#define KERNEL_PTR(mi, wm, wn, nc) (conv_sk_bf16x3_kernel<mi, nc>)
#endif
#define THREADS(wm, wn) 512
#define PARAMS MultiConvParams
#define SHAPES(X) X(1, 1, 1, 1) X(2, 1, 1, 1) X(4, 1, 1, 1) X(1, 1, 1, 2) X(2, 1, 1, 2) X(4, 1, 1, 2) X(1, 1, 1, 4) X(2, 1, 1, 4) X(4, 1, 1, 4)
#elif HIFICAR_INST_SET == 9
#define PARAMS PairParams
#define THREADS(wm, wn) 512
#else
#error "HIFICAR_INST_SET out of range"
#endif

#ifndef KERNEL_PTR
#define KERNEL_PTR(mi, wm, wn, nc) (KERNEL<mi, wm, wn, nc>)
#endif
#ifndef THREADS
#define THREADS(wm, wn) ((wm * wn + 4) * 64)
#endif

#define HIFICAR_CAT_(a, b) a##b
#define HIFICAR_CAT(a, b) HIFICAR_CAT_(a, b)

#if HIFICAR_INST_SET != 9
hipError_t HIFICAR_CAT(conv_inst_launch_, HIFICAR_INST_SET)(const ConvShape& s, const void* params, dim3 grid, size_t lds, hipStream_t stream, bool* handled) {
    *handled = false;
    if (s.family != FAMILY) return hipSuccess;
#define X(A_, B_, C_, D_)                                                                                                      \
    if (s.mi == A_ && s.wm == B_ && s.wn == C_ && s.nc16 == D_) {                                                              \
        *handled = true;                                                                                                       \
        hipLaunchKernelGGL(KERNEL_PTR(A_, B_, C_, D_), grid, dim3(THREADS(B_, C_)), lds, stream, *static_cast<const PARAMS*>(params)); \
        return hipGetLastError();                                                                                              \
    }
    SHAPES(X)
#undef X
    return hipSuccess;
}

hipError_t HIFICAR_CAT(conv_inst_attrs_, HIFICAR_INST_SET)() {
    hipError_t e = hipSuccess;
#define X(A_, B_, C_, D_)                                                                                                                        \
    if (e == hipSuccess)                                                                                                                         \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL_PTR(A_, B_, C_, D_)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SHAPES(X)
#undef X
    return e;
}
#else
// the fused pairs: C = 64 on a 128-row tile (2 x 2 waves), C = 32 on 512 rows (4 x 1) or — mid-size launches, exact fp32 — 128 rows
#define PAIR_F32(X) X(4, 2, 2, 4) X(4, 4, 1, 2) X(1, 4, 1, 2)
#define PAIR_BF(X) X(4, 2, 2, 4) X(4, 4, 1, 2)
hipError_t conv_inst_launch_9(const ConvShape& s, const void* params, dim3 grid, size_t lds, hipStream_t stream, bool* handled) {
    *handled = false;
    if (s.family != kPairF32 && s.family != kPairBf16x3) return hipSuccess;
#define X(A_, B_, C_, D_)                                                                                                  \
    if (s.family == kPairF32 && s.mi == A_ && s.wm == B_ && s.wn == C_ && s.nc16 == D_) {                                  \
        *handled = true;                                                                                                   \
        hipLaunchKernelGGL((conv_pair_f32_kernel<A_, B_, C_, D_>), grid, dim3(512), lds, stream, *static_cast<const PairParams*>(params)); \
        return hipGetLastError();                                                                                          \
    }
    PAIR_F32(X)
#undef X
#define X(A_, B_, C_, D_)                                                                                                     \
    if (s.family == kPairBf16x3 && s.mi == A_ && s.wm == B_ && s.wn == C_ && s.nc16 == D_) {                                  \
        *handled = true;                                                                                                      \
        hipLaunchKernelGGL((conv_pair_bf16x3_kernel<A_, B_, C_, D_>), grid, dim3(512), lds, stream, *static_cast<const PairParams*>(params)); \
        return hipGetLastError();                                                                                             \
    }
    PAIR_BF(X)
#undef X
    return hipSuccess;
}

hipError_t conv_inst_attrs_9() {
    hipError_t e = hipSuccess;
#define X(A_, B_, C_, D_) \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_f32_kernel<A_, B_, C_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PAIR_F32(X)
#undef X
#define X(A_, B_, C_, D_) \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_bf16x3_kernel<A_, B_, C_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PAIR_BF(X)
#undef X
    return e;
}
#endif

}  // namespace hificar
