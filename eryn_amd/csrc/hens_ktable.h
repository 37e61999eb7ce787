// hens_ktable.h - where the stepping kernels are instantiated (round 4: the library builds from several translation units).
//
// hens.hip (the C ABI and the host logic) no longer instantiates k_stretch_fast / k_stretch / k_split1_pt / k_iter itself: every
// likelihood kind has a translation unit of its own (hens_k_dense.hip, hens_k_diag.hip, hens_k_rosen.hip, from hens_ktable.inc)
// that instantiates its kernels and hands out their host function pointers; the host launches through hipLaunchKernel /
// hipExtLaunchKernel or, on the context's AQL queue, by the name HIP registered for the pointer.  Nothing about a kernel changes;
// the units compile in parallel (eryn_amd/_build.py: 2 m 50 s -> about a minute).
#pragma once
namespace hens {
// nullptr: no such instantiation.  `mode` MODE_STRETCH / MODE_EVAL / MODE_MH; D the compile-time row width.
#define HENS_KTABLE_DECL(KIND)                                                                                                \
    const void* ktab_stretch_fast_##KIND(int mode, int D, bool pipe, bool per);                                               \
    const void* ktab_stretch_##KIND(int mode);                                                                                \
    const void* ktab_stretch2_##KIND(int D, bool pipe);                                                                                  \
    const void* ktab_split1_pt_##KIND(int D, bool per, bool shrt, bool pipe, bool col);                                       \
    const void* ktab_iter_##KIND(int D, bool per);
HENS_KTABLE_DECL(dense)
HENS_KTABLE_DECL(diag)
HENS_KTABLE_DECL(rosen)
#undef HENS_KTABLE_DECL
}  // namespace hens
