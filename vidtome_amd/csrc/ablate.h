// Ablation switches of the hand-scheduled kernels (filter_kernel in match_filter.hip, the projection GEMMs in linear.hip),
// in ONE place.  They exist for the "where does the time go" measurements quoted in profiles/HISTORY.md section 9: each one removes a
// part of a loop (loads, barriers, stores ...) and EVERY ONE OF THEM PRODUCES WRONG RESULTS.  The shipped library is built
// with none of them defined: every macro below then expands to its "shipped" argument and nothing else, the kernels'
// sources carry no #ifdef, and vtm_build_ablations() (api.hip) returns 0 -- tests/test_host.py checks exactly that on
// the library the tests load.  An experiment build is `python -m vidtome_amd.build --tag NAME -DVTM_EXP_...` (it lands in
// lib/variants/NAME/, never in the shipped path).
//
//   VTM_EXP_NOWRAP     filter: skip the candidate collection at the end of every dst tile
//   VTM_EXP_NOBARRIER  filter: drop the end-of-step wait + barrier
//   VTM_EXP_NOAWAIT    filter: never wait for fragment loads
//   VTM_EXP_NODMA      filter: do not fetch dst tiles
//   VTM_EXP_HOTMEM     filter (phased loop): fetch everything from one hot tile
//   VTM_EXP_NOBLOAD    filter: do not fetch src fragments
//   VTM_EXP_NOLDSREAD  filter: do not read dst fragments from LDS
//   VTM_LIN_NOA / VTM_LIN_NOW / VTM_LIN_NOSTORE / VTM_LIN_NOLDS   projection GEMMs: no token loads / no weight-tile
//                      loads / no output stores / no LDS fragment reads
#pragma once

#define VTM_ABL_BIT_NOWRAP 0x001
#define VTM_ABL_BIT_NOBARRIER 0x002
#define VTM_ABL_BIT_NOAWAIT 0x004
#define VTM_ABL_BIT_NODMA 0x008
#define VTM_ABL_BIT_HOTMEM 0x010
#define VTM_ABL_BIT_NOBLOAD 0x020
#define VTM_ABL_BIT_NOLDSREAD 0x040
#define VTM_ABL_BIT_LIN_NOA 0x080
#define VTM_ABL_BIT_LIN_NOW 0x100
#define VTM_ABL_BIT_LIN_NOSTORE 0x200
#define VTM_ABL_BIT_LIN_NOLDS 0x400

// ABL_X(code...)     = the shipped code of part X (dropped by the switch)
// ABL_NO_X(code...)  = what the ablated build runs in its place (nothing in the shipped build)
#ifdef VTM_EXP_NOWRAP
#define ABL_WRAP_COND(cond) ((cond) && jt < 0)
#define VTM_ABL_NOWRAP VTM_ABL_BIT_NOWRAP
#else
#define ABL_WRAP_COND(cond) (cond)
#define VTM_ABL_NOWRAP 0
#endif

#ifdef VTM_EXP_NOBARRIER
#define ABL_BARRIER(...)
#define VTM_ABL_NOBARRIER VTM_ABL_BIT_NOBARRIER
#else
#define ABL_BARRIER(...) __VA_ARGS__
#define VTM_ABL_NOBARRIER 0
#endif

#ifdef VTM_EXP_NOAWAIT
#define ABL_AWAIT_COUNT(n) 63
#define VTM_ABL_NOAWAIT VTM_ABL_BIT_NOAWAIT
#else
#define ABL_AWAIT_COUNT(n) (n)
#define VTM_ABL_NOAWAIT 0
#endif

#ifdef VTM_EXP_NODMA
#define ABL_DMA(...)
#define VTM_ABL_NODMA VTM_ABL_BIT_NODMA
#else
#define ABL_DMA(...) __VA_ARGS__
#define VTM_ABL_NODMA 0
#endif

#ifdef VTM_EXP_HOTMEM
#define ABL_STREAMED(shipped, hot) (hot)
#define VTM_ABL_HOTMEM VTM_ABL_BIT_HOTMEM
#else
#define ABL_STREAMED(shipped, hot) (shipped)
#define VTM_ABL_HOTMEM 0
#endif

#ifdef VTM_EXP_NOBLOAD
#define ABL_BLOAD(...)
#define VTM_ABL_NOBLOAD VTM_ABL_BIT_NOBLOAD
#else
#define ABL_BLOAD(...) __VA_ARGS__
#define VTM_ABL_NOBLOAD 0
#endif

#ifdef VTM_EXP_NOLDSREAD
#define ABL_LDSREAD(...)
#define ABL_NO_LDSREAD(...) __VA_ARGS__
#define VTM_ABL_NOLDSREAD VTM_ABL_BIT_NOLDSREAD
#else
#define ABL_LDSREAD(...) __VA_ARGS__
#define ABL_NO_LDSREAD(...)
#define VTM_ABL_NOLDSREAD 0
#endif

#ifdef VTM_LIN_NOA
#define ABL_LIN_A(...)
#define ABL_NO_LIN_A(...) __VA_ARGS__
#define VTM_ABL_LIN_NOA VTM_ABL_BIT_LIN_NOA
#else
#define ABL_LIN_A(...) __VA_ARGS__
#define ABL_NO_LIN_A(...)
#define VTM_ABL_LIN_NOA 0
#endif

#ifdef VTM_LIN_NOW
#define ABL_LIN_W(...)
#define ABL_NO_LIN_W(...) __VA_ARGS__
#define VTM_ABL_LIN_NOW VTM_ABL_BIT_LIN_NOW
#else
#define ABL_LIN_W(...) __VA_ARGS__
#define ABL_NO_LIN_W(...)
#define VTM_ABL_LIN_NOW 0
#endif

#ifdef VTM_LIN_NOSTORE
#define ABL_NO_LIN_STORE(...) __VA_ARGS__
#define VTM_ABL_LIN_NOSTORE VTM_ABL_BIT_LIN_NOSTORE
#else
#define ABL_NO_LIN_STORE(...)
#define VTM_ABL_LIN_NOSTORE 0
#endif

#ifdef VTM_LIN_NOLDS
#define ABL_LIN_LDS(...)
#define ABL_NO_LIN_LDS(...) __VA_ARGS__
#define VTM_ABL_LIN_NOLDS VTM_ABL_BIT_LIN_NOLDS
#else
#define ABL_LIN_LDS(...) __VA_ARGS__
#define ABL_NO_LIN_LDS(...)
#define VTM_ABL_LIN_NOLDS 0
#endif

// which switches THIS translation unit was compiled with (0 in the shipped build)
#define VTM_ABLATIONS                                                                                                    \
    (VTM_ABL_NOWRAP | VTM_ABL_NOBARRIER | VTM_ABL_NOAWAIT | VTM_ABL_NODMA | VTM_ABL_HOTMEM | VTM_ABL_NOBLOAD |            \
     VTM_ABL_NOLDSREAD | VTM_ABL_LIN_NOA | VTM_ABL_LIN_NOW | VTM_ABL_LIN_NOSTORE | VTM_ABL_LIN_NOLDS)
