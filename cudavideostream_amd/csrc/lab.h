// lab.h -- the ONE place where laboratory builds differ from the product.
//
// The shipped library (make -C cudavideostream_amd/csrc) is built without MI355_LAB: the three constants below are 0
// and every `if (kAblate ...)` in the kernels folds away.  tools/ab_build.sh builds variants with
//   -DMI355_LAB=1 -DMI355_ABLATE=n   pack kernel: 1 = no log stores, 2 = also no meta stores, 3 = loads + a fold of the state only
//                 -DMI355_XABLATE=n  expander: 1 = prologue only, 2 = + code loads, 3 = + rounds, 9 = nothing but the dispatch
//                 -DMI355_PAD=n      n extra vector instructions per frame and tile of the pack kernel
//                 -DMI355_XWAVES=n   waves (= items) per workgroup of the expander
//                 -DMI355_XSTORE=n   expander: cache policy of the output stores -- 0 = both arrays non-temporal (the product), 1 = the
//                                    value array plain, 2 = the index array plain, 3 = both plain
// to price parts of the kernels (outputs of such builds are wrong by design; only their times matter).  Nothing else in
// the diff path is switchable at build time (filters.hip keeps one documented option, MI355_GRAY_FP64=0: the proven integer
// form of the weighted gray), and nothing but what include/mi355diff.h documents ("Options") at run time.
#ifndef MI355_LAB_H_
#define MI355_LAB_H_
namespace mi355 {
#if defined(MI355_LAB) && MI355_LAB
#ifndef MI355_ABLATE
#define MI355_ABLATE 0
#endif
#ifndef MI355_XABLATE
#define MI355_XABLATE 0
#endif
#ifndef MI355_PAD
#define MI355_PAD 0
#endif
#ifndef MI355_XWAVES
#define MI355_XWAVES 1
#endif
#ifndef MI355_XSTORE
#define MI355_XSTORE 0
#endif
constexpr int kAblate = MI355_ABLATE, kXAblate = MI355_XABLATE, kPad = MI355_PAD, kXWaves = MI355_XWAVES, kXStore = MI355_XSTORE;
#else
constexpr int kAblate = 0, kXAblate = 0, kPad = 0, kXWaves = 1, kXStore = 0;
#endif
}  // namespace mi355
#endif
