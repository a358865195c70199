// compat/include/common.h -- compile-time configuration of the drop-in, same macro names as the
// reference's server/include/common.h:4-18 so that a server built against this tree is configured
// the same way (edit + rebuild).  The values are also overridable at run time through the
// environment (MI355_NOISE_FILTER=0|1, MI355_VISUALIZER=0..5) without a rebuild.
#ifndef MI355_COMPAT_COMMON_H_
#define MI355_COMPAT_COMMON_H_

// 3x3 noise filter before the diff (reference: commented out by default)
// #define NOISE_FILTER
#define K 3

// Noise visualizer: 1 heat map, 2 red-black, 3 red-black overlap, 4 grayscale, 5 binarization
// #define NOISE_VISUALIZER 2

#define CHARS_STR "0123456789BFPSWbkps :/"
#define LR_THRESHOLDS 20
#define GPU

#endif
