// oracle/ref_harness/layout_check.cpp -- compile-time check against the REFERENCE header (included
// from the reference tree, not copied): the drop-in's CUDACore must fit the storage the reference's
// server.cpp reserves on its stack (server/src/server.cpp:53), and matsz must have the same layout.
#include "server/include/kernels.cuh"

static_assert(sizeof(diff::cuda::CUDACore) == 160, "reference CUDACore size changed: update compat/include/kernels.cuh");
static_assert(sizeof(diff::utils::matsz) == 8, "reference matsz layout changed");
int main() { return 0; }
