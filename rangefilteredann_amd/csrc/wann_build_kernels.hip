// wann_build_kernels.hip -- the GPU Vamana build kernels for float32 rows (+ the dispatchers): one translation unit per element
// type of the point set, see wann_build_kernels_body.inc.
#define WANN_DT 0
#include "wann_build_kernels_body.inc"
