// wann_build_kernels_i8.hip -- the GPU Vamana build kernels for int8 rows: one translation unit per element
// type of the point set, see wann_build_kernels_body.inc.
#define WANN_DT 2
#include "wann_build_kernels_body.inc"
