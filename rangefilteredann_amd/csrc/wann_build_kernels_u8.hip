// wann_build_kernels_u8.hip -- the GPU Vamana build kernels for uint8 rows: one translation unit per element
// type of the point set, see wann_build_kernels_body.inc.
#define WANN_DT 1
#include "wann_build_kernels_body.inc"
