// wann_kernels_i8.hip -- the search / scan kernels of the window-filtered ANN engine for int8 rows:
// one translation unit per element type of the point set (python_bindings.cpp:232-237), see wann_kernels_body.inc.
#define WANN_DT 2
#include "wann_kernels_body.inc"
