// wann_kernels.hip -- the search / scan kernels of the window-filtered ANN engine for float32 rows (+ the type-independent kernels and the dispatchers):
// one translation unit per element type of the point set (python_bindings.cpp:232-237), see wann_kernels_body.inc.
#define WANN_DT 0
#include "wann_kernels_body.inc"
