/* LD_PRELOAD shim for tests/golden/make_vamana_golden.py ONLY: free() does nothing.
 * The reference's VamanaIndex constructor copy-assigns a PointRange that owns a raw buffer
 * (ParlayANN/python/vamana_index.cpp:47-48, point_range.h:113-115): the temporary's destructor frees the buffer
 * the index keeps using.  With free() disabled the index reads the data it was meant to read, so the golden vectors
 * are the reference's INTENDED outputs (they then agree with the oracle's restatement on every case). */
#include <stddef.h>
void free(void *p) { (void)p; }
