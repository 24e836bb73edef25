// wann_gpu_build.h -- host orchestration of the GPU Vamana build (kernels: wann_build_kernels.hip).
#pragma once
#include <stdint.h>

#include <vector>

#include "wann_build.h"
#include "wann_device.h"

namespace wann {

struct GpuBuildTarget {
  int32_t part_index;  // index into the device parts[] table
  HostPart *part;      // receives the finished graph (reference in-memory layout)
};

// Build the graphs of `targets` directly in the device adjacency pool `d_graph` (rows of the
// targets must be initialised to -1), then copy them back into target.part->g.
// view: device index view with points / parts / graph already resident.  Throws on HIP errors and
// std::runtime_error("gpu build overflow ...") when a visited list exceeds the LDS budget.
void gpu_build_graphs(const IndexView &view, int32_t *d_graph, const std::vector<PartDesc> &parts,
                      std::vector<GpuBuildTarget> &targets, int64_t R, int64_t L, double alpha, int num_cus,
                      int threads, void *stream, int vis_scale = 1);  // visited-list capacity = vis_scale x (2 L + 64)

}  // namespace wann
