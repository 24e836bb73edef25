// wann_hip_util.h -- small host-side HIP helpers shared by wann_host.cpp and wann_gpu_build.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

namespace wann {

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
#define HIP_CHECK(expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
  } while (0)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  void ensure(size_t n) {
    if (n <= cap) return;
    release();
    HIP_CHECK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)));
    cap = n;
  }
  void upload(const std::vector<T> &v) {
    ensure(v.size());
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  size_t bytes() const { return cap * sizeof(T); }
};


}  // namespace wann
