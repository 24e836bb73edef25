// wann_stdsort.h -- the permutation libstdc++'s std::sort produces, restated so that it can run on the device.
//
// Why: the reference orders the candidates of robustPrune and a node's final neighbour list with std::sort on
// the DISTANCE ALONE (ParlayANN/algorithms/vamana/index.h:77-78, utils/graph.h:106).  std::sort is not stable, so
// where exactly equidistant candidates end up is a property of the algorithm -- deterministic for a given input
// sequence, but not expressible as a tie-breaking rule.  Integer-valued vectors (SIFT) are full of such ties, and
// which of two equidistant candidates is examined first decides what robustPrune keeps.  To write the graph files
// the reference writes, the GPU builder therefore runs the same algorithm on the same sequence:
//
//   introsort: while a range is longer than 16: depth limit 2 floor(log2 n) (heapsort when exhausted); median of
//   (first+1, middle, last-1) moved to first; unguarded Hoare partition of (first+1, last) around it; the right
//   part is sorted first (recursion), the left part by the loop; at the end one insertion sort over everything
//   (guarded for the first 16 elements, unguarded after).
//
// This is the algorithm GCC's <bits/stl_algo.h> documents and implements (std::__sort / __introsort_loop /
// __final_insertion_sort / the <bits/stl_heap.h> heap routines); it is written here from that description, with an
// explicit stack instead of recursion.  host_sanitize_test.cpp checks it against std::sort itself on tie-heavy
// sequences of every length up to a few thousand, including ones that exhaust the depth limit.
//
// Keys (K) are 64-bit unsigned (order-preserving distance bits << 32 | id << 1 | flag); LESS is the strict order used.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define WANN_HD __host__ __device__ __forceinline__
#else
#define WANN_HD inline
#endif

namespace wann {

struct DistOnlyLess {  // the reference's comparator: distance alone
  WANN_HD bool operator()(uint64_t a, uint64_t b) const { return (a >> 32) < (b >> 32); }
};
struct FullKeyLess {   // (distance, id): a total order on distinct candidates
  WANN_HD bool operator()(uint64_t a, uint64_t b) const { return (a | 1ull) < (b | 1ull); }
};

namespace stdsort_detail {

template <typename K, typename Less>
WANN_HD void push_heap(K *first, int hole, int top, K value, Less less) {
  int parent = (hole - 1) / 2;
  while (hole > top && less(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <typename K, typename Less>
WANN_HD void adjust_heap(K *first, int hole, int len, K value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  push_heap(first, hole, top, value, less);
}

template <typename K, typename Less>
WANN_HD void heap_sort(K *first, int len, Less less) {  // partial_sort(first, last, last)
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      const K value = first[parent];
      adjust_heap(first, parent, len, value, less);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    const K value = first[last];
    first[last] = first[0];
    adjust_heap(first, 0, last, value, less);
  }
}

template <typename K, typename Less>
WANN_HD void unguarded_linear_insert(K *a, int last, Less less) {
  const K val = a[last];
  int next = last - 1;
  while (less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <typename K, typename Less>
WANN_HD void insertion_sort(K *a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      const K val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else
      unguarded_linear_insert(a, i, less);
  }
}

}  // namespace stdsort_detail

// Sorts a[0..n) exactly as std::sort(a, a + n, less) does.  stack: scratch for 3 (2 floor(log2 n) + 1) ints (128 ints
// cover every n below 2^20).  (Sub-ranges are disjoint, so the order in which pending ranges are finished is immaterial.)
template <typename K, typename Less>
WANN_HD void std_sort_emulated(K *a, int n, int32_t *stack, Less less) {
  using namespace stdsort_detail;
  if (n <= 0) return;
  int lg = 0;
  while ((n >> (lg + 1)) != 0) lg++;
  // explicit stack of (first, last, depth): the right part of every partition is sorted before the left one
  int sp = 0;
  int first = 0, last = n, depth = 2 * lg;
  for (;;) {
    while (last - first > 16) {
      if (depth == 0) {
        heap_sort(a + first, last - first, less);
        break;
      }
      --depth;
      // median of (first + 1, mid, last - 1) to first
      const int mid = first + (last - first) / 2;
      const int ia = first + 1, ib = mid, ic = last - 1;
      int m;
      if (less(a[ia], a[ib])) {
        if (less(a[ib], a[ic])) m = ib;
        else if (less(a[ia], a[ic])) m = ic;
        else m = ia;
      } else if (less(a[ia], a[ic])) m = ia;
      else if (less(a[ib], a[ic])) m = ic;
      else m = ib;
      {
        const K t = a[first];
        a[first] = a[m];
        a[m] = t;
      }
      // unguarded partition of (first + 1, last) around a[first]
      int lo = first + 1, hi = last;
      const K pivot = a[first];
      for (;;) {
        while (less(a[lo], pivot)) ++lo;
        --hi;
        while (less(pivot, a[hi])) --hi;
        if (!(lo < hi)) break;
        const K t = a[lo];
        a[lo] = a[hi];
        a[hi] = t;
        ++lo;
      }
      const int cut = lo;
      // "recursion" on (cut, last) comes first: remember the left part (first, cut) for later
      stack[sp++] = first;
      stack[sp++] = cut;
      stack[sp++] = depth;
      first = cut;
    }
    if (sp == 0) break;
    depth = stack[--sp];
    last = stack[--sp];
    first = stack[--sp];
  }
  // final insertion sort
  if (n > 16) {
    insertion_sort(a, 0, 16, less);
    for (int i = 16; i != n; ++i) unguarded_linear_insert(a, i, less);
  } else
    insertion_sort(a, 0, n, less);
}

}  // namespace wann
