// nth_select.h -- std::nth_element as libstdc++ implements it, usable in device code.
//
// IVox::GetClosestPoint hands esti_plane its <= 5 neighbours in the order three std::nth_element calls leave them
// (/root/reference/src/jueying_lio/include/ivox3d/ivox3d_node.hpp:176-181, ivox3d.h:173-178).  That order is a property of the
// standard library the reference is built with, not of the reference's code; the kernels' default order is ascending distance
// (same neighbour set, plane identical up to the float rounding of a row-permuted QR).  PCM_FLAG_REFERENCE_KNN_ORDER reproduces
// the libstdc++ order on the device, for which std::nth_element itself is restated here -- from the container's own
// <bits/stl_algo.h> and <bits/stl_heap.h> (GCC 11.4; the algorithm is unchanged since GCC 4.x):
//   nth_element        stl_algo.h:4795-4812   depth limit 2 * lg(n)
//   __introselect      stl_algo.h:1962-1986   median-of-three quickselect on (first + 1, mid, last - 1), ranges of <= 3 insertion-sorted
//   __unguarded_partition(_pivot), __move_median_to_first   stl_algo.h:1876-1907, 78-98
//   __insertion_sort, __unguarded_linear_insert             stl_algo.h:1797-1849
//   __heap_select -> __make_heap / __pop_heap / __adjust_heap / __push_heap   stl_algo.h:1640-1650, stl_heap.h:132-146, 221-266, 337-360
// Checked on the host against the real std::nth_element, permutation for permutation (tests/test_knn_order.py compiles this
// header with g++), ties included: the comparison is `a.d < b.d` only (DistPoint::operator<, ivox3d_node.hpp:118).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PCM_NTH_FN __host__ __device__ inline
#else
#define PCM_NTH_FN inline
#endif

namespace pcm {

struct DistId {
  float d;        // squared distance (the reference keeps the same float widened to double: same order)
  uint32_t id;
};

PCM_NTH_FN bool nth_less(const DistId& a, const DistId& b) { return a.d < b.d; }
PCM_NTH_FN void nth_swap(DistId* a, int i, int j) { const DistId t = a[i]; a[i] = a[j]; a[j] = t; }

// stl_heap.h:132-146
PCM_NTH_FN void nth_push_heap(DistId* a, int hole, int top, DistId value) {
  int parent = (hole - 1) / 2;
  while (hole > top && nth_less(a[parent], value)) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = value;
}

// stl_heap.h:221-248
PCM_NTH_FN void nth_adjust_heap(DistId* a, int hole, int len, DistId value) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (nth_less(a[second], a[second - 1])) second--;
    a[hole] = a[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    a[hole] = a[second - 1];
    hole = second - 1;
  }
  nth_push_heap(a, hole, top, value);
}

// __heap_select(first, middle, last)  stl_algo.h:1640-1650 with __make_heap (stl_heap.h:337-360) and __pop_heap (:251-266)
PCM_NTH_FN void nth_heap_select(DistId* a, int middle, int last) {
  if (middle >= 2) {
    int parent = (middle - 2) / 2;
    for (;;) {
      const DistId v = a[parent];
      nth_adjust_heap(a, parent, middle, v);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; i++) {
    if (nth_less(a[i], a[0])) {
      const DistId v = a[i];
      a[i] = a[0];
      nth_adjust_heap(a, 0, middle, v);
    }
  }
}

// __insertion_sort  stl_algo.h:1817-1838
PCM_NTH_FN void nth_insertion_sort(DistId* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const DistId v = a[i];
    if (nth_less(v, a[first])) {
      for (int k = i; k > first; k--) a[k] = a[k - 1];   // move_backward(first, i, i + 1)
      a[first] = v;
    } else {   // __unguarded_linear_insert
      int lastp = i, next = i - 1;
      while (nth_less(v, a[next])) {
        a[lastp] = a[next];
        lastp = next;
        --next;
      }
      a[lastp] = v;
    }
  }
}

// std::nth_element(a, a + nth, a + n)
PCM_NTH_FN void nth_element_libstdcxx(DistId* a, int nth, int n) {
  if (n == 0 || nth == n) return;
  int depth = 0;
  for (int m = n; m > 1; m >>= 1) depth++;   // std::__lg(n)
  depth *= 2;
  int first = 0, last = n;
  while (last - first > 3) {
    if (depth == 0) {
#if defined(PCM_NTH_COUNT_HEAP_SELECT)
      PCM_NTH_COUNT_HEAP_SELECT;   // host test hook: the depth limit was reached
#endif
      nth_heap_select(a + first, nth + 1 - first, last - first);
      nth_swap(a, first, nth);
      return;
    }
    --depth;
    // __unguarded_partition_pivot: the median of a[first + 1], a[mid], a[last - 1] goes to a[first]
    const int mid = first + (last - first) / 2;
    {
      const int ia = first + 1, ib = mid, ic = last - 1;
      if (nth_less(a[ia], a[ib])) {
        if (nth_less(a[ib], a[ic])) nth_swap(a, first, ib);
        else if (nth_less(a[ia], a[ic])) nth_swap(a, first, ic);
        else nth_swap(a, first, ia);
      } else if (nth_less(a[ia], a[ic])) nth_swap(a, first, ia);
      else if (nth_less(a[ib], a[ic])) nth_swap(a, first, ic);
      else nth_swap(a, first, ib);
    }
    // __unguarded_partition(first + 1, last, pivot = first)
    int lo = first + 1, hi = last;
    for (;;) {
      while (nth_less(a[lo], a[first])) ++lo;
      --hi;
      while (nth_less(a[first], a[hi])) --hi;
      if (!(lo < hi)) break;
      nth_swap(a, lo, hi);
      ++lo;
    }
    const int cut = lo;
    if (cut <= nth) first = cut; else last = cut;
  }
  nth_insertion_sort(a, first, last);
}

}  // namespace pcm
