// Host check of pointcloud-slam_amd/csrc/nth_select.h (the device restatement of libstdc++ std::nth_element) against the real
// std::nth_element of the container: identical permutations on 200 000 arrays -- random, tied, sorted, reversed, organ-pipe,
// constant and median-of-three-killer inputs (the last reach the heap-select fallback).  Built and run by tests/test_knn_order.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
static long g_heap = 0;
#define PCM_NTH_COUNT_HEAP_SELECT g_heap++
#include "nth_select.h"
struct DP { double dist; void* node; int idx; bool operator<(const DP& r) const { return dist < r.dist; } };
int main() {
  std::mt19937 rng(7);
  long cases = 0, heap = 0;
  for (int rep = 0; rep < 200000; rep++) {
    int n = 1 + rng() % (rep % 50 == 0 ? 300 : 40);
    int kind = rng() % 8;
    std::vector<float> d(n);
    for (int i = 0; i < n; i++) {
      switch (kind) {
        case 0: d[i] = (float)(rng() % 1000) * 0.37f; break;
        case 1: d[i] = (float)(rng() % 4); break;            // many ties
        case 2: d[i] = (float)i; break;                      // sorted
        case 3: d[i] = (float)(n - i); break;                // reverse
        case 4: d[i] = (float)(i < n / 2 ? i : n - i); break;  // organ pipe
        case 5: d[i] = 1.f; break;                           // all equal
        case 6: d[i] = (float)((i * 7919) % 13); break;
        default: d[i] = (float)(i % 2 ? i : n - i) + (rng() % 3); break;
      }
    }
    if (kind == 7 && n >= 8) {   // median-of-3 killer (Musser): drives introselect to its depth limit
      int k = n / 2;
      for (int i = 0; i < k; i++) { d[i] = (i % 2 == 0) ? (float)(i + 1) : (float)(k + i + (k % 2 == 0 ? 0 : 1)); }
      for (int i = k; i < n; i++) d[i] = (float)((i - k + 1) * 2);
    }
    int nth = rng() % n;
    if (rep % 3 == 0) nth = std::min(4, n - 1);
    if (rep % 7 == 0) nth = 0;
    std::vector<DP> a(n); std::vector<pcm::DistId> b(n);
    for (int i = 0; i < n; i++) { a[i].dist = d[i]; a[i].node = nullptr; a[i].idx = i; b[i].d = d[i]; b[i].id = (uint32_t)i; }
    std::nth_element(a.begin(), a.begin() + nth, a.end());
    pcm::nth_element_libstdcxx(b.data(), nth, n);
    for (int i = 0; i < n; i++) if ((uint32_t)a[i].idx != b[i].id) { printf("MISMATCH rep %d n %d nth %d kind %d at %d\n", rep, n, nth, kind, i); return 1; }
    cases++;
  }
  printf("ok %ld cases, heap_select reached %ld times\n", cases, g_heap);
  return 0;
}
