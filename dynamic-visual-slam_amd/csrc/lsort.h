// lsort.h — operation-for-operation replica of libstdc++'s std::sort (GCC 4.x..14: introsort with
// median-of-3 pivot, _S_threshold = 16, heapsort fallback at depth 2*floor(lg n), final insertion
// sort).  The reference sorts its quad-tree expansion candidates with std::sort and an order that
// has ties (ORBextractor.cpp:538-553, 700), so WHICH permutation comes out is part of the result.
// This header is compiled for host and device; tests/test_host_logic.py checks it against the real
// std::sort on adversarial tie-heavy inputs.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LSORT_HD __host__ __device__ __forceinline__
#else
#define LSORT_HD inline
#endif

namespace lsort {

// element = 64-bit word; ordering = less(a, b) on (a >> SHIFT): payload bits below SHIFT are ignored.
template <int SHIFT>
struct Less {
  LSORT_HD bool operator()(uint64_t a, uint64_t b) const { return (a >> SHIFT) < (b >> SHIFT); }
};

template <class T>
LSORT_HD void swp(T* a, T* b) { T t = *a; *a = *b; *b = t; }

template <class T, class C>
LSORT_HD void unguarded_linear_insert(T* last, C comp) {
  T val = *last;
  T* next = last - 1;
  while (comp(val, *next)) { *last = *next; last = next; --next; }
  *last = val;
}

template <class T, class C>
LSORT_HD void insertion_sort(T* first, T* last, C comp) {
  if (first == last) return;
  for (T* i = first + 1; i != last; ++i) {
    if (comp(*i, *first)) {
      T val = *i;
      for (T* p = i; p != first; --p) *p = *(p - 1);  // move_backward(first, i, i + 1)
      *first = val;
    } else {
      unguarded_linear_insert(i, comp);
    }
  }
}

template <class T, class C>
LSORT_HD void push_heap_(T* first, long holeIndex, long topIndex, T value, C comp) {
  long parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && comp(first[parent], value)) {
    first[holeIndex] = first[parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  first[holeIndex] = value;
}

template <class T, class C>
LSORT_HD void adjust_heap(T* first, long holeIndex, long len, T value, C comp) {
  const long topIndex = holeIndex;
  long secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (comp(first[secondChild], first[secondChild - 1])) secondChild--;
    first[holeIndex] = first[secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    first[holeIndex] = first[secondChild - 1];
    holeIndex = secondChild - 1;
  }
  push_heap_(first, holeIndex, topIndex, value, comp);
}

template <class T, class C>
LSORT_HD void heap_sort_all(T* first, T* last, C comp) {  // __partial_sort(first, last, last)
  long len = last - first;
  if (len >= 2) {  // __make_heap
    long parent = (len - 2) / 2;
    while (true) {
      T value = first[parent];
      adjust_heap(first, parent, len, value, comp);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {  // __sort_heap / __pop_heap
    --last;
    T value = *last;
    *last = *first;
    adjust_heap(first, 0L, (long)(last - first), value, comp);
  }
}

template <class T, class C>
LSORT_HD void move_median_to_first(T* result, T* a, T* b, T* c, C comp) {
  if (comp(*a, *b)) {
    if (comp(*b, *c)) swp(result, b);
    else if (comp(*a, *c)) swp(result, c);
    else swp(result, a);
  } else if (comp(*a, *c)) swp(result, a);
  else if (comp(*b, *c)) swp(result, c);
  else swp(result, b);
}

template <class T, class C>
LSORT_HD T* unguarded_partition(T* first, T* last, T* pivot, C comp) {
  while (true) {
    while (comp(*first, *pivot)) ++first;
    --last;
    while (comp(*pivot, *last)) --last;
    if (!(first < last)) return first;
    swp(first, last);
    ++first;
  }
}

// std::sort(first, first + n, comp).  Recursion of __introsort_loop is unrolled with an explicit
// stack of (first, last, depth) — the right part is pushed, the left part continues, exactly the
// order in which libstdc++ recurses (right half first via recursion, then loops on the left).
// NOTE the libstdc++ code recurses into [cut, last) BEFORE continuing with [first, cut); since the
// two ranges are disjoint the visiting order does not change the result, only the stack shape.
template <class T, class C>
LSORT_HD void sort(T* first, long n, C comp) {
  if (n <= 0) return;
  T* last = first + n;
  int lg = 0;
  for (long m = n; m > 1; m >>= 1) lg++;
  struct Frame { T* f; T* l; int d; };
  Frame stack[64];
  int sp = 0;
  stack[sp++] = Frame{first, last, lg * 2};
  while (sp > 0) {
    Frame fr = stack[--sp];
    T* f = fr.f; T* l = fr.l; int depth = fr.d;
    while (l - f > 16) {
      if (depth == 0) { heap_sort_all(f, l, comp); break; }
      --depth;
      T* mid = f + (l - f) / 2;
      move_median_to_first(f, f + 1, mid, l - 1, comp);
      T* cut = unguarded_partition(f + 1, l, f, comp);
      stack[sp++] = Frame{cut, l, depth};
      l = cut;
    }
  }
  if (n > 16) {  // __final_insertion_sort
    insertion_sort(first, first + 16, comp);
    for (T* i = first + 16; i != last; ++i) unguarded_linear_insert(i, comp);
  } else {
    insertion_sort(first, last, comp);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same std::sort, restated so that it parallelises (the GPU quad-tree runs this form, one wavefront per range):
//
//  * __unguarded_partition: with GE = positions of [first, last) holding an element that stops the left scan (!(x < pivot))
//    in ascending order, and LE = positions that stop the right scan (!(pivot < x)) in descending order, the k-th swap of the
//    Hoare loop exchanges GE[k] and LE[k] as long as GE[k] < LE[k] (both scans only ever cross untouched elements: swapped ones
//    lie outside the open interval between the previous pair).  With K swaps done, the left scan stops at GE[K] or, if that lies
//    beyond it, at LE[K-1] — which now holds a GE element; the returned cut is the smaller of the two.
//  * __final_insertion_sort: after the introsort loop every range of <= 16 elements holds nothing smaller than its left
//    neighbour range's elements, so the insertion sort never carries an element across a range boundary: it is a stable sort
//    of every leaf range by itself (position = first + #smaller + #equal-before).
//  * ranges are disjoint, so the order in which they are partitioned is irrelevant.
// sort_ranked is the sequential statement of that form, used by the host tests (against the real std::sort) and mirrored
// lane-for-lane by qt_sort_block in orb_kernels.h.  Lp / Rp: scratch of n ints each.
template <class T, class C>
LSORT_HD long ranked_partition(T* a, long f, long l, C comp, int* Lp, int* Rp) {
  move_median_to_first(a + f, a + f + 1, a + f + (l - f) / 2, a + l - 1, comp);
  const T pivot = a[f];
  long nge = 0, nle = 0;
  for (long i = f + 1; i < l; i++) {
    if (!comp(a[i], pivot)) Lp[f + nge++] = (int)i;
    if (!comp(pivot, a[i])) Rp[f + nle++] = (int)i;  // ascending; k-th from the right = Rp[f + nle - 1 - k]
  }
  const long mm = nge < nle ? nge : nle;
  long K = 0;
  while (K < mm && Lp[f + K] < Rp[f + nle - 1 - K]) K++;
  for (long k = 0; k < K; k++) swp(a + Lp[f + k], a + Rp[f + nle - 1 - k]);
  const long big = 0x7fffffff;
  const long lK = K < nge ? Lp[f + K] : big;
  const long rprev = K > 0 ? Rp[f + nle - K] : big;
  return lK < rprev ? lK : rprev;
}

template <class T, class C>
LSORT_HD void ranked_leaf(T* a, long f, long l, C comp) {  // stable placement of a[f, l), l - f <= 16
  T tmp[16];
  int pos[16];
  for (long i = f; i < l; i++) {
    int r = 0;
    for (long j = f; j < l; j++) r += (comp(a[j], a[i]) || (!comp(a[i], a[j]) && j < i)) ? 1 : 0;
    tmp[i - f] = a[i]; pos[i - f] = r;
  }
  for (long i = f; i < l; i++) a[f + pos[i - f]] = tmp[i - f];
}

template <class T, class C>
LSORT_HD void sort_ranked(T* a, long n, C comp, int* Lp, int* Rp) {
  if (n <= 1) return;
  int lg = 0;
  for (long m = n; m > 1; m >>= 1) lg++;
  struct Frame { long f, l; int d; };
  Frame stack[128];
  int sp = 0;
  stack[sp++] = Frame{0, n, lg * 2};
  while (sp > 0) {
    const Frame fr = stack[--sp];
    if (fr.l - fr.f <= 16) { ranked_leaf(a, fr.f, fr.l, comp); continue; }
    if (fr.d == 0) { heap_sort_all(a + fr.f, a + fr.l, comp); continue; }
    const long cut = ranked_partition(a, fr.f, fr.l, comp, Lp, Rp);
    stack[sp++] = Frame{cut, fr.l, fr.d - 1};
    stack[sp++] = Frame{fr.f, cut, fr.d - 1};
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// std::nth_element and std::partition as KeyPointsFilter::retainBest calls them (OpenCV features2d/src/keypoint.cpp, reached from
// cv::ORB: row N4).  retainBest's result ORDER is whatever these two leave behind, so it is part of "identical rows".
//   nth_element = __introselect: __unguarded_partition_pivot until <= 3 elements remain around nth (then __insertion_sort), or
//   __heap_select + iter_swap(first, nth) once the depth limit 2 * floor(lg n) is spent.
template <class T, class C>
LSORT_HD void heap_select(T* first, T* middle, T* last, C comp) {
  const long len = middle - first;
  if (len >= 2) {  // __make_heap(first, middle)
    long parent = (len - 2) / 2;
    while (true) {
      T value = first[parent];
      adjust_heap(first, parent, len, value, comp);
      if (parent == 0) break;
      parent--;
    }
  }
  for (T* i = middle; i < last; ++i)
    if (comp(*i, *first)) {  // __pop_heap(first, middle, i)
      T value = *i;
      *i = *first;
      adjust_heap(first, 0L, len, value, comp);
    }
}

template <class T, class C>
LSORT_HD void nth_element(T* first, T* nth, T* last, C comp) {
  if (first == last || nth == last) return;
  int depth = 0;
  for (long m = last - first; m > 1; m >>= 1) depth++;
  depth *= 2;
  while (last - first > 3) {
    if (depth == 0) {
      heap_select(first, nth + 1, last, comp);
      swp(first, nth);
      return;
    }
    --depth;
    T* mid = first + (last - first) / 2;
    move_median_to_first(first, first + 1, mid, last - 1, comp);
    T* cut = unguarded_partition(first + 1, last, first, comp);
    if (cut <= nth) first = cut; else last = cut;
  }
  insertion_sort(first, last, comp);
}

// std::partition for bidirectional (and random access) iterators: __partition(first, last, pred, bidirectional_iterator_tag)
template <class T, class P>
LSORT_HD T* partition(T* first, T* last, P pred) {
  while (true) {
    while (true) {
      if (first == last) return first;
      else if (pred(*first)) ++first;
      else break;
    }
    --last;
    while (true) {
      if (first == last) return first;
      else if (!pred(*last)) --last;
      else break;
    }
    swp(first, last);
    ++first;
  }
}

}  // namespace lsort
