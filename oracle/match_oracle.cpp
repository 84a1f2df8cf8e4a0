// ============================================================================================
// oracle/match_oracle.cpp — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of cv::BFMatcher(NORM_HAMMING, crossCheck=false).match(query, train, matches)
// as called at frontend.cpp:614, frontend.cpp:1123 and backend.cpp:1072 of the reference.
// OpenCV (un-vendored, un-pinned) implements it as knnMatch(k=1) -> batchDistance(K=1,
// NORM_HAMMING, dtype CV_32S): per query row the best distance starts at INT_MAX / index -1 and
// is replaced only when d < best while scanning train rows in increasing index, so the lowest
// index wins ties.  One DMatch{queryIdx=i, trainIdx, imgIdx=0, distance=(float)d} per query.
// Not checkable against OpenCV here (absent) => parity unpinned; tests pin it by definition.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
// ============================================================================================
#include <climits>
#include <cstdint>
#include <cstring>

extern "C" {

// descriptors: rows of `width` bytes (32 for ORB). Outputs: train index (or -1 if nt==0) and distance.
void orc_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int width, int32_t* train_idx, int32_t* dist) {
  for (int i = 0; i < nq; i++) {
    int best = INT_MAX, bi = -1;
    const uint8_t* a = q + (size_t)i * width;
    for (int j = 0; j < nt; j++) {
      const uint8_t* b = t + (size_t)j * width;
      int d = 0;
      for (int k = 0; k < width; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
      if (d < best) { best = d; bi = j; }
    }
    train_idx[i] = bi;
    dist[i] = best;
  }
}

// Faster equivalent for the timed CPU baseline (64-bit words; same results). width must be 32.
void orc_match_hamming256(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* train_idx, int32_t* dist) {
  for (int i = 0; i < nq; i++) {
    uint64_t a[4]; memcpy(a, q + (size_t)i * 32, 32);
    int best = INT_MAX, bi = -1;
    for (int j = 0; j < nt; j++) {
      uint64_t b[4]; memcpy(b, t + (size_t)j * 32, 32);
      int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
              __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
      if (d < best) { best = d; bi = j; }
    }
    train_idx[i] = bi;
    dist[i] = best;
  }
}

// Backend shape (backend.cpp:1068-1077): every (obs, landmark) pair with distance < max_dist,
// obs-major then landmark order.  Returns the number of pairs (may exceed cap; only cap written).
int orc_match_hamming_thresh(const uint8_t* q, int nq, const uint8_t* t, int nt, int max_dist, int32_t* pairs, int cap) {
  int n = 0;
  for (int i = 0; i < nq; i++) {
    uint64_t a[4]; memcpy(a, q + (size_t)i * 32, 32);
    for (int j = 0; j < nt; j++) {
      uint64_t b[4]; memcpy(b, t + (size_t)j * 32, 32);
      int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
              __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
      if (d < max_dist) {
        if (n < cap) { pairs[3 * n] = i; pairs[3 * n + 1] = j; pairs[3 * n + 2] = d; }
        n++;
      }
    }
  }
  return n;
}
}
