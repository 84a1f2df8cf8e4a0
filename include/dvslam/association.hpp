// dvslam/association.hpp — Backend::syncCallback's association loop (backend.cpp:735-797) with the reference's SEQUENTIAL
// semantics on top of the batched device evaluation (dvs_associate_candidates).
//
// The reference tests observation i against the landmark database as it stands after observations 0 .. i-1 were applied: a
// matched landmark is re-triangulated at once (landmark_info.triangulate, backend.cpp:772), so its position may have moved when a
// later observation of the same keyframe is tested.  (Landmarks CREATED by this keyframe only join the database after the loop,
// backend.cpp:779-797: they are never candidates, exactly as in the snapshot.)  The device evaluates every observation against
// the snapshot at once; this adapter then walks the observations in order, and whenever the caller's onMatch moved a landmark it
// re-evaluates — on the host, in the device kernel's own arithmetic (reprojectPoint, backend.cpp:1153-1173) — exactly the later
// observations that have that landmark among their descriptor candidates.  Result: identical to the one-by-one loop.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../dvslam_hip.h"

namespace dvslam {

// reprojection error of one (observation, landmark) pair: |pixel - reprojectPoint(X)| with the reference's mixed precision
// (double camera coordinates, float pixel, float difference, double norm; (-1, -1) behind the camera)
inline double reprojection_error(const float* px, const float* X, const double* R, const double* t, double fx, double fy, double cx, double cy) {
  const double d0 = (double)X[0] - t[0], d1 = (double)X[1] - t[1], d2 = (double)X[2] - t[2];
  const double c0 = R[0] * d0 + R[3] * d1 + R[6] * d2, c1 = R[1] * d0 + R[4] * d1 + R[7] * d2, c2 = R[2] * d0 + R[5] * d1 + R[8] * d2;
  float u = -1.f, v = -1.f;
  if (!(c2 <= 0)) { u = (float)(fx * c0 / c2 + cx); v = (float)(fy * c1 / c2 + cy); }
  const float dx = px[0] - u, dy = px[1] - v;
  return std::sqrt((double)dx * dx + (double)dy * dy);
}

// onMatch(observation, landmark, xyz): called for every association in observation order; returns true if it moved the
// landmark and wrote the new position to xyz[3] (the caller's triangulation).  lm_xyz is updated in place.
// Returns best[i] = landmark index or -1, exactly as the reference's loop would assign them.
inline std::vector<int32_t> associateSequential(dvs_matcher* ctx, const uint8_t* obs_desc, const float* obs_px, int nobs, const uint8_t* lm_desc,
                                                float* lm_xyz, int nlm, const double* R, const double* t, double fx, double fy, double cx, double cy,
                                                double max_descriptor_distance, double max_reprojection_distance,
                                                const std::function<bool(int, int, float*)>& onMatch) {
  std::vector<int32_t> best(nobs, -1);
  std::vector<int64_t> offs((size_t)nobs + 1, 0);
  std::vector<int32_t> cand;
  int64_t total = 0;
  // one evaluation in the common case: a candidate list sized for 16 landmarks below the Hamming bound per observation; the call is
  // repeated (threshold match, reprojection kernel, uploads) only when a keyframe has more
  cand.resize((size_t)nobs * 16 + 64);
  dvs_status st = dvs_associate_candidates(ctx, obs_desc, obs_px, nobs, lm_desc, lm_xyz, nlm, R, t, fx, fy, cx, cy, max_descriptor_distance,
                                           max_reprojection_distance, best.data(), offs.data(), cand.data(), (int64_t)cand.size(), &total);
  if (st == DVS_ERR_CAPACITY && total > (int64_t)cand.size()) {
    cand.resize((size_t)total);
    st = dvs_associate_candidates(ctx, obs_desc, obs_px, nobs, lm_desc, lm_xyz, nlm, R, t, fx, fy, cx, cy, max_descriptor_distance,
                                  max_reprojection_distance, best.data(), offs.data(), cand.data(), total, &total);
  }
  if (st != DVS_OK) throw std::runtime_error(std::string("dvs_associate_candidates: ") + dvs_last_error());
  std::vector<uint8_t> moved(nlm, 0);
  bool any_moved = false;
  for (int i = 0; i < nobs; i++) {
    if (any_moved) {  // does a candidate of this observation sit at a new position?  then its snapshot result is stale
      bool stale = false;
      for (int64_t p = offs[i]; p < offs[i + 1] && !stale; p++) stale = moved[cand[p]] != 0;
      if (stale) {
        int bl = -1;
        double be = DBL_MAX;
        for (int64_t p = offs[i]; p < offs[i + 1]; p++) {  // landmark order, first wins ties (backend.cpp:1091-1111)
          const int j = cand[p];
          const double e = reprojection_error(obs_px + 2 * (size_t)i, lm_xyz + 3 * (size_t)j, R, t, fx, fy, cx, cy);
          if (e < max_reprojection_distance && e < be) { bl = j; be = e; }
        }
        best[i] = bl;
      }
    }
    if (best[i] >= 0 && onMatch) {
      float xyz[3] = {lm_xyz[3 * (size_t)best[i]], lm_xyz[3 * (size_t)best[i] + 1], lm_xyz[3 * (size_t)best[i] + 2]};
      if (onMatch(i, best[i], xyz)) {
        for (int k = 0; k < 3; k++) lm_xyz[3 * (size_t)best[i] + k] = xyz[k];
        moved[best[i]] = 1; any_moved = true;
      }
    }
  }
  return best;
}

}  // namespace dvslam
