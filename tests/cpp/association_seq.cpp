// N2 sequential semantics (backend.cpp:735-797): two observations of one keyframe choose the same landmark; the first association
// re-triangulates (moves) it, so the second must be tested against the NEW position.  Checks dvslam::associateSequential against
// a literal one-by-one loop over a live database (the reference's order of operations), and that it differs from the snapshot
// where it should.  Exit 0 = ok, 3 = no GPU.
#include <cstdio>
#include <cstring>
#include <vector>
#include "dvslam/association.hpp"

static int popc(const uint8_t* a, const uint8_t* b) { int d = 0; for (int k = 0; k < 32; k++) d += __builtin_popcount(a[k] ^ b[k]); return d; }

int main() {
  if (dvs_device_count() < 1) { std::printf("no device: association adapter compiled, nothing run\n"); return 3; }
  dvs_matcher* m = nullptr;
  if (dvs_matcher_create(0, &m) != DVS_OK) return 1;
  const double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  const double fx = 600, fy = 600, cx = 320, cy = 240;
  const int nlm = 300, nobs = 200;
  std::vector<uint8_t> lmd((size_t)nlm * 32), obd((size_t)nobs * 32);
  std::vector<float> lmx((size_t)nlm * 3), obp((size_t)nobs * 2);
  uint32_t s = 99;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int j = 0; j < nlm; j++) {
    for (int k = 0; k < 32; k++) lmd[(size_t)j * 32 + k] = (uint8_t)rnd();
    lmx[3 * j] = -1.0f + 2.0f * (rnd() % 1000) / 1000.f; lmx[3 * j + 1] = -0.7f + 1.4f * (rnd() % 1000) / 1000.f; lmx[3 * j + 2] = 2.0f + (rnd() % 1000) / 500.f;
  }
  // observation i looks at landmark (i * 7) % 150: ~every landmark below 150 is chosen by one or two observations; a few bit flips, a
  // pixel within ~2 px of the projection.  Landmark 5 gets an exact twin (landmark 210, same descriptor, 3 px away) as second best.
  memcpy(&lmd[(size_t)210 * 32], &lmd[(size_t)5 * 32], 32);
  lmx[3 * 210] = lmx[3 * 5] + 0.015f; lmx[3 * 210 + 1] = lmx[3 * 5 + 1]; lmx[3 * 210 + 2] = lmx[3 * 5 + 2];
  for (int i = 0; i < nobs; i++) {
    const int j = i < 2 ? 5 : (i * 7) % 150;           // observations 0 and 1 both see landmark 5
    memcpy(&obd[(size_t)i * 32], &lmd[(size_t)j * 32], 32);
    for (int f = 0; f < 6; f++) obd[(size_t)i * 32 + rnd() % 32] ^= (uint8_t)(1u << (rnd() % 8));
    obp[2 * i] = (float)(fx * lmx[3 * j] / lmx[3 * j + 2] + cx) + ((int)(rnd() % 300) - 150) / 100.f;
    obp[2 * i + 1] = (float)(fy * lmx[3 * j + 1] / lmx[3 * j + 2] + cy) + ((int)(rnd() % 300) - 150) / 100.f;
  }
  // "triangulation": every association moves its landmark 4 cm sideways (9.6 px at 2.5 m ... 12 px at 2 m: beyond the 5 px gate)
  auto onMatch = [&](int, int, float* xyz) { xyz[0] += 0.04f; return true; };
  // (a) literal sequential loop over a live copy of the database
  std::vector<float> live = lmx;
  std::vector<int> want(nobs, -1);
  for (int i = 0; i < nobs; i++) {
    int bl = -1; double be = 1e300;
    for (int j = 0; j < nlm; j++) {
      if (!((float)popc(&obd[(size_t)i * 32], &lmd[(size_t)j * 32]) < 50.0)) continue;
      const double e = dvslam::reprojection_error(&obp[2 * i], &live[3 * j], R, t, fx, fy, cx, cy);
      if (e < 5.0 && e < be) { bl = j; be = e; }
    }
    want[i] = bl;
    if (bl >= 0) live[3 * bl] += 0.04f;
  }
  // (b) snapshot (no updates) and (c) the adapter
  std::vector<int32_t> snap(nobs, -1);
  if (dvs_associate(m, obd.data(), obp.data(), nobs, lmd.data(), lmx.data(), nlm, R, t, fx, fy, cx, cy, 50.0, 5.0, snap.data()) != DVS_OK) return 1;
  std::vector<float> db = lmx;
  std::vector<int32_t> got = dvslam::associateSequential(m, obd.data(), obp.data(), nobs, lmd.data(), db.data(), nlm, R, t, fx, fy, cx, cy, 50.0, 5.0, onMatch);
  int diff_seq = 0, diff_snap = 0, assoc = 0;
  for (int i = 0; i < nobs; i++) { diff_seq += got[i] != want[i]; diff_snap += snap[i] != want[i]; assoc += want[i] >= 0; }
  bool same_db = memcmp(db.data(), live.data(), db.size() * 4) == 0;
  std::printf("associations %d, adapter vs sequential loop: %d differences, snapshot vs sequential loop: %d differences, obs0 -> %d, obs1 -> %d (snapshot %d), db equal %d\n",
              assoc, diff_seq, diff_snap, got[0], got[1], snap[1], (int)same_db);
  dvs_matcher_destroy(m);
  // observation 0 takes landmark 5 and moves it; observation 1 must then fall to the twin 210 (snapshot says 5)
  if (diff_seq != 0 || !same_db || got[0] != 5 || got[1] != 210 || snap[1] != 5 || diff_snap == 0) return 1;
  return 0;
}
