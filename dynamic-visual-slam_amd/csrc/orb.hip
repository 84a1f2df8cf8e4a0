// orb.hip — host side + C-ABI of the ORB extractor (boundary B1, include/dvslam_hip.h).
// Replaces ORB_SLAM3::ORBextractor (reference include/dynamic_visual_slam/ORBextractor.hpp:44-110,
// src/ORBextractor.cpp:409-469 ctor, 1086-1194 operator()/ComputePyramid).  Host code only builds
// tables and enqueues kernels; there is no CPU compute path.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <chrono>
#include <vector>
#include "common.h"
#ifdef DVS_TEST_HOOKS
#include "../../include/dvslam_hip_test.h"
#endif
#include "orb_kernels.h"

namespace dvs {

static inline int cv_round_f(float v) { return (int)lrintf(v); }  // cvRound: round-half-even
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

}  // namespace dvs

using namespace dvs;

static int env_int(const char* name, int dflt) { return dvs::env_switch(name, dflt); }   // (util.hip: the one table of switches)

struct dvs_orb {
  dvs_orb_params prm;
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipStream_t pf_stream = nullptr;         // highest priority: the NEXT batch's level chain (dvs_orb_hint_next_batch_device)
  hipStream_t aux_stream = nullptr;        // lowest priority: in-step level chain, blur, a deferred descriptor stage
  hipEvent_t ev_fork = nullptr;            // FAST finished (main stream): the blur's fork when the caller gave no after-FAST event
  hipEvent_t ev_blur = nullptr;            // blur finished (auxiliary stream): the descriptor stage's join
  hipEvent_t ev_start = nullptr;           // start of a call on the main stream: gate of an in-step level chain on the auxiliary stream
  hipEvent_t ev_chain_gate = nullptr;      // gate of the next batch's chain when the previous call left no end-of-call event
  hipEvent_t ev_prefetch = nullptr;        // the announced batch's chain is complete; alternates between ev_pf2[]: this call may still
  hipEvent_t ev_pf2[2] = {nullptr, nullptr};   // wait for its own chain after it has launched (and recorded) the next batch's
  int pf_idx = 0;
  hipEvent_t ev_level[DVS_MAX_LEVELS] = {};  // level l of the pyramid is complete (in-step chain beside FAST)
  bool overlap = true;
  bool single_stream = false;              // dvs_orb_create_single_stream: no auxiliary / prefetch streams exist, overlap stays off
  int max_batch = 1;
  // ctor tables (ORBextractor.cpp:414-445)
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> feat_per_level;
  int umax[16];
  // per-resolution state
  int rows = 0, cols = 0;
  Geom geom;
  Geom* d_geom = nullptr;
  Cell* d_cells = nullptr;
  BlurTile* d_tiles = nullptr;
  BlurStrip* d_strips = nullptr;
  BlurCol* d_blurcols = nullptr;   // matrix-core blur: work items + operand fragment table (k_blur_mfma)
  uint4* d_blurtab = nullptr;
  int n_blurcols = 0, blur_avt = 0;
  bool blur_mfma_ok = false;       // weights fit the int8 band products (0 .. 127, sum 256)
  ResizeGroup* d_rgroups = nullptr;
  PyrTile* d_pyrtiles = nullptr;
  int *d_xofs = nullptr, *d_alpha = nullptr, *d_yofs = nullptr, *d_beta = nullptr;
  // pyramid blocks: d_pyr = this batch; d_pyr_alt = the announced next batch (built on pf_stream beside this batch's FAST, swapped in
  // by the next call); d_pyr_3rd = with deferred descriptor stages the pyramid of the batch before is still being read while the next
  // one is built, so the three rotate.  The last two are allocated on first use.
  u8 *d_pyr = nullptr, *d_pyr_alt = nullptr, *d_pyr_3rd = nullptr, *d_pyr_4th = nullptr, *d_blur = nullptr;
  // Depth of the rings a pipelined caller's batches rotate through (pyramids, candidate-list sets, level keypoint sets, blurred blocks): 3, or 4
  // in the four-stream form (dvs_orb_set_tail_stream).  A batch passes chain -> blur -> FAST -> quad-tree -> descriptors on four streams; the
  // chain of batch i + 1 waits for the descriptor stage that last read its buffer — batch i + 1 - ring — so ring periods of the schedule
  // cannot be shorter than that chain of stages: at 8 frames 264 us / 3 = 88 us per step with three buffers, 66 with four (EXPERIMENTS.md).
  int ring = 3;
  int async_run = 0;               // consecutive asynchronous (prefetched + deferred) calls so far
  // deferred descriptor stage (dvs_orb_set_output_event + dvs_orb_set_defer_outputs): ordered on the auxiliary stream only
  hipEvent_t ev_outs[3] = {nullptr, nullptr, nullptr};   // completion of the last ring - 1 deferred stages (ev_out = the latest)
  hipEvent_t ev_out = nullptr, ev_oct = nullptr;  // ... / quad-tree finished (main stream): the deferred stage's join
  int out_gen = 0;                         // deferred stages enqueued so far
  bool out_pending = false;                // the previous call's descriptor stage has not been joined with the main stream
  bool defer_outputs = false;              // dvs_orb_set_defer_outputs
  hipEvent_t output_event = nullptr;       // caller's event: outputs complete
  hipEvent_t guard_event = nullptr;        // caller's event: outputs may only be overwritten behind it (one-shot)
  hipEvent_t after_fast_event = nullptr;   // caller's event, recorded on the main stream behind FAST
  hipEvent_t ev_end = nullptr;             // end of the previous call (gate of the next call's prefetch chain: no extra record)
  hipEvent_t gate_event = nullptr;         // = ev_end or the caller's output event, whichever the last call recorded at its end
  bool pf_joined = false;                  // the prefetched pyramid's completion already precedes the main stream (joined through the blur)
  const u8* next_hint = nullptr;           // one-shot, set by the hint call, consumed by the next extract_batch_device
  // The announced batch's level chain as ONE graph launch (launch_prefetch): the chain is 7 dependent launches whose arguments depend only on
  // (source block, frame count, destination pyramid); callers stream from a ring of device buffers, so the same few argument sets come
  // back — the second time a set is seen its chain is captured, from then on it costs one hipGraphLaunch (4 us of host time against 17:
  // tools/probe/graph_launch_cost.hip).  Used where the host's enqueue is the limit (<= 12 frames per step; DVS_CHAIN_GRAPH=1 / 0 always / never).
  struct ChainGraph { const u8* img0; uint64_t step0, fstride0; int nimg; u8* pyr; hipGraph_t graph; hipGraphExec_t exec; };
  std::vector<ChainGraph> chain_graphs;
  int env_chain_graph = -1;
  int64_t chain_graph_launches = 0;
  bool pf_valid = false;                   // d_pyr_alt holds (or is being filled with) the pyramid of exactly this announced batch:
  const u8* pf_img = nullptr; uint64_t pf_step = 0, pf_fstride = 0; int pf_nimg = 0;
  // switches read ONCE at creation (dvs_orb_create); each is covered by tests/test_gpu_orb.py::test_opt_in_kernel_variants_are_bit_identical
  int env_cascade = -1;            // DVS_CASCADE=1 / 0: all-levels-in-one-launch pyramid always / never (-1 = automatic: <= 8 frames)
  int fast_byte_dma = 0;           // LDS-DMA with byte-aligned global addresses probed exact (DVS_FAST_BYTE_DMA=0 turns it off)
  int env_blur_mfma = 0;           // DVS_BLUR_MFMA=1: the matrix-core blur (k_blur_mfma), 2: its LDS-free form; measured slower in the step, DESIGN.md 4b
  int env_oct_threads = 0;         // DVS_OCT_T=256 / 512: quad-tree workgroup size for every batch size (0 = by batch size)
  int env_host_poll = 1;           // DVS_HOST_POLL=0: three device-to-host copy commands and a stream wait instead of k_export_host
  uint32_t *d_cand = nullptr, *d_pts = nullptr, *d_lvlkp = nullptr;
  uint32_t* d_kpident = nullptr;   // idx[slot] = slot's position in its level's list: the visiting order of DVS_DESC_ORDER=0
  int env_desc_order = 1;          // DVS_DESC_ORDER=0: the descriptor stage visits a level's keypoints in list order instead of tile by tile (k_kp_order)
  uint32_t *d_kpsorted = nullptr, *d_kpsortidx = nullptr;   // a level's keypoints in the descriptor stage's visiting order + their list positions (k_kp_order)
  // level keypoint lists, three sets in rotation: deferred descriptor stages k - 1 and k - 2 may both still read theirs when call k's
  // quad-tree writes (stage k - 3 precedes the level chain call k's FAST waited for) — no wait on the main stream in front of it
  uint32_t* d_lvlkp3[4] = {nullptr, nullptr, nullptr, nullptr};   // (the fourth of each ring: allocated when first used)
  int* d_lvlcount3[4] = {nullptr, nullptr, nullptr, nullptr};
  int lset = 0;
  int *d_nodeof = nullptr, *d_cellcount = nullptr, *d_celloff = nullptr, *d_candtotal = nullptr, *d_lvlcount = nullptr;
  dvs_keypoint* d_kps = nullptr;   // internal outputs for the host entry points [max_batch][outCap]
  u8* d_desc = nullptr;
  int* d_nout = nullptr;
  dvs_keypoint* h_kps = nullptr;   // pinned
  u8* h_desc = nullptr;
  int* h_nout = nullptr;
  int* h_seq = nullptr;            // pinned: sequence number k_export_host publishes (dvs_orb_extract[_batch] poll it)
  int* d_ticket = nullptr;         // ... its last-workgroup ticket
  int export_seq = 0;
  size_t octree_smem = 0;
  int octree_nmax = 0, octree_ptscap = 0;
  // The quad-tree off the main stream (dvs_orb_set_async_quadtree): it runs on the auxiliary stream beside the NEXT call's FAST.  FAST then
  // writes the candidate lists of two sets in turn; launches are graded by level class — the dynamic LDS of a dispatch is uniform, so the
  // small levels get a launch of their own with their own (smaller) node / point capacities instead of the level-0 footprint.
  bool async_oct = false, last_async = false;
  hipStream_t tail_stream = nullptr;       // dvs_orb_set_tail_stream: the descriptor stage of an asynchronous call runs there (the caller's match stream)
  // ... and its blur on the MAIN stream ahead of FAST (main: blur + FAST, prefetch: level chain, auxiliary: quad-tree, tail: descriptors +
  // the caller's match — four streams of similar length for batches whose kernels do not fill the machine).  The blurred block then exists
  // three times: blur k + 1 rewrites the block descriptor stage k - 2 read, which the level chain FAST k + 1 waited for was gated on.
  u8* d_blur3[4] = {nullptr, nullptr, nullptr, nullptr};
  int bset = 0;
  // THREE candidate sets in turn (the second and third allocated on first use): FAST k + 1 writes while tree k reads, and the level chain
  // of call k + 2 — launched in call k + 1, ahead of FAST k + 1 — is gated on tree k - 1, the last reader of the set FAST k + 2 will write:
  // a tree that finished a whole step ago (with two sets the gate would be the tree still running beside that FAST)
  uint32_t* d_cand2[4] = {nullptr, nullptr, nullptr, nullptr};
  int* d_cellcount2[4] = {nullptr, nullptr, nullptr, nullptr};
  int cset = 0;
  hipEvent_t ev_octdone[4] = {nullptr, nullptr, nullptr, nullptr};
  bool octdone_valid[4] = {false, false, false, false};
  int oct_ncls = 0;                        // level classes of the graded launches (0 = no grading): class c = levels [oct_l0[c], oct_l0[c + 1])
  int oct_l0[4] = {0, 0, 0, 0};
  size_t oct_smem_cls[3] = {0, 0, 0};
  int oct_nmax_cls[3] = {0, 0, 0}, oct_ptscap_cls[3] = {0, 0, 0};
  int last_nimg = 0;
  ImgSrc last_src{};
  StageTimer timer;
};

namespace {

void drop_chain_graphs(dvs_orb* h) {
  for (auto& c : h->chain_graphs) {
    if (c.exec) (void)hipGraphExecDestroy(c.exec);
    if (c.graph) (void)hipGraphDestroy(c.graph);
  }
  h->chain_graphs.clear();
}

void free_workspace(dvs_orb* h) {
  void* ptrs[] = {h->d_blurcols, h->d_blurtab, h->d_pyrtiles, h->d_rgroups, h->d_strips, h->d_geom, h->d_cells, h->d_tiles, h->d_xofs, h->d_alpha, h->d_yofs, h->d_beta, h->d_pyr, h->d_blur3[0],
                  h->d_pyr_alt, h->d_pyr_3rd, h->d_pyr_4th, h->d_pts, h->d_lvlkp3[0], h->d_lvlkp3[1], h->d_lvlkp3[2], h->d_lvlkp3[3], h->d_nodeof, h->d_celloff, h->d_candtotal,
                  h->d_lvlcount3[0], h->d_lvlcount3[1], h->d_lvlcount3[2], h->d_lvlcount3[3], h->d_kps, h->d_desc, h->d_nout, h->d_ticket, h->d_kpsorted, h->d_kpsortidx, h->d_kpident};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (int k = 1; k < 4; k++) { if (h->d_blur3[k]) (void)hipFree(h->d_blur3[k]); h->d_blur3[k] = nullptr; }   // ([0] = the workspace's block, freed above)
  h->d_blur3[0] = nullptr; h->bset = 0;
  for (int k = 0; k < 4; k++) {   // (set 0 is the pair allocated with the workspace; h->d_cand / h->d_cellcount point at the set in use)
    if (h->d_cand2[k]) (void)hipFree(h->d_cand2[k]);
    if (h->d_cellcount2[k]) (void)hipFree(h->d_cellcount2[k]);
    h->d_cand2[k] = nullptr; h->d_cellcount2[k] = nullptr; h->octdone_valid[k] = false;
  }
  h->cset = 0;
  drop_chain_graphs(h);
  void* pinned[] = {h->h_kps, h->h_desc, h->h_nout, h->h_seq};
  for (void* p : pinned) if (p) (void)hipHostFree(p);
  h->h_seq = nullptr; h->d_ticket = nullptr;
  h->d_blurcols = nullptr; h->d_blurtab = nullptr;
  h->d_strips = nullptr; h->d_rgroups = nullptr; h->d_pyrtiles = nullptr;
  h->d_geom = nullptr; h->d_cells = nullptr; h->d_tiles = nullptr; h->d_xofs = h->d_alpha = h->d_yofs = h->d_beta = nullptr;
  h->d_pyr = h->d_blur = nullptr; h->d_cand = h->d_pts = h->d_lvlkp = nullptr;
  for (int k = 0; k < 4; k++) { h->d_lvlkp3[k] = nullptr; h->d_lvlcount3[k] = nullptr; }
  h->d_pyr_alt = nullptr; h->d_pyr_3rd = nullptr; h->d_pyr_4th = nullptr; h->out_gen = 0; h->pf_valid = false; h->next_hint = nullptr;
  h->d_nodeof = h->d_cellcount = h->d_celloff = h->d_candtotal = h->d_lvlcount = nullptr;
  h->d_kpsorted = nullptr; h->d_kpsortidx = nullptr; h->d_kpident = nullptr;
  h->d_kps = nullptr; h->d_desc = nullptr; h->d_nout = nullptr; h->h_kps = nullptr; h->h_desc = nullptr; h->h_nout = nullptr;
  h->rows = h->cols = 0;
}

// ORBextractor ctor tables.  NOTE the member `scaleFactor` is a double holding the float argument
// (ORBextractor.hpp:97), so products/quotients with it are formed in double and rounded to float once.
void build_ctor_tables(dvs_orb* h) {
  const int nl = h->prm.nlevels;
  const double scaleFactor = (double)h->prm.scale_factor;
  h->scale.assign(nl, 0.f); h->sigma2.assign(nl, 0.f); h->inv_scale.assign(nl, 0.f); h->inv_sigma2.assign(nl, 0.f);
  h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {
    h->scale[i] = (float)(h->scale[i - 1] * scaleFactor);
    h->sigma2[i] = h->scale[i] * h->scale[i];
  }
  for (int i = 0; i < nl; i++) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
  h->feat_per_level.assign(nl, 0);
  const float factor = (float)(1.0f / scaleFactor);
  float desired = h->prm.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    h->feat_per_level[l] = cv_round_f(desired);
    sum += h->feat_per_level[l];
    desired *= factor;
  }
  h->feat_per_level[nl - 1] = std::max(h->prm.nfeatures - sum, 0);
  // umax (ORBextractor.cpp:451-468)
  int v, v0;
  const int vmax = cv_floor_f(kHalfPatch * sqrtf(2.f) / 2 + 1);
  const int vmin = cv_ceil_f(kHalfPatch * sqrtf(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round_d(sqrt(hp2 - v * v));
  for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
    h->umax[v] = v0;
    ++v0;
  }
}

void level_size(const dvs_orb* h, int rows, int cols, int level, int& lr, int& lc) {
  const float s = h->inv_scale[level];  // ORBextractor.cpp:1173-1174
  lc = cv_round_f((float)cols * s);
  lr = cv_round_f((float)rows * s);
}

// cv::resize INTER_LINEAR coefficient tables for one axis (imgproc/resize.cpp, 8UC1 fixed-point path)
void build_axis_table(int ssize, int dsize, bool clamp_like_x, std::vector<int>& ofs, std::vector<int>& coef) {
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1. / inv_scale;
  for (int d = 0; d < dsize; d++) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (clamp_like_x) {
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= ssize - 1) { fx = 0; sx = ssize - 1; }
    }
    const short c0 = sat_short(cv_round_f((1.f - fx) * 2048));
    const short c1 = sat_short(cv_round_f(fx * 2048));
    ofs.push_back(sx);
    coef.push_back((int)(uint16_t)c0 | ((int)c1 << 16));
  }
}

dvs_status build_geometry(dvs_orb* h, int rows, int cols, Geom& G, std::vector<Cell>& cells, std::vector<BlurTile>& tiles,
                          std::vector<BlurStrip>& strips, std::vector<ResizeGroup>& rgroups, std::vector<PyrTile>& ptiles,
                          std::vector<int>& xofs, std::vector<int>& alpha, std::vector<int>& yofs, std::vector<int>& beta) {
  memset(&G, 0, sizeof(G));
  const int nl = h->prm.nlevels;
  G.nlevels = nl; G.rows = rows; G.cols = cols;
  // the streaming blur's band: a wavefront walks its rows one after the other, so a few frames get shorter bands (more wavefronts, 6 halo
  // rows each): 16 rows up to 2 frames per launch sequence, 32 up to 8, kBlurBand beyond (128 measured slower than 64 at 64 frames, round 3)
  G.blurBand = h->max_batch <= 2 ? 16 : (h->max_batch <= 8 ? 32 : kBlurBand);
  G.iniTh = std::min(std::max(h->prm.ini_th_fast, 0), 255);  // cv::FAST clamps the threshold
  G.minTh = std::min(std::max(h->prm.min_th_fast, 0), 255);
  G.outCap = h->prm.nfeatures + 3 * nl;
  memcpy(G.gk, h->prm.gauss_kernel, sizeof(G.gk));
  memcpy(G.umax, h->umax, sizeof(G.umax));
  for (int lane = 0; lane < 64; lane++) {
    const int row = lane >> 1, half = lane & 1, v = row - kHalfPatch;
    for (int i = 0; i < 16; i++) {
      const int u = half ? 1 + i : -kHalfPatch + i;
      const bool in = row <= 2 * kHalfPatch && std::abs(u) <= kHalfPatch && std::abs(u) <= h->umax[std::abs(v)];
      if (in) {
        G.icw[lane][i >> 2] |= (uint32_t)(u + kHalfPatch) << (8 * (i & 3));
        G.icw[lane][4 + (i >> 2)] |= 1u << (8 * (i & 3));
      }
    }
  }
  uint64_t off = 0, candOff = 0, ptsOff = 0;
  int kpOff = 0;
  for (int l = 0; l < nl; l++) {
    LevelGeom& L = G.lv[l];
    level_size(h, rows, cols, l, L.h, L.w);
    // ComputeKeyPointsOctTree cell grid (ORBextractor.cpp:789-803)
    const int minBX = kMinBorder, minBY = kMinBorder;
    const int maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    if (width < 1 || height < 1) { set_error("level %d (%dx%d) is smaller than the 16-px borders", l, L.w, L.h); return DVS_ERR_UNSUPPORTED; }
    const int nCols = (int)(width / 35.f), nRows = (int)(height / 35.f);
    const int nIni = (int)round((double)(width / height));  // round(static_cast<float>(w)/(h)) (:559)
    if (nCols < 1 || nRows < 1 || nIni < 1) {
      set_error("level %d (%dx%d): the reference divides by zero here (nCols=%d nRows=%d nIni=%d)", l, L.w, L.h, nCols, nRows, nIni);
      return DVS_ERR_UNSUPPORTED;
    }
    L.nCols = nCols; L.nRows = nRows;
    L.wCell = (int)ceilf(width / nCols);
    L.hCell = (int)ceilf(height / nRows);
    L.regionW = maxBX - minBX; L.regionH = maxBY - minBY;
    L.nIni = nIni;
    L.hX = width / nIni;
    L.N = h->feat_per_level[l];
    L.scale = h->scale[l];
    L.kpSize = (float)(int)(31 * h->scale[l]);  // PATCH_SIZE*mvScaleFactor[level] truncated (:880)
    if (L.wCell + 6 > kMaxCellDim || L.hCell + 6 > kMaxCellDim || L.regionW > 4095 || L.regionH > 4095 || L.N > kMaxQuota) {
      set_error("level %d: cell %dx%d / region %dx%d / quota %d exceeds the supported limits", l, L.wCell, L.hCell, L.regionW, L.regionH, L.N);
      return DVS_ERR_UNSUPPORTED;
    }
    L.pitch = (int)align_up(L.w + 8, 64);  // >= 8 spare columns: k_resize4 writes the REFLECT_101 continuation there
    L.off = off;
    off += align_up((uint64_t)L.pitch * L.h, 256);
    L.cellBase = (int)cells.size();
    L.cellCap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
    int slot = 0;
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBY + i * L.hCell);
      float maxY = iniY + L.hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBX + j * L.wCell);
        float maxX = iniX + L.wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        Cell c;
        c.level = (int16_t)l; c.i = (int16_t)i; c.j = (int16_t)j;
        c.x0 = (int16_t)(int)iniX; c.y0 = (int16_t)(int)iniY;
        c.cw = (int16_t)((int)maxX - (int)iniX); c.ch = (int16_t)((int)maxY - (int)iniY);
        {  // k_fast_wave's lane mapping of the rejection loop (same arithmetic as the kernel's: tile origin, interior column groups)
          const int ox = h->fast_byte_dma ? 1 : (c.x0 & 3), cx0 = ox + 3, cx1 = ox + c.cw - 3;
          const int ng = std::max(((cx1 - 1) >> 2) - (cx0 >> 2) + 1, 1);
          const int rpt = 64 / ng, ih = c.ch - 6;
          const int tail = ih > 0 ? ih - (ih - 1) / rpt * rpt : 0;   // rows of the loop's last trip
          c.rpt = (int16_t)(rpt | (tail << 8)); c.inv_ng = 1.0f / (float)ng;
        }
        c.slot = slot++;
        cells.push_back(c);
      }
    }
    L.nCells = slot;
    L.ptsCap = L.nCells * L.cellCap;
    if (L.ptsCap >= (1 << 24)) { set_error("level %d: candidate capacity too large", l); return DVS_ERR_UNSUPPORTED; }
    L.candOff = candOff; candOff += (uint64_t)L.ptsCap;
    L.ptsOff = ptsOff; ptsOff += (uint64_t)L.ptsCap;
    L.kpOff = kpOff; kpOff += L.N + 4;
    G.maxN = std::max(G.maxN, std::max(L.N + 3, 4 * nIni));
    for (int ty = 0; ty < (L.h + 15) / 16; ty++)
      for (int tx = 0; tx < (L.w + 63) / 64; tx++) tiles.push_back(BlurTile{(int16_t)l, (int16_t)tx, (int16_t)ty, 0});
    {  // streaming blur: equal-width strips of <= 248 columns (multiples of 4), bands of kBlurBand rows
      const int ns = (L.w + 247) / 248;  // 62 output lanes + 2 halo lanes per wavefront
      const int sw = ((L.w + ns - 1) / ns + 3) / 4 * 4;
      for (int y0 = 0; y0 < L.h; y0 += G.blurBand)
        for (int x0 = 0; x0 < L.w; x0 += sw) strips.push_back(BlurStrip{(int16_t)l, (int16_t)x0, (int16_t)std::min(sw, L.w - x0), (int16_t)y0});
    }
    if (l > 0) {
      L.xtab = (int)xofs.size(); L.ytab = (int)yofs.size();
      build_axis_table(G.lv[l - 1].w, L.w, true, xofs, alpha);
      build_axis_table(G.lv[l - 1].h, L.h, false, yofs, beta);
      // 4-column groups for k_resize4 (valid while the four left taps span <= 8 bytes, i.e. scale factor <= 2)
      L.gtab = (int)rgroups.size();
      bool fits = true;
      for (int x4 = 0; x4 < L.w + 8; x4 += 4) {  // 2 extra groups: columns >= w mirror column 2w-2-x (blur border)
        ResizeGroup rg{};
        int cols4[4], mn = 1 << 30;
        for (int i = 0; i < 4; i++) {
          int x = x4 + i;
          if (x >= L.w) x = std::max(0, 2 * L.w - 2 - x);
          cols4[i] = x;
          mn = std::min(mn, xofs[L.xtab + x]);
        }
        rg.base = mn & ~3;
        rg.shift = (uint32_t)(mn & 3) * 8;
        for (int i = 0; i < 4; i++) {
          const int o = xofs[L.xtab + cols4[i]] - mn;  // both taps (o, o + 1) must lie inside the 8-byte window
          if (o < 0 || o > 6) fits = false;
          rg.sel[i] = (uint32_t)o | (0x0cu << 8) | ((uint32_t)(o + 1) << 16) | (0x0cu << 24);
          rg.alpha[i] = alpha[L.xtab + cols4[i]];
        }
        rgroups.push_back(rg);
      }
      if (!fits) L.gtab = -1;
    }
  }
  // ---- pyramid cascade tiles (k_pyr_cascade) --------------------------------------------------------------------
  if (nl >= 2) {
    auto srcx = [&](int k, int x) { return xofs[G.lv[k].xtab + x]; };                                    // left tap (already clamped)
    auto srcy = [&](int k, int y) { return std::min(std::max(yofs[G.lv[k].ytab + y], 0), G.lv[k - 1].h - 1); };  // top tap
    int maxBytes = 0;
    bool cascade_ok = true;
    // tile = the level-1 pixels a workgroup owns (results do not depend on the tiling).  The cascade's latency is its seven dependent levels
    // plus a fixed part per workgroup (tile record -> tables -> staging), so the tile follows the batch the handle is built for: a
    // one-frame handle (a lane of dvs_pipeline at 1 frame per step) takes 64 x 16 — 32 -> 22 us for the launch, 18.6 -> 20.1 k frames/s
    // in the three-lane schedule — and everything else 128 x 64 (64 x 16 at 2 frames per step: 34.7 -> 31.7 k, at 8: 45.8 -> 91.7 us;
    // 512 / 1024 threads per workgroup with 256 x 64 tiles: +1 % at 4 and 8 frames, not kept).  DVS_CASC_TW / DVS_CASC_TH override
    // (tools/time_cascade_variants.py).  EXPERIMENTS.md, round 4.
    const int cascT = 256;   // threads per workgroup of k_pyr_cascade
    const int tileW = h->max_batch <= 1 ? 64 : kPyrTileW, tileH = h->max_batch <= 1 ? 16 : kPyrTileH;
    for (int ty = 0; ty < G.lv[1].h; ty += tileH)
      for (int tx = 0; tx < G.lv[1].w; tx += tileW) {
        PyrTile T{};
        int oxa[DVS_MAX_LEVELS], oxb[DVS_MAX_LEVELS], oya[DVS_MAX_LEVELS], oyb[DVS_MAX_LEVELS];
        oxa[1] = tx; oxb[1] = std::min(tx + tileW, G.lv[1].w);
        oya[1] = ty; oyb[1] = std::min(ty + tileH, G.lv[1].h);
        for (int k = 2; k < nl; k++) {  // owner of a pixel = owner of its top-left source tap
          int a = 0, b;
          while (a < G.lv[k].w && srcx(k, a) < oxa[k - 1]) a++;
          b = a;
          while (b < G.lv[k].w && srcx(k, b) < oxb[k - 1]) b++;
          oxa[k] = a; oxb[k] = b;
          a = 0;
          while (a < G.lv[k].h && srcy(k, a) < oya[k - 1]) a++;
          b = a;
          while (b < G.lv[k].h && srcy(k, b) < oyb[k - 1]) b++;
          oya[k] = a; oyb[k] = b;
        }
        int cxa = 0, cxb = 0, cya = 0, cyb = 0;  // compute region of the level below the current one (empty)
        for (int k = nl - 1; k >= 1; k--) {
          const bool own = oxb[k] > oxa[k] && oyb[k] > oya[k];
          int xa = own ? oxa[k] : 0, xb = own ? oxb[k] : 0, ya = own ? oya[k] : 0, yb = own ? oyb[k] : 0;
          if (cxb > cxa && cyb > cya) {  // what level k+1's region reads from level k
            const int nxa = srcx(k + 1, cxa), nxb = std::min(srcx(k + 1, cxb - 1) + 1, G.lv[k].w - 1) + 1;
            const int nya = srcy(k + 1, cya), nyb = std::min(srcy(k + 1, cyb - 1) + 1, G.lv[k].h - 1) + 1;
            if (own) { xa = std::min(xa, nxa); xb = std::max(xb, nxb); ya = std::min(ya, nya); yb = std::max(yb, nyb); }
            else { xa = nxa; xb = nxb; ya = nya; yb = nyb; }
          }
          xa &= ~3;  // dword-aligned groups in LDS and in HBM
          PyrTileLevel& R = T.lv[k];
          R.cx0 = (int16_t)xa; R.cx1 = (int16_t)xb; R.cy0 = (int16_t)ya; R.cy1 = (int16_t)yb;
          R.ox0 = (int16_t)(own ? oxa[k] : 0); R.ox1 = (int16_t)(own ? oxb[k] : 0);
          R.oy0 = (int16_t)(own ? oya[k] : 0); R.oy1 = (int16_t)(own ? oyb[k] : 0);
          cxa = xa; cxb = xb; cya = ya; cyb = yb;
          maxBytes = std::max(maxBytes, (((xb - xa) + 3) & ~3) * (yb - ya));
        }
        // level-0 source region of the level-1 compute region
        T.sx0 = (int16_t)(srcx(1, cxa) & ~3);
        T.sx1 = (int16_t)(std::min(srcx(1, cxb - 1) + 1, G.lv[0].w - 1) + 1);
        T.sy0 = (int16_t)srcy(1, cya);
        T.sy1 = (int16_t)(std::min(srcy(1, cyb - 1) + 1, G.lv[0].h - 1) + 1);
        maxBytes = std::max(maxBytes, (((T.sx1 - T.sx0) + 3) & ~3) * (T.sy1 - T.sy0));
        int tabx = 0, taby = 0;
        for (int k = 1; k < nl; k++) {
          const int cwp = ((T.lv[k].cx1 - T.lv[k].cx0) + 3) & ~3, chh = T.lv[k].cy1 - T.lv[k].cy0;
          tabx += cwp; taby += chh;
          if (cwp > cascT || chh > cascT) cascade_ok = false;   // k_pyr_cascade: one table entry per thread and level
        }
        if (tabx > kPyrTabX || taby > kPyrTabY) cascade_ok = false;
        ptiles.push_back(T);
      }
    G.pyrTiles = cascade_ok ? (int)ptiles.size() : 0;
    G.pyrLds = (int)align_up(maxBytes, 16);
  }
  G.frameBytes = off;
  G.candPerFrame = candOff; G.ptsPerFrame = ptsOff;
  G.totalCells = (int)cells.size();
  G.kpBlock = kpOff;
  G.blurTiles = (int)tiles.size();
  G.blurStrips = (int)strips.size();
  {  // wave-per-cell FAST: LDS tile geometry (pitch keeps the <= 3 byte phase of the aligned staging)
    int maxw = 0, maxh = 0;
    for (const Cell& c : cells) { maxw = std::max<int>(maxw, c.cw); maxh = std::max<int>(maxh, c.ch); }
    const int need = std::max(maxw + 3, maxh - 4);   // tile pitch: the widest cell at any byte phase; tile rows: <= pitch + 4 (fast_tile_bytes)
    G.fastP = need <= 48 ? 48 : (need <= 64 ? 64 : 80);
    G.fastRows = maxh;
    const int listBytes = (int)align_up(2 * (size_t)std::max(1, (maxw - 6) * (maxh - 6)), 16);   // worst case: every interior pixel survives
    G.fastByteDma = h->fast_byte_dma;
    G.fastTile = fast_tile_bytes(G.fastP);
    // tile (fast_tile_bytes) + score tile (rows of the tallest cell) + work list
    G.fastWaveLds = (int)align_up((size_t)G.fastTile + (size_t)G.fastRows * G.fastP + listBytes, 16);
  }
  return DVS_OK;
}

template <class T>
dvs_status upload(T** dptr, const std::vector<T>& v) {
  DVS_HIP(hipMalloc((void**)dptr, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) DVS_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return DVS_OK;
}

// k_blur_mfma: work items (level, 32-column strip, band of <= kBlurMfmaBand row tiles) and the int8 operand fragments in lane order
// (lane = 32 half + (index & 31) holds k = 16 half .. 16 half + 15):  per strip the two horizontal band matrices
// B[k][n] = sum_t gk[t] [reflect101(c0 + n + t - 3) == cin0 + k]  for source columns cin0 = c0 - 16 and c0 + 16, and once the two
// vertical band matrices with k in accumulator order (h-row (k & 3) + 8 ((k & 15) >> 2) + 4 half of the carried / the next block).
static bool build_blur_mfma(const Geom& G, std::vector<BlurCol>& items, std::vector<uint4>& tab, int& avt) {
  int sum = 0;
  for (int t = 0; t < 7; t++) { if (G.gk[t] < 0 || G.gk[t] > 127) return false; sum += G.gk[t]; }
  if (sum != 256) return false;
  auto refl = [](int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p; return p; };
  int nstrip = 0;
  for (int l = 0; l < G.nlevels; l++) {
    const LevelGeom& L = G.lv[l];
    const int SS = (L.w + 127) / 128, S = 4 * SS;   // strips of whole super-strips: the ones past the width get all-zero weights
    const int T = (L.h + 31) / 32, bands = (T + kBlurMfmaBand - 1) / kBlurMfmaBand, nt = (T + bands - 1) / bands;
    for (int sI = 0; sI < S; sI++) {
      const int c0 = 32 * sI;
      for (int mtx = 0; mtx < 2; mtx++) {
        const int cin0 = mtx == 0 ? c0 - 16 : c0 + 16;
        for (int lane = 0; lane < 64; lane++) {
          const int n = lane & 31, half = lane >> 5;
          uint8_t b[16];
          for (int k = 0; k < 16; k++) {
            const int incol = cin0 + 16 * half + k;
            int w = 0;
            if (c0 + n < L.w && incol >= 0 && incol < L.w)
              for (int t = 0; t < 7; t++) if (refl(c0 + n + t - 3, L.w) == incol) w += G.gk[t];
            b[k] = (uint8_t)w;   // <= 2 * 56 at a border: still int8
            if (w > 127) return false;
          }
          uint4 v; memcpy(&v, b, 16); tab.push_back(v);
        }
      }
    }
    for (int t0 = 0; t0 < T; t0 += nt)
      for (int ss = 0; ss < SS; ss++) items.push_back(BlurCol{(int16_t)l, (int16_t)ss, (int16_t)t0, (int16_t)std::min(nt, T - t0), nstrip + 4 * ss});
    nstrip += S;
  }
  avt = nstrip;
  for (int mtx = 0; mtx < 2; mtx++)
    for (int lane = 0; lane < 64; lane++) {
      const int i = lane & 31, half = lane >> 5;
      uint8_t b[16];
      for (int k = 0; k < 16; k++) {
        const int rho = (k & 3) + 8 * (k >> 2) + 4 * half, tau = (mtx ? 32 : 0) + rho - i;
        b[k] = (uint8_t)(tau >= 0 && tau <= 6 ? G.gk[tau] : 0);
      }
      uint4 v; memcpy(&v, b, 16); tab.push_back(v);
    }
  return true;
}

// one probe per process and device (see k_probe_lds_dma)
static int probe_byte_dma(int device, hipStream_t st) {
  static int cache[64]; static bool done[64];
  if (device < 0 || device >= 64) return 0;
  if (done[device]) return cache[device];
  uint8_t hsrc[512]; uint32_t hout[64];
  for (int i = 0; i < 512; i++) hsrc[i] = (uint8_t)(i * 7 + 3);
  uint8_t* d = nullptr; uint32_t* o = nullptr;
  int ok = 1;
  if (hipMalloc((void**)&d, 512) != hipSuccess || hipMalloc((void**)&o, 256) != hipSuccess) ok = 0;
  if (ok && hipMemcpy(d, hsrc, 512, hipMemcpyHostToDevice) != hipSuccess) ok = 0;
  for (int shift = 1; ok && shift < 4; shift++) {
    hipLaunchKernelGGL(k_probe_lds_dma, dim3(1), dim3(64), 0, st, d, o, shift);
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(hout, o, 256, hipMemcpyDeviceToHost) != hipSuccess) { ok = 0; break; }
    for (int l = 0; l < 64; l++) { uint32_t e; memcpy(&e, hsrc + shift + 4 * l, 4); if (e != hout[l]) ok = 0; }
  }
  if (d) (void)hipFree(d);
  if (o) (void)hipFree(o);
  cache[device] = ok; done[device] = true;
  return ok;
}

dvs_status ensure_workspace(dvs_orb* h, int rows, int cols) {
  if (h->rows == rows && h->cols == cols && h->d_geom) return DVS_OK;
  DVS_HIP(hipStreamSynchronize(h->stream));
  if (h->aux_stream) DVS_HIP(hipStreamSynchronize(h->aux_stream));
  if (h->pf_stream) DVS_HIP(hipStreamSynchronize(h->pf_stream));
  if (h->out_pending) DVS_HIP(hipEventSynchronize(h->ev_out));   // a deferred descriptor stage on the tail stream
  h->out_pending = false; h->pf_joined = false; h->pf_valid = false;
  free_workspace(h);
  Geom G;
  std::vector<Cell> cells; std::vector<BlurTile> tiles; std::vector<BlurStrip> strips; std::vector<ResizeGroup> rgroups;
  std::vector<PyrTile> ptiles;
  std::vector<int> xofs, alpha, yofs, beta;
  DVS_TRY(build_geometry(h, rows, cols, G, cells, tiles, strips, rgroups, ptiles, xofs, alpha, yofs, beta));
  h->geom = G;
  const size_t B = (size_t)h->max_batch;
  DVS_HIP(hipMalloc((void**)&h->d_geom, sizeof(Geom)));
  DVS_HIP(hipMemcpy(h->d_geom, &G, sizeof(Geom), hipMemcpyHostToDevice));
  DVS_TRY(upload(&h->d_cells, cells));
  DVS_TRY(upload(&h->d_tiles, tiles));
  DVS_TRY(upload(&h->d_strips, strips));
  {
    std::vector<BlurCol> bcols; std::vector<uint4> btab;
    h->blur_mfma_ok = build_blur_mfma(G, bcols, btab, h->blur_avt);
    if (h->blur_mfma_ok) { DVS_TRY(upload(&h->d_blurcols, bcols)); DVS_TRY(upload(&h->d_blurtab, btab)); h->n_blurcols = (int)bcols.size(); }
  }
  DVS_TRY(upload(&h->d_rgroups, rgroups));
  DVS_TRY(upload(&h->d_pyrtiles, ptiles));
  if (G.pyrLds > 0) DVS_HIP(hipFuncSetAttribute((const void*)k_pyr_cascade<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G.pyrLds));
  DVS_TRY(upload(&h->d_xofs, xofs)); DVS_TRY(upload(&h->d_alpha, alpha));
  DVS_TRY(upload(&h->d_yofs, yofs)); DVS_TRY(upload(&h->d_beta, beta));
  DVS_HIP(hipMalloc((void**)&h->d_pyr, B * G.frameBytes + 256));   // + slack: k_resize4 reads whole 12-byte windows at a row's end
  DVS_HIP(hipMalloc((void**)&h->d_blur, B * G.frameBytes));
  DVS_HIP(hipMalloc((void**)&h->d_cand, B * G.candPerFrame * 4));
  DVS_HIP(hipMalloc((void**)&h->d_pts, B * G.ptsPerFrame * 4));
  DVS_HIP(hipMalloc((void**)&h->d_nodeof, B * G.ptsPerFrame * 4));
  DVS_HIP(hipMalloc((void**)&h->d_cellcount, B * G.totalCells * 4));
  DVS_HIP(hipMalloc((void**)&h->d_celloff, B * G.totalCells * 4));
  DVS_HIP(hipMalloc((void**)&h->d_candtotal, B * G.nlevels * 4));
  for (int k = 0; k < 3; k++) {
    DVS_HIP(hipMalloc((void**)&h->d_lvlcount3[k], B * G.nlevels * 4));
    DVS_HIP(hipMalloc((void**)&h->d_lvlkp3[k], B * (size_t)G.kpBlock * 4));
  }
  h->lset = 0; h->d_lvlcount = h->d_lvlcount3[0]; h->d_lvlkp = h->d_lvlkp3[0];
  DVS_HIP(hipMalloc((void**)&h->d_kpsorted, B * (size_t)G.kpBlock * 4));
  DVS_HIP(hipMalloc((void**)&h->d_kpsortidx, B * (size_t)G.kpBlock * 4));
  {
    std::vector<uint32_t> ident(B * (size_t)G.kpBlock);
    for (size_t f = 0; f < B; f++)
      for (int l = 0; l < G.nlevels; l++) {
        const int end = l + 1 < G.nlevels ? G.lv[l + 1].kpOff : G.kpBlock;
        for (int sl = G.lv[l].kpOff; sl < end; sl++) ident[f * G.kpBlock + sl] = (uint32_t)(sl - G.lv[l].kpOff);
      }
    DVS_TRY(upload(&h->d_kpident, ident));
  }
  DVS_HIP(hipMalloc((void**)&h->d_kps, B * (size_t)G.outCap * sizeof(dvs_keypoint)));
  DVS_HIP(hipMalloc((void**)&h->d_desc, B * (size_t)G.outCap * 32));
  DVS_HIP(hipMalloc((void**)&h->d_nout, B * 4));
  DVS_HIP(hipHostMalloc((void**)&h->h_kps, B * (size_t)G.outCap * sizeof(dvs_keypoint)));
  DVS_HIP(hipHostMalloc((void**)&h->h_desc, B * (size_t)G.outCap * 32));
  DVS_HIP(hipHostMalloc((void**)&h->h_nout, B * 4));
  if (!h->h_seq) {
    DVS_HIP(hipHostMalloc((void**)&h->h_seq, 64));
    *h->h_seq = 0; h->export_seq = 0;
    DVS_HIP(hipMalloc((void**)&h->d_ticket, 4));
    DVS_HIP(hipMemset(h->d_ticket, 0, 4));
  }
  h->octree_nmax = G.maxN + 8;
  {
    int maxPts = 0;
    for (int l = 0; l < G.nlevels; l++) maxPts = std::max(maxPts, G.lv[l].ptsCap);
    h->octree_ptscap = std::min(6144, maxPts);
    // two quad-tree workgroups per CU (the launch is latency-bound: frames x levels workgroups, all resident at once) need
    // <= 80 KB each including k_octree's static LDS; give up a few point slots rather than half the residency
    const long fixed = (long)h->octree_nmax * (long)(2 * sizeof(QNode) + 8 + 16 + 4 * 4) + 2048;
    const long fit = (80 * 1024 - fixed) / 8;
    if (fit >= 4096 && h->octree_ptscap > fit) h->octree_ptscap = (int)fit;
  }
  h->octree_smem = (size_t)h->octree_nmax * (2 * sizeof(QNode) + 8 + 16 + 4 * 4) + (size_t)h->octree_ptscap * 8;
  DVS_HIP(hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->octree_smem));
  DVS_HIP(hipFuncSetAttribute((const void*)k_octree_blur, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->octree_smem));
  h->d_cand2[0] = h->d_cand; h->d_cellcount2[0] = h->d_cellcount; h->cset = 0;
  h->d_blur3[0] = h->d_blur; h->bset = 0;
  {
    // level classes for graded launches ({L0, L1}, {L2, L3}, {L4 ..}): node capacity from the class's largest quota, point capacity =
    // level 0's scaled by the level's share of pixels (a level whose candidates exceed it keeps them in HBM — slower, never wrong)
    const long per_node = (long)(2 * sizeof(QNode) + 8 + 16 + 4 * 4);
    h->oct_ncls = G.nlevels >= 6 ? 3 : (G.nlevels >= 4 ? 2 : 0);
    h->oct_l0[0] = 0; h->oct_l0[1] = 2; h->oct_l0[2] = h->oct_ncls == 3 ? 4 : G.nlevels; h->oct_l0[3] = G.nlevels;
    for (int c = 0; c < h->oct_ncls; c++) {
      const int l0 = h->oct_l0[c], l1 = h->oct_l0[c + 1];
      int nmax = 0; long pts = 0;
      for (int l = l0; l < l1; l++) {
        nmax = std::max(nmax, std::max(G.lv[l].N + 3, 4 * G.lv[l].nIni) + 8);
        const double share = (double)G.lv[l].w * G.lv[l].h / ((double)G.lv[0].w * G.lv[0].h);
        pts = std::max(pts, std::min<long>(G.lv[l].ptsCap, (long)(share * h->octree_ptscap) + 64));
      }
      h->oct_nmax_cls[c] = nmax;
      h->oct_ptscap_cls[c] = (int)std::min<long>(pts, h->octree_ptscap);
      h->oct_smem_cls[c] = (size_t)nmax * per_node + (size_t)h->oct_ptscap_cls[c] * 8;
      if (h->oct_smem_cls[c] > h->octree_smem) { h->oct_smem_cls[c] = h->octree_smem; h->oct_nmax_cls[c] = h->octree_nmax; h->oct_ptscap_cls[c] = h->octree_ptscap; }
    }
  }
  DVS_HIP(hipFuncSetAttribute((const void*)k_fast_wave<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G.fastWaveLds));
  DVS_HIP(hipFuncSetAttribute((const void*)k_fast_wave<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G.fastWaveLds));
  DVS_HIP(hipFuncSetAttribute((const void*)k_fast_wave<80>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G.fastWaveLds));
  h->rows = rows; h->cols = cols;
  return DVS_OK;
}

// the level chain: level l from level l-1 (ORBextractor.cpp:1171-1192) into `pyr`, one launch per level on `pst`
dvs_status launch_pyramid_chain(dvs_orb* h, const ImgSrc& src, int nimg, u8* pyr, hipStream_t pst, bool level_events, int top_level = -1) {
  const Geom& G = h->geom;
  const int last = top_level < 0 ? G.nlevels - 1 : std::min(top_level, G.nlevels - 1);   // level sharding: only up to the highest level owned
  for (int l = 1; l <= last; l++) {
    const LevelGeom& S = G.lv[l - 1];
    const LevelGeom& D = G.lv[l];
    const u8* sp = l == 1 ? src.img0 : pyr + S.off;
    const uint64_t sfs = l == 1 ? src.fstride0 : G.frameBytes;
    const int spitch = l == 1 ? (int)src.step0 : S.pitch;
    dim3 grid((D.w + 8 + 255) / 256, (D.h + 3) / 4, nimg), block(64, 4, 1);
    const bool aligned = (((uintptr_t)sp) | sfs | (uint64_t)spitch) % 4 == 0;
    if (aligned && D.gtab >= 0) {
      // the x axis runs over the 4-pixel groups of fpg consecutive frames (k_resize4): whole multiples of 8 frames are dealt to the XCDs
      // as 8 groups; the frames of a group must stay within 32-bit offsets of its first one
      const int dwp = (D.w + 8) & ~3, ngx = dwp / 4;
      int fpg = nimg >= 16 && nimg % 8 == 0 ? nimg / 8 : 1;
      if ((uint64_t)fpg * std::max<uint64_t>(sfs, G.frameBytes) >= (1ull << 31)) fpg = 1;
      const int ngroups = nimg / fpg;
      const uint32_t tx = (uint32_t)(fpg * ngx + 63) / 64, ty = (D.h + 4 * kResizeRows - 1) / (4 * kResizeRows), tiles = tx * ty;
      // n / d == umulhi(n, 2^32 / d + 1) for every n with n * d < 2^32
      auto magic = [](uint64_t nmax, uint32_t d) -> uint32_t { return d > 1 && nmax * d < (1ull << 32) ? (uint32_t)((1ull << 32) / d + 1) : 0u; };
      hipLaunchKernelGGL(k_resize4, dim3(tiles, ngroups), block, 0, pst,
                         sp, sfs, S.w, S.h, spitch, pyr + D.off, G.frameBytes, dwp, D.h, D.pitch,
                         h->d_rgroups + D.gtab, h->d_yofs + D.ytab, h->d_beta + D.ytab,
                         l == 1 && sp != h->d_pyr + G.lv[0].off && sp != h->d_pyr_alt + G.lv[0].off && sp != h->d_pyr_3rd + G.lv[0].off && sp != h->d_pyr_4th + G.lv[0].off ? nimg - 1 : -1,   // caller's buffer: no slack behind its last row
                         (int)tx, magic((uint64_t)tiles * ngroups, tiles), magic(tiles, tx), ngx, fpg, magic((uint64_t)tx * 64, (uint32_t)ngx), nimg);
    }
    else
      hipLaunchKernelGGL(k_resize, grid, block, 0, pst, sp, sfs, S.w, S.h, spitch, pyr + D.off, G.frameBytes, D.w, D.h,
                         D.pitch, h->d_xofs + D.xtab, h->d_alpha + D.xtab, h->d_yofs + D.ytab, h->d_beta + D.ytab);
    if (level_events) DVS_HIP(hipEventRecord(h->ev_level[l], pst));
  }
  return DVS_OK;
}

// ---- the stages of one extraction, each a launch on the stream the schedule (enqueue_extract) picks --------------------------------

// FAST on cells [c0, c1) of the level-major cell table
void launch_fast(dvs_orb* h, const ImgSrc& src, int nimg, hipStream_t fs, int c0, int c1) {
  if (c1 <= c0) return;
  const Geom& G = h->geom;
  if ((((uintptr_t)src.img0) | src.step0 | src.fstride0) % 4 != 0) {   // rows not dword aligned: the generic workgroup-per-cell kernel
    hipLaunchKernelGGL(k_fast_cell, dim3(c1 - c0, nimg), dim3(256), 0, fs, h->d_geom, h->d_cells, src, h->d_cand, h->d_cellcount, c0);
    return;
  }
  const dim3 grid((c1 - c0 + 3) / 4, nimg);
  const size_t lds = 4 * (size_t)G.fastWaveLds;
  // n / grid.x == umulhi(n, 2^32 / grid.x + 1) for every workgroup id n of this grid (n * grid.x < 2^32), else 0: the kernel divides
  const uint32_t magic = grid.x > 1 && (uint64_t)grid.x * grid.x * nimg < (1ull << 32) ? (uint32_t)((1ull << 32) / grid.x + 1) : 0u;
  if (G.fastP == 48) hipLaunchKernelGGL(k_fast_wave<48>, grid, dim3(256), lds, fs, h->d_geom, h->d_cells, src, h->d_cand, h->d_cellcount, c0, c1, magic);
  else if (G.fastP == 64) hipLaunchKernelGGL(k_fast_wave<64>, grid, dim3(256), lds, fs, h->d_geom, h->d_cells, src, h->d_cand, h->d_cellcount, c0, c1, magic);
  else hipLaunchKernelGGL(k_fast_wave<80>, grid, dim3(256), lds, fs, h->d_geom, h->d_cells, src, h->d_cand, h->d_cellcount, c0, c1, magic);
}

// The announced next batch's level chain on pf_stream into d_pyr_alt, beside THIS batch's FAST: the chain is latency-bound, FAST is
// VALU-bound and insensitive to its cache traffic (beside the fetch-bound descriptor stage the chain doubled that stage's time), and
// with its pyramid built ahead the next call launches FAST on all levels at once.  `pend`: the previous call's deferred descriptor
// stage still reads that batch's pyramid (d_pyr_alt after the caller's swap).
dvs_status launch_prefetch(dvs_orb* h, const ImgSrc& src, int nimg, const u8* next_img0, bool pend) {
  const Geom& G = h->geom;
  if (!h->d_pyr_alt) DVS_HIP(hipMalloc((void**)&h->d_pyr_alt, (size_t)h->max_batch * G.frameBytes + 256));
  if (pend) {
    // build into the THIRD buffer instead of waiting for that stage — the chain then runs beside FAST from the start, as without
    // deferral (when it waited it ended after FAST and became the critical path).  The third buffer held the pyramid of two batches
    // ago; its last reader is that batch's descriptor stage, whose event is the only gate (nothing on the main stream).
    if (!h->d_pyr_3rd) DVS_HIP(hipMalloc((void**)&h->d_pyr_3rd, (size_t)h->max_batch * G.frameBytes + 256));
    if (h->ring == 4) {   // the oldest of four: alt <- 3rd <- 4th <- alt
      if (!h->d_pyr_4th) DVS_HIP(hipMalloc((void**)&h->d_pyr_4th, (size_t)h->max_batch * G.frameBytes + 256));
      u8* a = h->d_pyr_alt; h->d_pyr_alt = h->d_pyr_3rd; h->d_pyr_3rd = h->d_pyr_4th; h->d_pyr_4th = a;
    } else {
      std::swap(h->d_pyr_alt, h->d_pyr_3rd);
    }
    // its last reader: the descriptor stage of ring - 1 batches ago (the slot the NEXT deferred stage records into)
    if (h->out_gen >= h->ring - 1) DVS_HIP(hipStreamWaitEvent(h->pf_stream, h->ev_outs[h->out_gen % (h->ring - 1)], 0));
  } else {
    // d_pyr_alt's last readers are the previous call's kernels: its end-of-call event if it left one (no extra record), else this
    // point of the main stream
    hipEvent_t gate = h->gate_event;
    if (!gate) { gate = h->ev_chain_gate; DVS_HIP(hipEventRecord(gate, h->stream)); }
    DVS_HIP(hipStreamWaitEvent(h->pf_stream, gate, 0));
  }
  // asynchronous quad-trees: the FAST that follows this chain writes candidate set cset + 1, last read by the tree of two calls ago
  // (after `ring` asynchronous calls in a row the wait above — the descriptor stage of that very batch, which followed its tree — covers it)
  if (h->async_oct && h->octdone_valid[(h->cset + 1) % h->ring] && !(pend && h->async_run >= h->ring))
    DVS_HIP(hipStreamWaitEvent(h->pf_stream, h->ev_octdone[(h->cset + 1) % h->ring], 0));
  ImgSrc nsrc = src;
  nsrc.img0 = next_img0;
  h->timer.begin(DVS_STAGE_PYRAMID, h->pf_stream);  // the pyramid stage of the overlapped schedule IS this prefetch chain
  // (the seven launches of the chain also for few frames: all levels in one launch — k_pyr_cascade — on this stream was measured at 6 / 8
  // frames per step: 0.103 / 0.119 ms against 0.092 / 0.103, its LDS tiles take FAST's workgroup slots; EXPERIMENTS.md)
  const bool want_graph = h->env_chain_graph >= 0 ? h->env_chain_graph == 1 : nimg <= 12;
  if (!want_graph) {
    DVS_TRY(launch_pyramid_chain(h, nsrc, nimg, h->d_pyr_alt, h->pf_stream, false));
  } else {
    dvs_orb::ChainGraph* cg = nullptr;
    for (auto& c : h->chain_graphs)
      if (c.img0 == next_img0 && c.step0 == src.step0 && c.fstride0 == src.fstride0 && c.nimg == nimg && c.pyr == h->d_pyr_alt) { cg = &c; break; }
    if (cg && cg->exec) {
      DVS_HIP(hipGraphLaunch(cg->exec, h->pf_stream));
      h->chain_graph_launches++;
    } else if (cg) {   // second time: capture (nothing runs), instantiate, launch
      DVS_HIP(hipStreamBeginCapture(h->pf_stream, hipStreamCaptureModeThreadLocal));
      const dvs_status cs = launch_pyramid_chain(h, nsrc, nimg, h->d_pyr_alt, h->pf_stream, false);
      const hipError_t ce = hipStreamEndCapture(h->pf_stream, &cg->graph);
      if (cs != DVS_OK) return cs;
      DVS_HIP(ce);
      DVS_HIP(hipGraphInstantiate(&cg->exec, cg->graph, nullptr, nullptr, 0));
      DVS_HIP(hipGraphLaunch(cg->exec, h->pf_stream));
      h->chain_graph_launches++;
    } else {           // first time: plain launches, remember the argument set
      if (h->chain_graphs.size() >= 48) {   // a caller whose buffers never come back: stay bounded (rare: let launched graphs finish first)
        DVS_HIP(hipStreamSynchronize(h->pf_stream));
        drop_chain_graphs(h);
      }
      h->chain_graphs.push_back({next_img0, src.step0, src.fstride0, nimg, h->d_pyr_alt, nullptr, nullptr});
      DVS_TRY(launch_pyramid_chain(h, nsrc, nimg, h->d_pyr_alt, h->pf_stream, false));
    }
  }
  h->timer.end(h->pf_stream);
  h->pf_idx ^= 1;
  h->ev_prefetch = h->ev_pf2[h->pf_idx];
  DVS_HIP(hipEventRecord(h->ev_prefetch, h->pf_stream));
  h->pf_valid = true; h->pf_img = next_img0; h->pf_step = src.step0; h->pf_fstride = src.fstride0; h->pf_nimg = nimg;
  return DVS_OK;
}

// 7x7 fixed-point Gaussian of every level (ORBextractor.cpp:1132-1133)
// streaming blur kernel: dword-aligned level-0 rows of width % 4 == 0 (border by byte permutes) and levels >= 1 written by
// k_resize4 / k_pyr_cascade (which also write the reflected border columns); anything else takes the generic tile kernel
bool blur_stream_ok(const dvs_orb* h, const ImgSrc& src, bool cascade) {
  const Geom& G = h->geom;
  bool stream_ok = (((uintptr_t)src.img0) | src.step0 | src.fstride0) % 4 == 0 && G.lv[0].w % 4 == 0;
  for (int l = 1; l < G.nlevels; l++) stream_ok = stream_ok && (cascade || G.lv[l].gtab >= 0);
  return stream_ok;
}

void launch_blur(dvs_orb* h, const ImgSrc& src, int nimg, hipStream_t bst, bool cascade) {
  const Geom& G = h->geom;
  const bool stream_ok = blur_stream_ok(h, src, cascade);
  // matrix-core blur: 16-byte aligned rows (the pyramid block always is; a caller's level 0 when its pointer and strides are)
  const bool mfma_ok = h->env_blur_mfma && h->blur_mfma_ok && stream_ok &&
                       (((uintptr_t)src.img0 | src.step0 | src.fstride0) % 16 == 0) && src.step0 >= 16;
  if (mfma_ok && h->env_blur_mfma == 2)
    hipLaunchKernelGGL(k_blur_mfma_direct, dim3(h->n_blurcols, nimg), dim3(256), 0, bst, h->d_geom, h->d_blurcols, h->n_blurcols, src, h->d_blur,
                       h->d_blurtab, h->blur_avt);
  else if (mfma_ok)
    hipLaunchKernelGGL(k_blur_mfma, dim3(h->n_blurcols, nimg), dim3(256), 0, bst, h->d_geom, h->d_blurcols, h->n_blurcols, src, h->d_blur,
                       h->d_blurtab, h->blur_avt);
  else if (stream_ok)
    hipLaunchKernelGGL(k_blur_stream, dim3((G.blurStrips + 3) / 4, nimg), dim3(256), 0, bst, h->d_geom, h->d_strips, G.blurStrips, src, h->d_blur);
  else
    hipLaunchKernelGGL(k_blur, dim3(G.blurTiles, nimg), dim3(256), 0, bst, h->d_geom, h->d_tiles, src, h->d_blur);
}

// quad-tree of every (frame, level) on `qs`.  graded: one launch per level class with that class's capacities (LDS) instead of level 0's
void launch_octree(dvs_orb* h, int nimg, hipStream_t qs, uint32_t levelMask, bool graded) {
  const Geom& G = h->geom;
  if (graded && h->oct_ncls >= 2) {
    for (int c = 0; c < h->oct_ncls; c++) {
      const int l0 = h->oct_l0[c], nl = h->oct_l0[c + 1] - l0;
      if (nl <= 0) continue;
      // the first class holds the long workgroups (levels 0 and 1): 512 threads each; the others 256
      hipLaunchKernelGGL(k_octree, dim3(nimg, nl), dim3(c == 0 ? kOctTMax : kOctT), h->oct_smem_cls[c], qs, h->d_geom, h->d_cand, h->d_cellcount, h->d_celloff,
                         h->d_pts, h->d_nodeof, h->d_candtotal, h->d_lvlkp, h->d_lvlcount, h->oct_nmax_cls[c], h->oct_ptscap_cls[c], levelMask, l0);
    }
    return;
  }
  // 512-thread workgroups while there is at most one of them per CU (<= 32 frames of 8 levels: +9..11 % at 8 / 16 / 32 frames); beyond,
  // two 256-thread workgroups per CU run beside the blur and a pipelined caller's match, and the wave slots 512 threads hold cost
  // those more than the shorter tree returns (64 frames: 0.681 -> 0.664 ms per step; 128 / 384 threads: 0.729 / 0.690)
  const int oct_t = h->env_oct_threads ? h->env_oct_threads : (G.nlevels * nimg <= 256 ? kOctTMax : kOctT);
  hipLaunchKernelGGL(k_octree, dim3(nimg, G.nlevels), dim3(oct_t), h->octree_smem, qs, h->d_geom, h->d_cand, h->d_cellcount,
                     h->d_celloff, h->d_pts, h->d_nodeof, h->d_candtotal, h->d_lvlkp, h->d_lvlcount, h->octree_nmax, h->octree_ptscap, levelMask, 0);
}

// Enqueue the whole extraction of `nimg` frames whose level 0 is described by `src`.  Schedule:
//   main stream      [wait: this batch's chain]  FAST ............  quad-tree ......................  [descriptors, if not deferred]
//   prefetch stream  next batch's level chain (beside FAST)
//   auxiliary stream [reuse guard, chain join]                     blur (beside the quad-tree)  ->  [descriptors, if deferred]
// A caller stream released by the after-FAST event runs beside quad-tree and blur (bench.py: the previous batch's match).
dvs_status enqueue_extract(dvs_orb* h, ImgSrc src, int nimg, dvs_keypoint* d_kps, u8* d_desc, int capacity, int* d_nout,
                           const u8* next_img0 = nullptr, bool may_defer = false) {
  const Geom& G = h->geom;
  hipStream_t st = h->stream;
  const bool aligned0 = (((uintptr_t)src.img0) | src.step0 | src.fstride0) % 4 == 0;
  const uint32_t allLevels = G.nlevels >= 32 ? ~0u : ((1u << G.nlevels) - 1u);
  const bool sharded = (src.levelMask & allLevels) != allLevels || src.slotted;   // level-sharded call: the chain up to its top level, its levels only
  // a pyramid prefetched for exactly this batch (same buffer, layout and count)?  then it is already (being) built in d_pyr_alt
  const bool prefetched = h->pf_valid && h->overlap && h->pf_img == src.img0 && h->pf_step == src.step0 &&
                          h->pf_fstride == src.fstride0 && h->pf_nimg == nimg;
  h->pf_valid = false;
  // A deferred descriptor stage of the previous call (`pend`) runs on the auxiliary stream beside THIS call's FAST.  What it still
  // reads is protected without a wait on the main stream: the pyramids rotate over three buffers (launch_prefetch), the level
  // keypoint lists over three sets (below), the blurred block is rewritten on its own stream.  Anything but the pipelined pattern
  // (a prefetched batch from a caller that takes its outputs by event) joins it first.
  bool pend = h->out_pending;
  h->out_pending = false;
  if (pend && !(prefetched && may_defer)) {
    DVS_HIP(hipStreamWaitEvent(st, h->ev_out, 0));
    pend = false;
  }
  // this call defers its own descriptor stage: outputs by event, stage on the auxiliary stream, main stream not joined
  const bool will_defer = h->overlap && may_defer && h->output_event && h->defer_outputs && !sharded;
  // quad-tree off the main stream: the pipelined pattern only (this call's pyramid prefetched, its descriptor stage deferred).  FAST
  // writes the other candidate set: the previous call's tree may still be reading its own.  (Any other call joined the previous call's
  // auxiliary work above — its tree included — and keeps the current set.)
  const bool async = h->async_oct && prefetched && will_defer && aligned0;
  if (async) {
    h->cset = (h->cset + 1) % h->ring;
    if (!h->d_cand2[h->cset]) {
      DVS_HIP(hipMalloc((void**)&h->d_cand2[h->cset], (size_t)h->max_batch * G.candPerFrame * 4));
      DVS_HIP(hipMalloc((void**)&h->d_cellcount2[h->cset], (size_t)h->max_batch * G.totalCells * 4));
    }
    h->d_cand = h->d_cand2[h->cset]; h->d_cellcount = h->d_cellcount2[h->cset];
    if (h->tail_stream) {
      h->bset = (h->bset + 1) % h->ring;
      if (!h->d_blur3[h->bset]) DVS_HIP(hipMalloc((void**)&h->d_blur3[h->bset], (size_t)h->max_batch * G.frameBytes));
      h->d_blur = h->d_blur3[h->bset];
    }
  } else if (h->last_async) {
    DVS_HIP(hipStreamWaitEvent(st, h->ev_octdone[h->cset], 0));   // this call's FAST rewrites the set the previous call's tree reads
  }
  h->last_async = async;
  h->async_run = async ? h->async_run + 1 : 0;

  // 1. pyramid: level l from level l-1 (serial chain, ORBextractor.cpp:1171-1192)
  //    prefetched: nothing to build (non-deferred calls joined the chain through their blur — no barrier packet at all then);
  //    few frames (the live one-frame-per-callback pattern): the launch chain, not the arithmetic, sets the latency -> all levels in
  //    ONE launch (k_pyr_cascade: +26 % at 2 frames, +12 % at 8, -2 % at 16; beside FAST at 64 frames the chain wins, 0.20 vs 0.29 ms);
  //    otherwise the seven launches run on the auxiliary stream while FAST starts on level 0 and follows level by level.
  if (prefetched) {
    std::swap(h->d_pyr, h->d_pyr_alt);
    if (!h->pf_joined) DVS_HIP(hipStreamWaitEvent(st, h->ev_prefetch, 0));
  }
  h->pf_joined = false;
  src.pyr = h->d_pyr;
  int topLevel = 0;
  for (int l = 0; l < G.nlevels; l++) if ((src.levelMask >> l) & 1u) topLevel = l;
  const bool want_cascade = h->env_cascade >= 0 ? h->env_cascade == 1 : nimg <= 8;
  const bool cascade = !prefetched && !sharded && want_cascade && aligned0 && G.pyrTiles > 0 && 2 * (size_t)G.pyrLds <= 160 * 1024;
  const bool ov = !prefetched && !cascade && !sharded && h->overlap && G.nlevels >= 2;   // in-step chain beside FAST
  if (cascade) {
    h->timer.begin(DVS_STAGE_PYRAMID, st);
    hipLaunchKernelGGL(k_pyr_cascade<256>, dim3(G.pyrTiles, nimg), dim3(256), 2 * (size_t)G.pyrLds, st, h->d_geom, h->d_pyrtiles, src,
                       h->d_xofs, h->d_alpha, h->d_yofs, h->d_beta, G.pyrLds);
    h->timer.end(st);
  } else if (!prefetched) {
    hipStream_t pst = st;
    if (ov) {
      pst = h->aux_stream;
      DVS_HIP(hipEventRecord(h->ev_start, st));          // inputs ready / previous call's consumers of the pyramid done
      DVS_HIP(hipStreamWaitEvent(pst, h->ev_start, 0));
    }
    h->timer.begin(DVS_STAGE_PYRAMID, pst);
    DVS_TRY(launch_pyramid_chain(h, src, nimg, h->d_pyr, pst, ov, sharded ? topLevel : -1));
    h->timer.end(pst);
  }
  if (next_img0 && h->overlap && G.nlevels >= 2) DVS_TRY(launch_prefetch(h, src, nimg, next_img0, pend));

  // Joins that only the descriptor stage needs ride on the blur's (auxiliary) stream, off the main stream's critical path, and are
  // enqueued there BEFORE FAST so that their barrier packets (5-8 us each between two dependent kernels of one queue) are consumed
  // while FAST runs: the caller's reuse guard (outputs are written by the descriptor stage, which follows the blur) and the
  // completion of the next batch's chain (then the next call's FAST needs no barrier packet in front of it).  A deferring call
  // takes the guard right in front of its descriptor stage instead, so that a slow reader does not hold up the blur as well.
  if (h->overlap) {
    if (h->guard_event && !will_defer) { DVS_HIP(hipStreamWaitEvent(h->aux_stream, h->guard_event, 0)); h->guard_event = nullptr; }
    if (h->pf_valid && !sharded && !(h->defer_outputs && h->output_event)) { DVS_HIP(hipStreamWaitEvent(h->aux_stream, h->ev_prefetch, 0)); h->pf_joined = true; }
  }

  // (tail mode: this call's blur first — it needs the pyramid only, which the main stream has just joined; the one record behind FAST
  // then covers it too)
  const bool tail = async && h->tail_stream != nullptr;
  if (tail) {
    h->timer.begin(DVS_STAGE_BLUR, st);
    launch_blur(h, src, nimg, st, cascade);
    h->timer.end(st);
  }

  // 2. FAST per cell.  One launch over all levels; with the in-step chain one launch per level, each gated on its own level only
  //    (the small tail levels — each < 1/16 of the cells — share one); a level-sharded call launches its levels.
  if (ov) {
    int tail = G.nlevels;
    while (tail > 2 && G.lv[tail - 1].nCells * 16 < G.totalCells) tail--;
    for (int l = 0; l <= tail && l < G.nlevels; l++) {
      const int lastl = l == tail ? G.nlevels - 1 : l;
      if (lastl > 0) DVS_HIP(hipStreamWaitEvent(st, h->ev_level[lastl], 0));
      h->timer.begin(DVS_STAGE_FAST, st, l == 0);
      launch_fast(h, src, nimg, st, G.lv[l].cellBase, G.lv[lastl].cellBase + G.lv[lastl].nCells);
      h->timer.end(st);
    }
  } else {
    h->timer.begin(DVS_STAGE_FAST, st);
    if (sharded) {
      for (int l = 0; l < G.nlevels; l++)
        if ((src.levelMask >> l) & 1u) launch_fast(h, src, nimg, st, G.lv[l].cellBase, G.lv[l].cellBase + G.lv[l].nCells);
    } else {
      launch_fast(h, src, nimg, st, 0, G.totalCells);
    }
    h->timer.end(st);
  }
  // The blur only depends on the pyramid, but it is forked onto the auxiliary stream AFTER FAST so that the throughput-bound blur
  // fills the machine while the latency-bound quad-tree (one workgroup per frame x level) runs beside it; forked before FAST the two
  // throughput-bound kernels merely shared the CUs.  One record behind FAST serves the caller and the blur's fork.
  hipStream_t bst = h->overlap ? h->aux_stream : st;
  hipEvent_t ev_fastdone = h->after_fast_event ? h->after_fast_event : h->ev_fork;
  if (h->after_fast_event || h->overlap) DVS_HIP(hipEventRecord(ev_fastdone, st));

  // 3. quad-tree (latency-bound: launched first so that its workgroups become resident ahead of the blur's).  After a deferred stage
  //    it takes the next of the three level keypoint sets: stages k - 1 and k - 2 may still read theirs; stage k - 3 wrote its event
  //    before the level chain of THIS batch started, which this call's FAST waited for.
  if (pend) {
    h->lset = (h->lset + 1) % h->ring;
    if (!h->d_lvlkp3[h->lset]) {
      DVS_HIP(hipMalloc((void**)&h->d_lvlcount3[h->lset], (size_t)h->max_batch * G.nlevels * 4));
      DVS_HIP(hipMalloc((void**)&h->d_lvlkp3[h->lset], (size_t)h->max_batch * (size_t)G.kpBlock * 4));
    }
    h->d_lvlkp = h->d_lvlkp3[h->lset]; h->d_lvlcount = h->d_lvlcount3[h->lset];
  }
  //    Asynchronous (dvs_orb_set_async_quadtree, pipelined callers): on the auxiliary stream behind this call's FAST — nothing on the main
  //    stream needs it, so the next call's FAST follows this one's immediately and the tree runs beside it.
  hipStream_t qs = async ? bst : st;
  if (async) DVS_HIP(hipStreamWaitEvent(bst, ev_fastdone, 0));
  // a lane (one stream, a few frames): quad-trees and streaming blur in one launch (k_octree_blur) — the blur then runs beside the trees
  // instead of behind them (the lane's chain: -13 us of ~140 at one frame)
  const int blurRows = (G.blurStrips + kOctTMax / 64 - 1) / (kOctTMax / 64);
  const bool fuse_blur = h->single_stream && !async && !sharded && !(h->env_blur_mfma && h->blur_mfma_ok) &&
                         blur_stream_ok(h, src, cascade) && (long)nimg * (G.nlevels + blurRows) <= 256;
  h->timer.begin(DVS_STAGE_OCTREE, qs);
  if (fuse_blur) {
    if (h->guard_event) { DVS_HIP(hipStreamWaitEvent(qs, h->guard_event, 0)); h->guard_event = nullptr; }   // (the blur's place below takes it otherwise)
    hipLaunchKernelGGL(k_octree_blur, dim3(nimg, G.nlevels + blurRows), dim3(kOctTMax), h->octree_smem, qs, h->d_geom, h->d_cand, h->d_cellcount,
                       h->d_celloff, h->d_pts, h->d_nodeof, h->d_candtotal, h->d_lvlkp, h->d_lvlcount, h->octree_nmax, h->octree_ptscap, src.levelMask,
                       G.nlevels, h->d_strips, G.blurStrips, src, h->d_blur);
  } else {
    launch_octree(h, nimg, qs, src.levelMask, async && G.nlevels * nimg > 256);
  }
  h->timer.end(qs);
  if (async) { DVS_HIP(hipEventRecord(h->ev_octdone[h->cset], qs)); h->octdone_valid[h->cset] = true; }

  // 4. blur
  if (async) {}   // (the auxiliary stream already waited for FAST in front of the quad-tree)
  else if (bst != st) DVS_HIP(hipStreamWaitEvent(bst, ev_fastdone, 0));
  else if (h->guard_event) DVS_HIP(hipStreamWaitEvent(st, h->guard_event, 0));
  hipEvent_t late_guard = bst != st ? h->guard_event : nullptr;   // still set only for a deferring call (see above)
  h->guard_event = nullptr;   // one-shot
  if (!tail && !fuse_blur) {
    h->timer.begin(DVS_STAGE_BLUR, bst);
    launch_blur(h, src, nimg, bst, cascade);
    h->timer.end(bst);
  }

  // 5. orientation + descriptors + output records
  hipStream_t dst = st;
  if (will_defer) {
    // the stage follows the blur on the auxiliary stream and the main stream is NOT joined: the next call's FAST (vector-ALU bound,
    // light on memory) starts at once and runs beside it (fetch-bound).  Consumers order themselves on the caller's output event.
    dst = tail ? h->tail_stream : bst;
    if (tail) {     // the quad-tree's own event: the tree followed FAST, FAST followed the blur
      DVS_HIP(hipStreamWaitEvent(dst, h->ev_octdone[h->cset], 0));
    } else if (!async) {   // (asynchronous: the quad-tree precedes on this very stream)
      DVS_HIP(hipEventRecord(h->ev_oct, st));
      DVS_HIP(hipStreamWaitEvent(bst, h->ev_oct, 0));
    }
    if (late_guard) DVS_HIP(hipStreamWaitEvent(dst, late_guard, 0));   // the caller's readers of the output buffers
  } else if (bst != st) {
    DVS_HIP(hipEventRecord(h->ev_blur, bst));
    DVS_HIP(hipStreamWaitEvent(st, h->ev_blur, 0));
  }
  h->timer.begin(DVS_STAGE_DESCRIBE, dst);
  // tile-by-tile visiting order from 16 frames on: below, the descriptor stage is latency and the ranking kernel's 12 us count
  if (h->env_desc_order && nimg >= 16) {
    hipLaunchKernelGGL(k_kp_order, dim3(nimg, G.nlevels), dim3(256), 0, dst, h->d_geom, h->d_lvlkp, h->d_lvlcount, h->d_kpsorted, h->d_kpsortidx);
    hipLaunchKernelGGL(k_describe, dim3((G.kpBlock + 4 * kDescKP - 1) / (4 * kDescKP), nimg), dim3(256), 0, dst, h->d_geom, src, h->d_blur,
                       h->d_kpsorted, h->d_kpsortidx, h->d_lvlcount, d_kps, d_desc, d_nout, capacity);
  } else {   // list order: the identity index table
    hipLaunchKernelGGL(k_describe, dim3((G.kpBlock + 4 * kDescKP - 1) / (4 * kDescKP), nimg), dim3(256), 0, dst, h->d_geom, src, h->d_blur,
                       h->d_lvlkp, h->d_kpident, h->d_lvlcount, d_kps, d_desc, d_nout, capacity);
  }
  h->timer.end(dst);
  if (will_defer) {
    h->ev_out = h->ev_outs[h->out_gen % (h->ring - 1)];
    h->out_gen++;
    DVS_HIP(hipEventRecord(h->ev_out, dst));
    DVS_HIP(hipEventRecord(h->output_event, dst));
    h->out_pending = true;
    h->gate_event = nullptr;   // the next call's chain is gated on ev_outs (launch_prefetch)
  } else {   // end of the call on the main stream: the caller's output event, or our own — the next call's prefetch gate
    h->gate_event = (may_defer && h->output_event) ? h->output_event : h->ev_end;
    DVS_HIP(hipEventRecord(h->gate_event, st));
  }
  DVS_HIP(hipGetLastError());
  h->last_nimg = nimg;
  h->last_src = src;
  return DVS_OK;
}

}  // namespace

extern "C" {

static dvs_status orb_create(const dvs_orb_params* params, int32_t device, bool single_stream, bool on_stream, hipStream_t ext, dvs_orb** out);
dvs_status dvs_orb_create(const dvs_orb_params* params, int32_t device, dvs_orb** out) { return orb_create(params, device, false, false, nullptr, out); }
DVS_HOOK dvs_status dvs_orb_create_single_stream(const dvs_orb_params* params, int32_t device, dvs_orb** out) { return orb_create(params, device, true, false, nullptr, out); }
DVS_HOOK dvs_status dvs_orb_create_on_stream(const dvs_orb_params* params, int32_t device, void* hip_stream, dvs_orb** out) {
  return orb_create(params, device, true, true, (hipStream_t)hip_stream, out);
}

static dvs_status orb_create(const dvs_orb_params* params, int32_t device, bool single_stream, bool on_stream, hipStream_t ext, dvs_orb** out) {
  DVS_ARG(params && out);
  *out = nullptr;
  DVS_ARG(params->nlevels >= 1 && params->nlevels <= DVS_MAX_LEVELS);
  DVS_ARG(params->nfeatures >= 0 && params->scale_factor > 1.0f && params->max_batch >= 0);
  DVS_TRY(check_device(device));
  dvs_orb* h = new (std::nothrow) dvs_orb();
  if (!h) { set_error("out of host memory"); return DVS_ERR_HIP; }
  h->prm = *params;
  bool allzero = true;
  for (int i = 0; i < 7; i++) allzero = allzero && params->gauss_kernel[i] == 0;
  if (allzero) { const int k[7] = {18, 34, 48, 56, 48, 34, 18}; memcpy(h->prm.gauss_kernel, k, sizeof(k)); }
  int ksum = 0;
  for (int i = 0; i < 7; i++) {
    ksum += h->prm.gauss_kernel[i];
    if (h->prm.gauss_kernel[i] < 0 || h->prm.gauss_kernel[i] > 255) ksum = 1 << 20;
  }
  if (ksum > 257) { delete h; set_error("gauss_kernel sum %d would overflow the Q8.8 row buffer", ksum); return DVS_ERR_ARG; }
  h->device = device;
  h->max_batch = params->max_batch > 0 ? params->max_batch : 1;
  if (on_stream) {   // the caller's stream for good: no stream of its own is ever created
    h->stream = ext;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; set_error("hipStreamCreate: %s", hipGetErrorString(e)); return DVS_ERR_HIP; }
    h->stream = h->own_stream;
  }
  // The switches of this handle, read once (see the struct).  DVS_NO_OVERLAP=1 = dvs_orb_set_overlap(h, 0) from the start: every
  // stage alone on the main stream.
  h->single_stream = single_stream;
  h->overlap = env_int("DVS_NO_OVERLAP", 0) == 0 && !single_stream;
  h->env_cascade = env_int("DVS_CASCADE", -1);
  h->env_chain_graph = env_int("DVS_CHAIN_GRAPH", -1);
  h->env_blur_mfma = env_int("DVS_BLUR_MFMA", 0);
  h->env_host_poll = env_int("DVS_HOST_POLL", 1);
  h->env_oct_threads = env_int("DVS_OCT_T", 0);
  h->env_desc_order = env_int("DVS_DESC_ORDER", 1);
  if (h->env_oct_threads != 0 && h->env_oct_threads != kOctT && h->env_oct_threads != kOctTMax) h->env_oct_threads = 0;
  // byte-aligned tile origin (probed once per process and device): every cell's interior then starts on a dword of the tile, so a
  // 36-pixel interior is always 9 column groups = 7 rows per trip of the rejection loop (aligned origin: 9 or 10 groups by the
  // cell's phase, 6 rows per trip for the latter).  FAST alone 0.295 -> 0.287 ms per 64 frames.
  h->fast_byte_dma = probe_byte_dma(device, h->stream) && env_int("DVS_FAST_BYTE_DMA", 1);
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  // the auxiliary stream carries the blur (and a deferred descriptor stage): LOWEST dispatch priority, so that the quad-tree
  // workgroups launched at the same moment on the main stream all become resident first — with the blur's workgroups dispatched
  // ahead of them some quad-tree workgroups started 90 us late and the kernel took 170 us instead of 110 (64 frames per step:
  // 0.628 -> 0.592 ms, neutral below 64)
  bool ok = single_stream || (hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio_lo) == hipSuccess &&
                              hipStreamCreateWithPriority(&h->pf_stream, hipStreamNonBlocking, prio_hi) == hipSuccess);
  hipEvent_t* evs[] = {&h->ev_fork, &h->ev_blur, &h->ev_start, &h->ev_chain_gate, &h->ev_pf2[0], &h->ev_pf2[1], &h->ev_outs[0], &h->ev_outs[1], &h->ev_outs[2],
                       &h->ev_oct, &h->ev_end, &h->ev_octdone[0], &h->ev_octdone[1], &h->ev_octdone[2], &h->ev_octdone[3]};
  for (hipEvent_t* ev : evs) ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
  for (int l = 1; l < params->nlevels; l++) ok = ok && hipEventCreateWithFlags(&h->ev_level[l], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    dvs_orb_destroy(h);
    set_error("stream / event creation failed");
    return DVS_ERR_HIP;
  }
  build_ctor_tables(h);
  *out = h;
  return DVS_OK;
}

void dvs_orb_destroy(dvs_orb* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
  if (h->pf_stream) (void)hipStreamSynchronize(h->pf_stream);
  h->timer.resolve();
  free_workspace(h);
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  if (h->pf_stream) (void)hipStreamDestroy(h->pf_stream);
  hipEvent_t evs[] = {h->ev_fork, h->ev_blur, h->ev_start, h->ev_chain_gate, h->ev_pf2[0], h->ev_pf2[1], h->ev_outs[0], h->ev_outs[1], h->ev_outs[2], h->ev_oct, h->ev_end,
                      h->ev_octdone[0], h->ev_octdone[1], h->ev_octdone[2], h->ev_octdone[3]};
  for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_level) if (e) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

int32_t dvs_orb_max_keypoints(const dvs_orb* h) { return h ? h->prm.nfeatures + 3 * h->prm.nlevels : 0; }

dvs_status dvs_orb_set_stream(dvs_orb* h, void* s) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)s;  // NULL is a real stream: HIP's legacy default stream
  return DVS_OK;
}
DVS_HOOK dvs_status dvs_orb_set_overlap(dvs_orb* h, int32_t on) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  if (on && h->single_stream) { set_error("dvs_orb_set_overlap: this extractor was created with one stream only"); return DVS_ERR_UNSUPPORTED; }
  DVS_HIP(hipStreamSynchronize(h->stream));
  if (h->aux_stream) DVS_HIP(hipStreamSynchronize(h->aux_stream));
  if (h->pf_stream) DVS_HIP(hipStreamSynchronize(h->pf_stream));
  h->pf_valid = false; h->pf_joined = false;
  if (h->out_pending) DVS_HIP(hipEventSynchronize(h->ev_out));
  h->out_pending = false;   // everything, a deferred descriptor stage included, has completed above
  h->overlap = on != 0;
  return DVS_OK;
}
DVS_HOOK dvs_status dvs_orb_set_async_quadtree(dvs_orb* h, int32_t on) {
  DVS_ARG(h);
  if (on && h->single_stream) { set_error("dvs_orb_set_async_quadtree: this extractor was created with one stream only"); return DVS_ERR_UNSUPPORTED; }
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  if (h->aux_stream) DVS_HIP(hipStreamSynchronize(h->aux_stream));
  if (h->pf_stream) DVS_HIP(hipStreamSynchronize(h->pf_stream));
  if (h->out_pending) DVS_HIP(hipEventSynchronize(h->ev_out));
  h->out_pending = false; h->last_async = false;
  h->async_oct = on != 0;
  return DVS_OK;
}
DVS_HOOK int64_t dvs_orb_chain_graph_launches(const dvs_orb* h) { return h ? h->chain_graph_launches : 0; }

DVS_HOOK dvs_status dvs_orb_set_tail_stream(dvs_orb* h, void* hip_stream) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  if (h->aux_stream) DVS_HIP(hipStreamSynchronize(h->aux_stream));
  if (h->pf_stream) DVS_HIP(hipStreamSynchronize(h->pf_stream));
  if (h->out_pending) { DVS_HIP(hipEventSynchronize(h->ev_out)); h->out_pending = false; }
  h->tail_stream = (hipStream_t)hip_stream;
  // the four-stream form rotates over rings of four (see `ring`); everything is idle here: restart every rotation at its first set
  h->ring = hip_stream ? 4 : 3;
  h->cset = h->bset = h->lset = 0; h->out_gen = 0; h->async_run = 0;
  for (bool& v : h->octdone_valid) v = false;
  h->d_cand = h->d_cand2[0]; h->d_cellcount = h->d_cellcount2[0]; h->d_blur = h->d_blur3[0];
  h->d_lvlkp = h->d_lvlkp3[0]; h->d_lvlcount = h->d_lvlcount3[0];
  h->last_async = false;
  return DVS_OK;
}
DVS_HOOK dvs_status dvs_orb_use_own_stream(dvs_orb* h) {
  DVS_ARG(h);
  if (!h->own_stream) { set_error("dvs_orb_use_own_stream: this extractor was created on a caller's stream and has none of its own"); return DVS_ERR_UNSUPPORTED; }
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  h->stream = h->own_stream;
  return DVS_OK;
}
void* dvs_orb_get_stream(dvs_orb* h) { return h ? (void*)h->stream : nullptr; }

dvs_status dvs_orb_synchronize(dvs_orb* h) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  if (h->out_pending) { DVS_HIP(hipEventSynchronize(h->ev_out)); h->out_pending = false; }   // a deferred descriptor stage (auxiliary or tail stream)
  if (h->pf_stream) DVS_HIP(hipStreamSynchronize(h->pf_stream));  // an announced next batch's pyramid may still be reading the caller's images
  return DVS_OK;
}

dvs_status dvs_orb_get_tables(const dvs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                              int32_t* features_per_level, int32_t* umax16) {
  DVS_ARG(h);
  const int nl = h->prm.nlevels;
  if (scale) memcpy(scale, h->scale.data(), nl * 4);
  if (inv_scale) memcpy(inv_scale, h->inv_scale.data(), nl * 4);
  if (sigma2) memcpy(sigma2, h->sigma2.data(), nl * 4);
  if (inv_sigma2) memcpy(inv_sigma2, h->inv_sigma2.data(), nl * 4);
  if (features_per_level) memcpy(features_per_level, h->feat_per_level.data(), nl * 4);
  if (umax16) memcpy(umax16, h->umax, 16 * 4);
  return DVS_OK;
}

dvs_status dvs_orb_level_size(const dvs_orb* h, int32_t rows, int32_t cols, int32_t level, int32_t* lr, int32_t* lc) {
  DVS_ARG(h && lr && lc && level >= 0 && level < h->prm.nlevels);
  int r, c;
  level_size(h, rows, cols, level, r, c);
  *lr = r; *lc = c;
  return DVS_OK;
}

dvs_status dvs_orb_extract_batch_device(dvs_orb* h, const uint8_t* d_imgs, int32_t nimg, int32_t rows, int32_t cols,
                                        size_t step, size_t frame_stride, dvs_keypoint* d_kps, uint8_t* d_desc,
                                        int32_t capacity, int32_t* d_n_out) {
  DVS_ARG(h && d_kps && d_desc && d_n_out && nimg >= 0);
  if (!d_imgs || rows <= 0 || cols <= 0) { set_error("empty image"); return DVS_ERR_EMPTY; }
  DVS_ARG(step >= (size_t)cols);
  if (nimg > h->max_batch) { set_error("nimg %d exceeds max_batch %d", nimg, h->max_batch); return DVS_ERR_CAPACITY; }
  if (capacity < dvs_orb_max_keypoints(h)) { set_error("capacity %d < %d", capacity, dvs_orb_max_keypoints(h)); return DVS_ERR_CAPACITY; }
  DVS_HIP(hipSetDevice(h->device));
  DVS_TRY(ensure_workspace(h, rows, cols));
  if (nimg == 0) return DVS_OK;
  ImgSrc src{d_imgs, (uint64_t)step, (uint64_t)frame_stride, h->d_pyr, ~0u, 0};
  const u8* next = h->next_hint;
  h->next_hint = nullptr;
  return enqueue_extract(h, src, nimg, d_kps, d_desc, capacity, d_n_out, next, true);
}

// ---- level-sharded extraction -----------------------------------------------------------------------------------
static LevelBlockLayout level_block_layout(const dvs_orb* h, int nimg) {
  LevelBlockLayout Y{};
  Y.nl = h->prm.nlevels;
  int off = 0;
  for (int l = 0; l < Y.nl; l++) { Y.kpOff[l] = off; off += h->feat_per_level[l] + 4; }   // = LevelGeom::kpOff (build_geometry)
  Y.kpBlock = off;
  Y.kpsOff = align_up((uint64_t)nimg * Y.nl * 4, 64);
  Y.descOff = Y.kpsOff + align_up((uint64_t)nimg * Y.kpBlock * sizeof(dvs_keypoint), 64);
  Y.blockBytes = Y.descOff + align_up((uint64_t)nimg * Y.kpBlock * 32, 64);
  return Y;
}

size_t dvs_orb_level_block_bytes(const dvs_orb* h, int32_t nimg) { return h && nimg > 0 ? (size_t)level_block_layout(h, nimg).blockBytes : 0; }

dvs_status dvs_orb_extract_levels_device(dvs_orb* h, const uint8_t* d_imgs, int32_t nimg, int32_t rows, int32_t cols, size_t step,
                                         size_t frame_stride, uint32_t level_mask, uint8_t* d_block) {
  DVS_ARG(h && d_block && nimg >= 0);
  if (!d_imgs || rows <= 0 || cols <= 0) { set_error("empty image"); return DVS_ERR_EMPTY; }
  DVS_ARG(step >= (size_t)cols && ((uintptr_t)d_block) % 16 == 0);
  if (nimg > h->max_batch) { set_error("nimg %d exceeds max_batch %d", nimg, h->max_batch); return DVS_ERR_CAPACITY; }
  DVS_HIP(hipSetDevice(h->device));
  DVS_TRY(ensure_workspace(h, rows, cols));
  if (nimg == 0) return DVS_OK;
  const LevelBlockLayout Y = level_block_layout(h, nimg);
  if (Y.kpBlock != h->geom.kpBlock) { set_error("level block layout mismatch"); return DVS_ERR_ARG; }
  const uint32_t all = h->prm.nlevels >= 32 ? ~0u : ((1u << h->prm.nlevels) - 1u);
  ImgSrc src{d_imgs, (uint64_t)step, (uint64_t)frame_stride, h->d_pyr, level_mask & all, 1};
  h->next_hint = nullptr;
  return enqueue_extract(h, src, nimg, (dvs_keypoint*)(d_block + Y.kpsOff), d_block + Y.descOff, Y.kpBlock, (int*)d_block, nullptr);
}

dvs_status dvs_orb_merge_levels_device(dvs_orb* h, const uint8_t* d_blocks, int32_t world, const int32_t* level_owner, int32_t nimg,
                                       dvs_keypoint* d_kps, uint8_t* d_desc, int32_t capacity, int32_t* d_n_out) {
  DVS_ARG(h && d_blocks && level_owner && d_kps && d_desc && d_n_out && world >= 1 && nimg >= 0);
  if (capacity < dvs_orb_max_keypoints(h)) { set_error("capacity %d < %d", capacity, dvs_orb_max_keypoints(h)); return DVS_ERR_CAPACITY; }
  if (nimg == 0) return DVS_OK;
  DVS_HIP(hipSetDevice(h->device));
  LevelBlockLayout Y = level_block_layout(h, nimg);
  for (int l = 0; l < Y.nl; l++) {
    DVS_ARG(level_owner[l] >= 0 && level_owner[l] < world);
    Y.owner[l] = level_owner[l];
  }
  hipLaunchKernelGGL(k_merge_levels, dim3((Y.kpBlock + 255) / 256, nimg), dim3(256), 0, h->stream, Y, d_blocks, nimg, d_kps, d_desc, capacity, d_n_out);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_set_output_event(dvs_orb* h, void* hip_event) {
  DVS_ARG(h);
  h->output_event = (hipEvent_t)hip_event;
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_set_defer_outputs(dvs_orb* h, int32_t on) {
  DVS_ARG(h);
  h->defer_outputs = on != 0;
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_set_reuse_guard_event(dvs_orb* h, void* hip_event) {
  DVS_ARG(h);
  h->guard_event = (hipEvent_t)hip_event;
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_set_after_fast_event(dvs_orb* h, void* hip_event) {
  DVS_ARG(h);
  h->after_fast_event = (hipEvent_t)hip_event;
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_hint_next_batch_device(dvs_orb* h, const uint8_t* d_next_imgs) {
  DVS_ARG(h);
  h->next_hint = d_next_imgs;
  return DVS_OK;
}

dvs_status dvs_orb_extract_batch(dvs_orb* h, const uint8_t* const* imgs, int32_t nimg, int32_t rows, int32_t cols, size_t step,
                                 dvs_keypoint* kps, uint8_t* desc, int32_t capacity, int32_t* n_out) {
  DVS_ARG(h && n_out && nimg >= 0);
  for (int i = 0; i < nimg; i++) n_out[i] = 0;
  if (!imgs || rows <= 0 || cols <= 0) { set_error("empty image"); return DVS_ERR_EMPTY; }
  for (int i = 0; i < nimg; i++) if (!imgs[i]) { set_error("empty image %d", i); return DVS_ERR_EMPTY; }
  DVS_ARG(kps && desc && step >= (size_t)cols);
  DVS_HIP(hipSetDevice(h->device));
  DVS_TRY(ensure_workspace(h, rows, cols));
  const Geom& G = h->geom;
  const int cap = G.outCap;
  for (int b0 = 0; b0 < nimg; b0 += h->max_batch) {
    const int nb = std::min(h->max_batch, nimg - b0);
    // level 0 staged into the frame's pyramid block (the reference copies it too: copyMakeBorder, :1189)
    for (int i = 0; i < nb; i++)
      DVS_HIP(hipMemcpy2DAsync(h->d_pyr + (uint64_t)i * G.frameBytes + G.lv[0].off, G.lv[0].pitch, imgs[b0 + i], step, cols, rows,
                               hipMemcpyHostToDevice, h->stream));
    ImgSrc src{h->d_pyr + G.lv[0].off, (uint64_t)G.lv[0].pitch, G.frameBytes, h->d_pyr, ~0u, 0};
    DVS_TRY(enqueue_extract(h, src, nb, h->d_kps, h->d_desc, cap, h->d_nout));
    if (h->env_host_poll) {
      // results by k_export_host into the pinned block; poll its sequence number (bounded spin, then the stream wait)
      static_assert(sizeof(dvs_keypoint) == 28, "k_export_host copies keypoints as 7 dwords");
      const int seq = ++h->export_seq;
      hipLaunchKernelGGL(k_export_host, dim3(4, nb), dim3(256), 0, h->stream, nb, cap, (const uint32_t*)h->d_kps, (const uint32_t*)h->d_desc,
                         h->d_nout, (uint32_t*)h->h_kps, (uint32_t*)h->h_desc, h->h_nout, h->d_ticket, h->h_seq, seq);
      DVS_HIP(hipGetLastError());
      const volatile int* ps = h->h_seq;
      const auto t0 = std::chrono::steady_clock::now();
      for (int spin = 1; *ps != seq; spin++) {
        __builtin_ia32_pause();
        if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      if (*ps != seq) DVS_HIP(hipStreamSynchronize(h->stream));
    } else {
      DVS_HIP(hipMemcpyAsync(h->h_nout, h->d_nout, nb * 4, hipMemcpyDeviceToHost, h->stream));
      DVS_HIP(hipMemcpyAsync(h->h_kps, h->d_kps, (size_t)nb * cap * sizeof(dvs_keypoint), hipMemcpyDeviceToHost, h->stream));
      DVS_HIP(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)nb * cap * 32, hipMemcpyDeviceToHost, h->stream));
      DVS_HIP(hipStreamSynchronize(h->stream));
    }
    for (int i = 0; i < nb; i++) {
      const int n = h->h_nout[i];
      if (n > capacity) { set_error("frame %d: %d keypoints > capacity %d", b0 + i, n, capacity); return DVS_ERR_CAPACITY; }
      n_out[b0 + i] = n;
      memcpy(kps + (size_t)(b0 + i) * capacity, h->h_kps + (size_t)i * cap, (size_t)n * sizeof(dvs_keypoint));
      memcpy(desc + (size_t)(b0 + i) * capacity * 32, h->h_desc + (size_t)i * cap * 32, (size_t)n * 32);
    }
  }
  return DVS_OK;
}

dvs_status dvs_orb_extract(dvs_orb* h, const uint8_t* gray, int32_t rows, int32_t cols, size_t step, dvs_keypoint* kps,
                           uint8_t* desc, int32_t capacity, int32_t* n_out) {
  DVS_ARG(h && n_out);
  *n_out = 0;
  if (!gray || rows <= 0 || cols <= 0) { set_error("empty image"); return DVS_ERR_EMPTY; }
  const uint8_t* one[1] = {gray};
  return dvs_orb_extract_batch(h, one, 1, rows, cols, step, kps, desc, capacity, n_out);
}

dvs_status dvs_orb_get_level(dvs_orb* h, int32_t frame, int32_t level, int32_t blurred, uint8_t* dst, int32_t cap_bytes) {
  DVS_ARG(h && dst && h->d_geom && frame >= 0 && frame < h->last_nimg && level >= 0 && level < h->geom.nlevels);
  const LevelGeom& L = h->geom.lv[level];
  if ((int64_t)L.w * L.h > cap_bytes) return DVS_ERR_CAPACITY;
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  if (level == 0 && !blurred) {
    DVS_HIP(hipMemcpy2D(dst, L.w, h->last_src.img0 + (uint64_t)frame * h->last_src.fstride0, h->last_src.step0, L.w, L.h, hipMemcpyDeviceToHost));
  } else {
    const u8* base = (blurred ? h->d_blur : h->d_pyr) + (uint64_t)frame * h->geom.frameBytes + L.off;
    DVS_HIP(hipMemcpy2D(dst, L.w, base, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
  }
  return DVS_OK;
}

static dvs_status read_packed(dvs_orb* h, const uint32_t* dsrc, int n, int32_t* xys) {
  std::vector<uint32_t> tmp(n);
  if (n) DVS_HIP(hipMemcpy(tmp.data(), dsrc, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) { xys[3 * i] = pt_x(tmp[i]); xys[3 * i + 1] = pt_y(tmp[i]); xys[3 * i + 2] = pt_s(tmp[i]); }
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_get_candidates(dvs_orb* h, int32_t frame, int32_t level, int32_t* xys, int32_t cap, int32_t* n) {
  DVS_ARG(h && xys && n && h->d_geom && frame >= 0 && frame < h->last_nimg && level >= 0 && level < h->geom.nlevels);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  int cnt = 0;
  DVS_HIP(hipMemcpy(&cnt, h->d_candtotal + frame * h->geom.nlevels + level, 4, hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) return DVS_ERR_CAPACITY;
  return read_packed(h, h->d_pts + (uint64_t)frame * h->geom.ptsPerFrame + h->geom.lv[level].ptsOff, cnt, xys);
}

DVS_HOOK dvs_status dvs_orb_get_level_keypoints(dvs_orb* h, int32_t frame, int32_t level, int32_t* xys, int32_t cap, int32_t* n) {
  DVS_ARG(h && xys && n && h->d_geom && frame >= 0 && frame < h->last_nimg && level >= 0 && level < h->geom.nlevels);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  int cnt = 0;
  DVS_HIP(hipMemcpy(&cnt, h->d_lvlcount + frame * h->geom.nlevels + level, 4, hipMemcpyDeviceToHost));
  *n = cnt;
  if (cnt > cap) return DVS_ERR_CAPACITY;
  return read_packed(h, h->d_lvlkp + (uint64_t)frame * h->geom.kpBlock + h->geom.lv[level].kpOff, cnt, xys);
}

DVS_HOOK dvs_status dvs_orb_enable_stage_timing(dvs_orb* h, int32_t on) {
  DVS_ARG(h);
  h->timer.resolve();
  h->timer.on = on != 0;
  return DVS_OK;
}

DVS_HOOK dvs_status dvs_orb_get_stage_times(dvs_orb* h, double* ms, int64_t* calls, int32_t reset) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  h->timer.resolve();
  for (int i = 0; i < DVS_STAGE_COUNT; i++) { if (ms) ms[i] = h->timer.ms[i]; if (calls) calls[i] = h->timer.calls[i]; }
  if (reset) h->timer.reset();
  return DVS_OK;
}

#ifdef DVS_TEST_HOOKS   // libdvslam_hip_test.so only (include/dvslam_hip_test.h)
// ---- host-logic test hooks (no GPU needed): the introsort replica and the glibc sincosf restatement
void dvs_test_sort_nodes(const int32_t* count, const int32_t* ulx, int32_t n, int32_t* perm) {
  std::vector<unsigned long long> v(n);
  for (int i = 0; i < n; i++) v[i] = ((unsigned long long)(uint32_t)count[i] << 28) | ((unsigned long long)(uint16_t)ulx[i] << 12) | (unsigned long long)i;
  lsort::sort(v.data(), (long)n, lsort::Less<12>());
  for (int i = 0; i < n; i++) perm[i] = (int)(v[i] & 0xFFFull);
}
void dvs_test_sort_nodes_ranked(const int32_t* count, const int32_t* ulx, int32_t n, int32_t* perm) {
  std::vector<unsigned long long> v(n);
  std::vector<int> lp(n + 1), rp(n + 1);
  for (int i = 0; i < n; i++) v[i] = ((unsigned long long)(uint32_t)count[i] << 28) | ((unsigned long long)(uint16_t)ulx[i] << 12) | (unsigned long long)i;
  lsort::sort_ranked(v.data(), (long)n, lsort::Less<12>(), lp.data(), rp.data());
  for (int i = 0; i < n; i++) perm[i] = (int)(v[i] & 0xFFFull);
}
dvs_status dvs_test_sort_nodes_device(const int32_t* count, const int32_t* ulx, int32_t n, int32_t* perm) {
  DVS_ARG(n >= 0 && n <= kMaxQuota && (n == 0 || (count && ulx && perm)));
  if (n == 0) return DVS_OK;
  std::vector<unsigned long long> v(n);
  for (int i = 0; i < n; i++) v[i] = ((unsigned long long)(uint32_t)count[i] << 28) | ((unsigned long long)(uint16_t)ulx[i] << 12) | (unsigned long long)i;
  unsigned long long* d = nullptr;
  DVS_HIP(hipMalloc(&d, sizeof(unsigned long long) * n));
  hipError_t e = hipMemcpy(d, v.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_test_sort, dim3(1), dim3(kOctT), 0, 0, d, n); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(v.data(), d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  DVS_HIP(e);
  for (int i = 0; i < n; i++) perm[i] = (int)(v[i] & 0xFFFull);
  return DVS_OK;
}
void dvs_test_sincosf(float a, float* s, float* c) { *s = gsc::sinf_(a); *c = gsc::cosf_(a); }
// geometry without touching the GPU: fills level sizes / cell grid / quotas for a resolution
dvs_status dvs_test_geometry(const dvs_orb_params* params, int32_t rows, int32_t cols, int32_t* level_w, int32_t* level_h,
                             int32_t* ncells, int32_t* quota, int32_t* wcell, int32_t* hcell) {
  DVS_ARG(params && params->nlevels >= 1 && params->nlevels <= DVS_MAX_LEVELS);
  dvs_orb h;
  h.prm = *params;
  build_ctor_tables(&h);
  Geom G;
  std::vector<Cell> cells; std::vector<BlurTile> tiles; std::vector<BlurStrip> strips; std::vector<ResizeGroup> rg; std::vector<PyrTile> pt;
  std::vector<int> xo, al, yo, be;
  DVS_TRY(build_geometry(&h, rows, cols, G, cells, tiles, strips, rg, pt, xo, al, yo, be));
  for (int l = 0; l < G.nlevels; l++) {
    if (level_w) level_w[l] = G.lv[l].w;
    if (level_h) level_h[l] = G.lv[l].h;
    if (ncells) ncells[l] = G.lv[l].nCells;
    if (quota) quota[l] = G.lv[l].N;
    if (wcell) wcell[l] = G.lv[l].wCell;
    if (hcell) hcell[l] = G.lv[l].hCell;
  }
  return DVS_OK;
}

#endif  // DVS_TEST_HOOKS

}  // extern "C"

#ifdef DVS_QT_PROF   // tools/qt_phase_profile.sh only: never part of the product or test libraries
extern "C" int dvs_prof_qt(unsigned long long* out96) {
  return (int)hipMemcpyFromSymbol(out96, HIP_SYMBOL(dvs::g_qt_prof), 96 * sizeof(unsigned long long));
}
#endif
