// dvslam/streaming_pipeline.hpp — the C++ host of the streaming step: extraction of batch i + match of batch i - 1, ONE call per
// step (dvs_pipeline_* of the C-ABI, csrc/pipeline.hip).  It replaces the reference's per-frame sequence in the frontend callback
// (src/frontend.cpp:1084 cvtColor, :1094-1096 (*orb_extractor_)(...) , :1123 orb_matcher_->match(current, previous)) for a host
// that keeps B frames at a time resident in device memory: the schedule bench.py times.
//
//   dvslam::StreamingPipeline pipe(64, 720, 1280, 2000);            // ORBextractor(2000, 1.2f, 8, 20, 7) inside
//   for (i = 0; i < steps; i++) pipe.step(d_batch[i], d_batch[i + 1]);   // asynchronous; d_batch[i + 1] may be nullptr
//   pipe.flush();                                                    // the last batch's match (the pipeline runs it one step late)
//   pipe.synchronize();
//   auto s = pipe.results(i);   // device pointers of batch i: keypoints, descriptors, counts, trainIdx, distance (+ events)
//
// Everything is owned by the handle; nothing is allocated per step.  A consumer on another HIP stream orders itself on
// s.ev_extracted / s.ev_matched (dvs_stream_wait_event).  Multi-GPU: one process (or thread) per GPU, frames sharded contiguously
// over the ranks, attach(comm) — the boundary frame then comes from dvs_exchange_boundary (one ncclAllGather per global batch).
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../dvslam_hip.h"

namespace dvslam {

class StreamingPipeline {
 public:
  StreamingPipeline(int batch, int rows, int cols, int nfeatures = 2000, float scaleFactor = 1.2f, int nlevels = 8, int iniThFAST = 20,
                    int minThFAST = 7, int device = 0, int nsets = 0 /* the library's choice: 4, or two per lane */, bool pipelined = true, int lanes = 0, int quadtree_async = 0)
      : device_(device), batch_(batch) {
    dvs_pipeline_params p;
    std::memset(&p, 0, sizeof(p));
    p.orb.nfeatures = nfeatures; p.orb.scale_factor = scaleFactor; p.orb.nlevels = nlevels; p.orb.ini_th_fast = iniThFAST; p.orb.min_th_fast = minThFAST;
    p.batch = batch; p.rows = rows; p.cols = cols; p.nsets = nsets; p.pipelined = pipelined ? 1 : 0; p.lanes = lanes; p.quadtree_async = quadtree_async;
    check(dvs_pipeline_create(&p, device, &h_), "dvs_pipeline_create");
    dvs_pipeline_set s;
    check(dvs_pipeline_get_set(h_, 0, &s), "dvs_pipeline_get_set");
    capacity_ = s.capacity;
  }
  ~StreamingPipeline() { dvs_pipeline_destroy(h_); }
  StreamingPipeline(const StreamingPipeline&) = delete;
  StreamingPipeline& operator=(const StreamingPipeline&) = delete;

  void attach(dvs_comm* comm) { check(dvs_pipeline_attach_comm(h_, comm), "dvs_pipeline_attach_comm"); }
  // d_imgs: `batch` gray frames, tight rows, device memory; d_next: the batch the next step() will pass, or nullptr
  void step(const uint8_t* d_imgs, const uint8_t* d_next = nullptr) { check(dvs_pipeline_step(h_, d_imgs, d_next, 0), "dvs_pipeline_step"); }
  void flush() { check(dvs_pipeline_flush(h_), "dvs_pipeline_flush"); }
  void synchronize() { check(dvs_pipeline_synchronize(h_), "dvs_pipeline_synchronize"); }
  void reset() { check(dvs_pipeline_reset(h_), "dvs_pipeline_reset"); }
  long long steps() const { return dvs_pipeline_steps(h_); }
  int batch() const { return batch_; }
  int nsets() const { return dvs_pipeline_nsets(h_); }   // output sets in rotation: results(step) stays valid until step + nsets() is enqueued
  int lanes() const { return dvs_pipeline_lanes(h_); }   // 0: serial, 1: two-stream software pipeline, >= 2: lane schedule (small batches)
  int capacity() const { return capacity_; }   // rows per frame in every output block: nfeatures + 3 * nlevels
  dvs_pipeline_set results(long long step) const {
    dvs_pipeline_set s;
    check(dvs_pipeline_get_set(h_, step, &s), "dvs_pipeline_get_set");
    return s;
  }
  // host copies of one batch's results (after synchronize(), or at least after the set's events): convenience for hosts without HIP
  void download(long long step, std::vector<int32_t>& n, std::vector<dvs_keypoint>& kps, std::vector<uint8_t>& desc, std::vector<int32_t>* idx = nullptr,
                std::vector<int32_t>* dist = nullptr) const {
    const dvs_pipeline_set s = results(step);
    const size_t B = batch_, cap = capacity_;
    n.resize(B); kps.resize(B * cap); desc.resize(B * cap * 32);
    check(dvs_memcpy_d2h(device_, n.data(), s.d_n, B * 4), "d2h");
    check(dvs_memcpy_d2h(device_, kps.data(), s.d_kps, B * cap * sizeof(dvs_keypoint)), "d2h");
    check(dvs_memcpy_d2h(device_, desc.data(), s.d_desc, B * cap * 32), "d2h");
    if (idx) { idx->resize(B * cap); check(dvs_memcpy_d2h(device_, idx->data(), s.d_idx, B * cap * 4), "d2h"); }
    if (dist) { dist->resize(B * cap); check(dvs_memcpy_d2h(device_, dist->data(), s.d_dist, B * cap * 4), "d2h"); }
  }
  dvs_pipeline* handle() { return h_; }
  // measurement (bench-style per-stage report): every kernel alone on the main stream / per-stage GPU times of lane 0's extractor
  void setSerialized(bool on) { check(dvs_pipeline_set_serialized(h_, on ? 1 : 0), "dvs_pipeline_set_serialized"); }
  void stageTiming(bool on) { check(dvs_pipeline_stage_timing(h_, on ? 1 : 0), "dvs_pipeline_stage_timing"); }
  void stageTimes(double ms[DVS_STAGE_COUNT], int64_t calls[DVS_STAGE_COUNT], bool reset = true) {
    check(dvs_pipeline_get_stage_times(h_, ms, calls, reset ? 1 : 0), "dvs_pipeline_get_stage_times");
  }

 private:
  static void check(dvs_status s, const char* what) {
    if (s != DVS_OK) throw std::runtime_error(std::string(what) + ": " + dvs_last_error());
  }
  dvs_pipeline* h_ = nullptr;
  int device_ = 0, batch_ = 0, capacity_ = 0;
};

}  // namespace dvslam
