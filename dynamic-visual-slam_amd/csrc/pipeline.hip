// pipeline.hip — dvs_pipeline_*: the streaming step (extraction of batch i + match of batch i - 1) as ONE C-ABI call per step.
// Host code only: the schedule over the extractor's / matcher's / communicator's entry points (orb.hip, match.hip, comm.hip) that
// bench.py times and tests/test_gpu_pipeline.py + tests/cpp/pipeline_stream.cpp check against the oracle.  The reference's
// counterpart is the frame loop of frontend.cpp:1084-1123 (extract, then match against the previous frame's descriptors).
//
// Step i of the pipelined schedule, software pipelined over two streams:
//   * main stream (the extractor's): FAST and quad-tree of batch i; its prefetch stream builds batch i + 1's pyramid beside FAST
//     (hint_next_batch), its auxiliary stream runs the blur and — deferred — the descriptor stage beside the next step's FAST;
//   * match stream: the B jobs of batch i - 1, released by the extractor's after-FAST event so that the matrix-core match runs
//     beside the quad-tree / blur phase; a communicator's boundary exchange shares this stream (the match is its only consumer);
//   * nsets output sets rotate: step i writes set i % nsets; its LAST reader is the match of batch i + 1 (frame 0 of batch i + 1
//     against the last frame of batch i), handed to the extractor as the reuse guard of step i + nsets.  That match is enqueued in
//     step i + 2, so the pipelined schedule needs nsets >= 3 (with two sets step i + 2 would overwrite the set the match enqueued
//     BEHIND it still reads: no event of that match exists yet when the extraction is enqueued).
//
// Lane schedule (small batches, dvs_pipeline_params::lanes): within one step of a few frames the machine is mostly idle and the step
// takes the LATENCY of its kernel chain (one 720p frame: ~0.1 ms; the level-0 quad-tree workgroup alone 0.07-0.08), so instead of
// overlapping the stages of one step the schedule overlaps whole steps: `lanes` extractor / matcher pairs with one stream each, step i
// on lane i % lanes, every stage serial on that stream (dvs_orb_set_overlap(0): no forks, no barrier packets inside), its match
// right behind its extraction.  Cross-lane order only where data flows: match i waits for batch i - 1's output event; step i + nsets
// waits for the matches of batches i and i + 1 (the readers of set i).
#include <vector>
#include "common.h"

using namespace dvs;

struct dvs_pipeline {
  int device = 0, B = 0, rows = 0, cols = 0, nsets = 0, cap = 0;
  bool pipelined = true;
  int lanes = 1;                          // >= 2: the lane schedule
  bool quadtree_async = false;            // the four-stream form of the two-stream pipeline
  dvs_orb* orb = nullptr;                 // = orbs[0]
  dvs_matcher* mat = nullptr;             // = mats[0]
  std::vector<dvs_orb*> orbs;             // one extractor / matcher pair per lane (one pair for the other schedules)
  std::vector<dvs_matcher*> mats;
  dvs_comm* comm = nullptr;
  hipStream_t T = nullptr, M = nullptr;   // lane 0's main stream; match stream (= T for the serial and the lane schedule)
  bool own_M = false;
  hipStream_t lane4 = nullptr;            // the fourth lane's stream (highest dispatch priority), owned here
  uint8_t* arena = nullptr;               // all output sets in one allocation
  std::vector<dvs_keypoint*> kps;
  std::vector<uint8_t*> desc;
  std::vector<int32_t*> n, idx, dist;
  std::vector<hipEvent_t> ev_ext, ev_match;
  hipEvent_t ev_fast = nullptr;
  int64_t i = 0;
  int64_t matched = -1;                   // pipelined schedule: the last batch whose match has been enqueued (a flush enqueues it early)
};

namespace {

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// enqueue the B match jobs of batch j on the match stream (its extraction is ordered by events)
dvs_status enqueue_match(dvs_pipeline* p, int64_t j, bool behind_fast) {
  const int sj = (int)(j % p->nsets), B = p->B, cap = p->cap;
  const uint8_t* prev_desc = nullptr;
  const int32_t* prev_n = nullptr;
  if (p->lanes >= 2) {
    // lane schedule: batch j's extraction precedes on this lane's stream; the predecessor frame was extracted on another lane
    const int l = (int)(j % p->lanes);
    hipStream_t S = (hipStream_t)dvs_orb_get_stream(p->orbs[l]);
    const uint8_t* last_desc = p->desc[sj] + (size_t)(B - 1) * cap * 32;
    const int32_t* last_n = p->n[sj] + (B - 1);
    if (p->comm) {
      // the communicator's three gather buffers rotate: this call packs into the buffer of call j - 3, whose readers were the matches of
      // batches j - 3 (this lane or an earlier event of its lane) and j - 2 (rank 0's predecessor block)
      if (j >= 3) DVS_HIP(hipStreamWaitEvent(S, p->ev_match[(j - 3) % p->nsets], 0));
      if (j >= 2) DVS_HIP(hipStreamWaitEvent(S, p->ev_match[(j - 2) % p->nsets], 0));
      DVS_TRY(dvs_exchange_boundary(p->comm, S, last_desc, last_n, cap, &prev_desc, &prev_n));
    } else if (j > 0) {
      const int sp = (int)((j - 1) % p->nsets);
      DVS_HIP(hipStreamWaitEvent(S, p->ev_ext[sp], 0));
      prev_desc = p->desc[sp] + (size_t)(B - 1) * cap * 32;
      prev_n = p->n[sp] + (B - 1);
    }
    DVS_TRY(dvs_match_hamming_sequence_device(p->mats[l], p->desc[sj], p->n[sj], cap, B, prev_desc, prev_n, p->idx[sj], p->dist[sj]));
    DVS_HIP(hipEventRecord(p->ev_match[sj], S));
    return DVS_OK;
  }
  if (p->pipelined) DVS_HIP(hipStreamWaitEvent(p->M, p->ev_ext[sj], 0));
  const uint8_t* last_desc = p->desc[sj] + (size_t)(B - 1) * cap * 32;
  const int32_t* last_n = p->n[sj] + (B - 1);
  if (p->comm) {
    // the one exchange step, once per global batch: every rank's LAST frame of batch j; this rank's first frame is matched against the
    // frame before it in the global order — the previous rank's last frame of the same batch, or (rank 0) the last rank's of the batch
    // before.  Depends only on batch j's extraction; shares the match stream.
    DVS_TRY(dvs_exchange_boundary(p->comm, p->M, last_desc, last_n, cap, &prev_desc, &prev_n));
  } else if (j > 0) {
    const int sp = (int)((j - 1) % p->nsets);   // one GPU: the previous batch's last frame, read in place
    prev_desc = p->desc[sp] + (size_t)(B - 1) * cap * 32;
    prev_n = p->n[sp] + (B - 1);
  }
  // released behind the FAST of the step just enqueued (four-stream form: that step's descriptor stage precedes on this very stream and
  // followed its FAST — no wait, one runtime call and one barrier packet less per step)
  if (behind_fast && !p->quadtree_async) DVS_HIP(hipStreamWaitEvent(p->M, p->ev_fast, 0));
  DVS_TRY(dvs_match_hamming_sequence_device(p->mat, p->desc[sj], p->n[sj], cap, B, prev_desc, prev_n, p->idx[sj], p->dist[sj]));
  DVS_HIP(hipEventRecord(p->ev_match[sj], p->M));
  return DVS_OK;
}

}  // namespace

extern "C" {

dvs_status dvs_pipeline_create(const dvs_pipeline_params* prm, int32_t device, dvs_pipeline** out) {
  DVS_ARG(prm && out);
  *out = nullptr;
  DVS_ARG(prm->quadtree_async >= -1 && prm->quadtree_async <= 1);
  DVS_ARG(prm->batch >= 1 && prm->rows > 0 && prm->cols > 0 && prm->nsets >= 0 && prm->lanes >= 0 && prm->lanes <= DVS_PIPELINE_MAX_LANES);
  int lanes = !prm->pipelined ? 1 : (prm->lanes ? prm->lanes : (prm->batch <= DVS_PIPELINE_LANE_BATCH ? DVS_PIPELINE_MAX_LANES : 1));
  // lanes steps are in flight and the match of the oldest still reads the set before it: two sets per lane keep every lane busy
  const int nsets = prm->nsets ? prm->nsets : (lanes >= 2 ? 2 * lanes : 4);
  // a set always belongs to the same lane (nsets a multiple of lanes): whatever was enqueued earlier for a set — its extraction, its match,
  // their events — precedes on that lane's stream
  while (lanes > 1 && nsets % lanes) lanes--;
  if (!prm->pipelined && nsets < 2) {
    set_error("the serial schedule rotates at least 2 output sets (frame 0 of batch i is matched against the last frame of batch "
              "i - 1, which one set would have overwritten); nsets = %d", nsets);
    return DVS_ERR_ARG;
  }
  if (prm->pipelined && nsets < 3) {
    set_error("the pipelined schedule rotates at least 3 output sets (the match of batch i + 1 reads batch i's last frame and is "
              "enqueued in step i + 2); nsets = %d", nsets);
    return DVS_ERR_ARG;
  }
  DVS_TRY(check_device(device));
  dvs_pipeline* p = new dvs_pipeline();
  p->device = device; p->B = prm->batch; p->rows = prm->rows; p->cols = prm->cols; p->nsets = nsets; p->pipelined = prm->pipelined != 0;
  p->lanes = lanes;
  dvs_status st = DVS_OK;
  auto fail = [&](dvs_status s) { dvs_pipeline_destroy(p); return s; };
  dvs_orb_params op = prm->orb;
  op.max_batch = prm->batch;
  for (int l = 0; l < lanes; l++) {
    dvs_orb* o = nullptr;
    // a lane runs every stage on its one stream: no auxiliary / prefetch streams (every HIP stream is a hardware queue).  Streams of one
    // dispatch priority share four hardware queues and the process's default stream holds one of them: a fourth lane on a stream of the
    // same priority shares a queue with another lane (0.088 against 0.057 ms per 1-frame step), on a stream of ANOTHER priority it has a
    // queue to itself (0.040 ms).  A fifth lane of any priority collapses all of them (0.074-0.097 ms): four queues run at a time.
    if (lanes >= 2 && l == 3) {
      void* s4 = nullptr;
      if ((st = dvs_stream_create(device, 1, &s4)) != DVS_OK) return fail(st);
      p->lane4 = (hipStream_t)s4;
      if ((st = dvs_orb_create_on_stream(&op, device, s4, &o)) != DVS_OK) return fail(st);
    } else if ((st = lanes >= 2 ? dvs_orb_create_single_stream(&op, device, &o) : dvs_orb_create(&op, device, &o)) != DVS_OK) {
      return fail(st);
    }
    p->orbs.push_back(o);
  }
  p->orb = p->orbs[0];
  p->cap = dvs_orb_max_keypoints(p->orb);
  // streams are created only when used, back to back and before any communicator comes up: every HIP stream is a hardware queue (an
  // idle fourth stream in the extractor handle cost 0.2 ms per step; RCCL initialised first moved the same job between 48 k and 72 k
  // frames/s depending on GPU_MAX_HW_QUEUES)
  p->T = (hipStream_t)dvs_orb_get_stream(p->orb);
  p->M = p->T;
  if (p->pipelined && lanes == 1) {
    void* s = nullptr;
    if ((st = dvs_stream_create(device, 0, &s)) != DVS_OK) return fail(st);
    p->M = (hipStream_t)s; p->own_M = true;
  }
  for (int l = 0; l < lanes; l++) {
    dvs_matcher* m = nullptr;
    if ((st = dvs_matcher_create_on_stream(device, lanes >= 2 ? dvs_orb_get_stream(p->orbs[l]) : (void*)p->M, &m)) != DVS_OK) return fail(st);
    p->mats.push_back(m);
  }
  p->mat = p->mats[0];
  const size_t B = p->B, cap = p->cap;
  const size_t b_kps = up256(B * cap * sizeof(dvs_keypoint)), b_desc = up256(B * cap * 32), b_n = up256(B * 4), b_idx = up256(B * cap * 4);
  const size_t per_set = b_kps + b_desc + b_n + 2 * b_idx;
  if (hipMalloc((void**)&p->arena, per_set * nsets) != hipSuccess) { set_error("dvs_pipeline_create: %zu bytes of output sets", per_set * nsets); return fail(DVS_ERR_HIP); }
  // counts and descriptor rows start at zero: the match of a set that was never extracted (or a row past a frame's count) reads zeros
  if (hipMemset(p->arena, 0, per_set * nsets) != hipSuccess) return fail(DVS_ERR_HIP);
  for (int s = 0; s < nsets; s++) {
    uint8_t* q = p->arena + per_set * s;
    p->kps.push_back((dvs_keypoint*)q); q += b_kps;
    p->desc.push_back(q); q += b_desc;
    p->n.push_back((int32_t*)q); q += b_n;
    p->idx.push_back((int32_t*)q); q += b_idx;
    p->dist.push_back((int32_t*)q);
    hipEvent_t e1 = nullptr, e2 = nullptr;
    if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) return fail(DVS_ERR_HIP);
    p->ev_ext.push_back(e1); p->ev_match.push_back(e2);   // batch's outputs complete (recorded by the extractor) / batch's match complete
  }
  if (hipEventCreateWithFlags(&p->ev_fast, hipEventDisableTiming) != hipSuccess) return fail(DVS_ERR_HIP);
  if (p->pipelined && lanes == 1) {
    if ((st = dvs_orb_set_after_fast_event(p->orb, p->ev_fast)) != DVS_OK) return fail(st);
  }
  // the four-stream form for batches whose kernels do not fill the machine (main: blur + FAST, prefetch: level chain, auxiliary:
  // quad-tree, match stream: descriptors + match)
  p->quadtree_async = p->pipelined && lanes == 1 &&
                      (prm->quadtree_async > 0 || (prm->quadtree_async == 0 && prm->batch > DVS_PIPELINE_LANE_BATCH && prm->batch <= DVS_PIPELINE_ASYNC_BATCH));
  if (p->quadtree_async) {
    if ((st = dvs_orb_set_async_quadtree(p->orb, 1)) != DVS_OK) return fail(st);
    if ((st = dvs_orb_set_tail_stream(p->orb, p->M)) != DVS_OK) return fail(st);
  }
  if ((st = dvs_orb_set_output_event(p->orb, p->ev_ext[0])) != DVS_OK) return fail(st);
  if ((st = dvs_orb_set_defer_outputs(p->orb, p->pipelined && lanes == 1 ? 1 : 0)) != DVS_OK) return fail(st);
  *out = p;
  return DVS_OK;
}

void dvs_pipeline_destroy(dvs_pipeline* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  for (dvs_orb* o : p->orbs) (void)dvs_orb_synchronize(o);
  if (p->M) (void)hipStreamSynchronize(p->M);
  for (dvs_orb* o : p->orbs) {
    (void)dvs_orb_set_output_event(o, nullptr);
    (void)dvs_orb_set_defer_outputs(o, 0);
    (void)dvs_orb_set_after_fast_event(o, nullptr);
  }
  for (hipEvent_t e : p->ev_ext) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_match) (void)hipEventDestroy(e);
  if (p->ev_fast) (void)hipEventDestroy(p->ev_fast);
  for (dvs_matcher* m : p->mats) dvs_matcher_destroy(m);
  if (p->own_M && p->M) (void)dvs_stream_destroy(p->M);
  for (dvs_orb* o : p->orbs) dvs_orb_destroy(o);
  if (p->lane4) (void)dvs_stream_destroy(p->lane4);
  if (p->arena) (void)hipFree(p->arena);
  delete p;
}

dvs_status dvs_pipeline_attach_comm(dvs_pipeline* p, dvs_comm* comm) {
  DVS_ARG(p);
  if (comm && dvs_comm_is_host(comm)) {
    set_error("dvs_pipeline_attach_comm: a host-transport communicator exchanges host blocks; the pipeline's are device memory");
    return DVS_ERR_ARG;
  }
  p->comm = comm;
  return DVS_OK;
}

dvs_status dvs_pipeline_step(dvs_pipeline* p, const uint8_t* d_imgs, const uint8_t* d_next_imgs, int32_t flags) {
  DVS_ARG(p && d_imgs);
  const int64_t i = p->i++;
  const int s = (int)(i % p->nsets);
  if (p->lanes >= 2) {
    dvs_orb* o = p->orbs[i % p->lanes];
    if (i >= p->nsets) {
      // set s was written by step i - nsets and read by the matches of batches i - nsets (on this lane: nsets is a multiple of lanes) and
      // i - nsets + 1 (another lane)
      DVS_TRY(dvs_orb_set_reuse_guard_event(o, p->ev_match[(i - p->nsets + 1) % p->nsets]));
    }
    DVS_TRY(dvs_orb_set_output_event(o, p->ev_ext[s]));
    DVS_TRY(dvs_orb_extract_batch_device(o, d_imgs, p->B, p->rows, p->cols, (size_t)p->cols, (size_t)p->rows * p->cols, p->kps[s], p->desc[s], p->cap,
                                         p->n[s]));
    if (!(flags & DVS_PIPELINE_NO_MATCH)) DVS_TRY(enqueue_match(p, i, false));
    return DVS_OK;
  }
  if (p->pipelined) {
    // the last reader of the set this step overwrites: the match of batch i - nsets + 1
    if (i >= p->nsets) DVS_TRY(dvs_orb_set_reuse_guard_event(p->orb, p->ev_match[(i - p->nsets + 1) % p->nsets]));
    if (d_next_imgs) DVS_TRY(dvs_orb_hint_next_batch_device(p->orb, d_next_imgs));
  }
  DVS_TRY(dvs_orb_set_output_event(p->orb, p->ev_ext[s]));
  DVS_TRY(dvs_orb_extract_batch_device(p->orb, d_imgs, p->B, p->rows, p->cols, (size_t)p->cols, (size_t)p->rows * p->cols, p->kps[s], p->desc[s],
                                       p->cap, p->n[s]));
  if (flags & DVS_PIPELINE_NO_MATCH) return DVS_OK;
  if (p->pipelined) {
    // (a flush may have enqueued it already: a second exchange for the same batch would hand rank 0 that batch as its predecessor)
    if (i >= 1 && p->matched < i - 1) { DVS_TRY(enqueue_match(p, i - 1, true)); p->matched = i - 1; }
  } else {
    DVS_TRY(enqueue_match(p, i, false));
  }
  return DVS_OK;
}

dvs_status dvs_pipeline_flush(dvs_pipeline* p) {
  DVS_ARG(p);
  if (p->pipelined && p->lanes == 1 && p->i >= 1 && p->matched < p->i - 1) {
    DVS_TRY(enqueue_match(p, p->i - 1, false));
    p->matched = p->i - 1;
  }
  return DVS_OK;
}

dvs_status dvs_pipeline_synchronize(dvs_pipeline* p) {
  DVS_ARG(p);
  for (dvs_orb* o : p->orbs) DVS_TRY(dvs_orb_synchronize(o));
  DVS_HIP(hipStreamSynchronize(p->M));
  return DVS_OK;
}

dvs_status dvs_pipeline_reset(dvs_pipeline* p) {
  DVS_ARG(p);
  DVS_TRY(dvs_pipeline_synchronize(p));
  p->i = 0;
  p->matched = -1;
  if (p->comm) DVS_TRY(dvs_comm_reset_sequence(p->comm));   // batch 0 of the next run has no predecessor on rank 0 either
  return DVS_OK;
}

int64_t dvs_pipeline_steps(const dvs_pipeline* p) { return p ? p->i : 0; }

dvs_status dvs_pipeline_get_set(const dvs_pipeline* p, int64_t step, dvs_pipeline_set* out) {
  DVS_ARG(p && out && step >= 0);
  const int s = (int)(step % p->nsets);
  out->d_kps = p->kps[s]; out->d_desc = p->desc[s]; out->d_n = p->n[s]; out->d_idx = p->idx[s]; out->d_dist = p->dist[s];
  out->ev_extracted = p->ev_ext[s]; out->ev_matched = p->ev_match[s]; out->capacity = p->cap;
  return DVS_OK;
}

dvs_status dvs_pipeline_set_serialized(dvs_pipeline* p, int32_t on) {
  DVS_ARG(p);
  if (p->lanes >= 2) { set_error("dvs_pipeline_set_serialized: the lane schedule's extractors have one stream for good (nothing to serialise)"); return DVS_ERR_UNSUPPORTED; }
  DVS_TRY(dvs_pipeline_synchronize(p));
  return dvs_orb_set_overlap(p->orb, on ? 0 : 1);
}

dvs_status dvs_pipeline_stage_timing(dvs_pipeline* p, int32_t on) {
  DVS_ARG(p);
  return dvs_orb_enable_stage_timing(p->orb, on);
}

dvs_status dvs_pipeline_get_stage_times(dvs_pipeline* p, double* ms, int64_t* calls, int32_t reset) {
  DVS_ARG(p);
  return dvs_orb_get_stage_times(p->orb, ms, calls, reset);
}

int32_t dvs_pipeline_quadtree_async(const dvs_pipeline* p) { return p && p->quadtree_async ? 1 : 0; }
int32_t dvs_pipeline_nsets(const dvs_pipeline* p) { return p ? p->nsets : 0; }
int32_t dvs_pipeline_lanes(const dvs_pipeline* p) { return !p ? 0 : (p->pipelined ? p->lanes : 0); }
DVS_HOOK dvs_orb* dvs_pipeline_extractor(dvs_pipeline* p) { return p ? p->orb : nullptr; }
DVS_HOOK dvs_matcher* dvs_pipeline_matcher(dvs_pipeline* p) { return p ? p->mat : nullptr; }
DVS_HOOK void* dvs_pipeline_match_stream(dvs_pipeline* p) { return p ? (void*)p->M : nullptr; }

}  // extern "C"
