/*
 * dvslam_hip_test.h — test hooks of libdvslam_hip_test.so (the product library built with -DDVS_TEST_HOOKS; `make test-lib`).
 * NOT part of the product ABI: lib/libdvslam_hip.so does not export any of these.  They expose internals that the parity tests pin
 * one by one — the libstdc++ std::sort / nth_element / partition replicas, the glibc sinf / cosf restatement, the geometry tables,
 * the PnP stage's host-compiled minimal solvers, one scheduling aid (a kernel that holds a stream for a bounded time) — and the
 * extractor's scheduling / introspection hooks that the product library uses internally (dvs_pipeline_*) without exporting them.
 */
#ifndef DVSLAM_HIP_TEST_H
#define DVSLAM_HIP_TEST_H
#include "dvslam_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ======================================= host-logic test hooks ================================= */
/* (no GPU needed) libstdc++ std::sort replica used by the quad-tree, glibc sinf/cosf restatement, geometry tables */
void dvs_test_sort_nodes(const int32_t* count, const int32_t* ulx, int32_t n, int32_t* perm);
/* the same order through the rank-pairing restatement the quad-tree kernel runs (host, sequential) ... */
void dvs_test_sort_nodes_ranked(const int32_t* count, const int32_t* ulx, int32_t n, int32_t* perm);
/* ... and through the kernel's workgroup sort itself (needs a GPU; n <= 1500) */
dvs_status dvs_test_sort_nodes_device(const int32_t* count, const int32_t* ulx, int32_t n, int32_t* perm);
void dvs_test_sincosf(float a, float* s, float* c);
/* the PnP stage's quartic (Ferrari + Newton) and P3P (Grunert) routines on the host: real roots (unordered) / up to 4 poses x 12 */
int32_t dvs_test_quartic_roots(double a4, double a3, double a2, double a1, double a0, double* roots4);
int32_t dvs_test_p3p(const double* P9, const double* j9, double* poses48);
/* (needs a GPU) hold `stream` for the given time with one idle wavefront (<= 200 000 us): lets a test delay an event */
dvs_status dvs_test_stream_delay(void* stream, int32_t microseconds);
/* KeyPointsFilter::retainBest on bare responses: perm[i] = original index of the i-th survivor.  _host: csrc/lsort.h's sequential
 * restatement of std::nth_element + std::partition (no GPU); _device: the wavefront routine the cv::ORB kernels run */
void dvs_test_retain_best_host(const float* responses, int32_t n, int32_t n_points, int32_t* perm, int32_t* n_kept);
dvs_status dvs_test_retain_best_device(const float* responses, int32_t n, int32_t n_points, int32_t* perm, int32_t* n_kept);
dvs_status dvs_test_geometry(const dvs_orb_params* params, int32_t rows, int32_t cols, int32_t* level_w, int32_t* level_h,
                             int32_t* ncells, int32_t* quota, int32_t* wcell, int32_t* hcell);

/* ======================================= scheduling hooks of the extractor ====================== */
/* What csrc/pipeline.hip composes the streaming step from.  Until round 4 these were part of the product ABI; a maintainer calls
 * dvs_pipeline_* (or the plain extract / match entry points), so the product library keeps them internal (hidden visibility) and only
 * this test library exports them — for the tests that pin each hook against the plain calls (tests/test_gpu_orb.py) and for the
 * stage-by-stage parity checks (candidate lists, per-level keypoints). */
/* the same extractor with ONE stream for good: every stage runs in order on the handle's stream, no auxiliary / prefetch streams are
 * created (every HIP stream is a hardware queue, and a process has four) and dvs_orb_set_overlap(h, 1) is refused.  What the lanes of
 * dvs_pipeline are made of; results are identical. */
dvs_status dvs_orb_create_single_stream(const dvs_orb_params* params, int32_t device, dvs_orb** out);
/* ... and a single-stream extractor on a stream the CALLER owns (NULL: HIP's legacy default stream) — it never creates one of its own;
 * dvs_orb_use_own_stream is refused.  dvs_pipeline's fourth lane lives on a stream of another dispatch priority this way. */
dvs_status dvs_orb_create_on_stream(const dvs_orb_params* params, int32_t device, void* hip_stream, dvs_orb** out);

dvs_status dvs_orb_use_own_stream(dvs_orb* h);
/* 1 (default): independent stages overlap on an internal auxiliary stream (pyramid chain beside FAST, blur beside the quad-tree);
 * 0: every kernel runs alone on the handle's stream — what the per-kernel roofline durations are measured with */
dvs_status dvs_orb_set_overlap(dvs_orb* h, int32_t on);

/* Cross-batch software pipeline for streaming callers that already hold the next batch in device memory: announce it before
 * the dvs_orb_extract_batch_device call of the CURRENT batch.  That call then also enqueues the next batch's pyramid (same
 * nimg / rows / cols / step / frame_stride) on the handle's auxiliary stream, beside its own descriptor stage and whatever
 * the caller enqueues next (the match); the following call, if it is for exactly that buffer, finds its pyramid built and
 * starts with FAST on all levels at once.  One-shot; results are identical with or without the hint.  The announced images
 * must not change between the two calls.  (No counterpart in the reference, whose ComputePyramid runs inside operator(),
 * ORBextractor.cpp:1081; this is the MI355X replacement for running consecutive frames on separate CPU threads.) */
dvs_status dvs_orb_hint_next_batch_device(dvs_orb* h, const uint8_t* d_next_imgs);
/* Scheduling hook for a pipelined caller: `hip_event` (a hipEvent_t of the caller, NULL to clear) is recorded on the handle's main
 * stream by every following device-resident extraction right behind its FAST launch, i.e. at the point from which the machine's
 * vector ALUs are mostly idle (quad-tree / blur / descriptor gathers).  A caller that has independent matrix-core or copy work —
 * the PREVIOUS batch's match (BFMatcher call of frontend.cpp:1123) — makes its stream wait on it so that the work runs beside
 * that phase instead of beside FAST.  Results are unaffected. */
dvs_status dvs_orb_set_after_fast_event(dvs_orb* h, void* hip_event);
/* Output event of a pipelined caller: while `hip_event` (a hipEvent_t of the caller, NULL to clear) is set, every device-resident
 * extraction records it where its keypoints, descriptors and counts are complete (the library also uses it as the gate of the next
 * call's prefetch chain, which saves a record of its own).  With dvs_orb_set_defer_outputs(h, 1) that point is NOT the end of the
 * call on the main stream: the descriptor stage — fetch-bound gathers — stays on the handle's auxiliary stream without joining the
 * main one, so that the next call's FAST starts immediately and runs beside it; the library orders everything else (the next
 * quad-tree, blur and prefetch wait for it), the caller orders its consumers on the event and must leave the call's level-0 images
 * untouched until then.  dvs_orb_synchronize waits for a deferred stage too.  Results are unaffected. */
dvs_status dvs_orb_set_output_event(dvs_orb* h, void* hip_event);
dvs_status dvs_orb_set_defer_outputs(dvs_orb* h, int32_t on);
/* Reuse guard (one-shot, consumed by the next device-resident extraction): the call's OUTPUT buffers may still be read by work of
 * the caller on another stream (the match of an earlier batch); the extraction writes them only behind `hip_event`.  Same effect as
 * hipStreamWaitEvent on the main stream before the call, but the wait rides on the blur's stream, off the critical path in front of
 * FAST (outputs are only written by the descriptor stage, which joins the blur). */
dvs_status dvs_orb_set_reuse_guard_event(dvs_orb* h, void* hip_event);
/* Quad-tree off the main stream (pipelined callers: announced next batch + deferred outputs).  on = 1: the quad-tree of a call runs on
 * the auxiliary stream behind that call's FAST, so the next call's FAST follows immediately and the latency-bound tree runs beside it
 * (FAST writes three candidate-list sets in turn; with more than one workgroup per CU the tree is launched per level class with that
 * class's LDS footprint instead of level 0's).  Results are unaffected.  Synchronises the handle's streams. */
dvs_status dvs_orb_set_async_quadtree(dvs_orb* h, int32_t on);
/* ... and its descriptor stage on a stream of the caller (NULL: the auxiliary stream): with the quad-tree and the blur on the auxiliary
 * stream that stream alone would carry a whole step; a pipelined caller hands over its match stream (dvs_pipeline does). */
dvs_status dvs_orb_set_tail_stream(dvs_orb* h, void* hip_stream);
/* Diagnostics: how many announced level chains (dvs_orb_hint_next_batch_device) were enqueued as ONE hipGraphLaunch instead of one launch
 * per level.  The chain's arguments depend only on (source block, frame count, destination pyramid); the second time an argument set is
 * seen its chain is captured into a graph, from then on it is one runtime call (4 us of host time against 17).  Automatic up to 12 frames
 * per call, where the host's enqueue bounds the step (DVS_CHAIN_GRAPH=1 / 0: always / never). */
int64_t dvs_orb_chain_graph_launches(const dvs_orb* h);


/* FAST candidates handed to the quad-tree, in candidate order: int32 triplets (x, y, score), region-relative */
dvs_status dvs_orb_get_candidates(dvs_orb* h, int32_t frame, int32_t level, int32_t* xys, int32_t cap, int32_t* n);
/* per-level keypoints after the quad-tree, level coordinates: int32 triplets (x, y, score) in list order */
dvs_status dvs_orb_get_level_keypoints(dvs_orb* h, int32_t frame, int32_t level, int32_t* xys, int32_t cap, int32_t* n);

/* per-stage GPU timing with hipEvents on the handle's stream (stage ids: DVS_STAGE_* of dvslam_hip.h; the product reaches it through
 * dvs_pipeline_stage_timing / dvs_pipeline_get_stage_times) */
dvs_status dvs_orb_enable_stage_timing(dvs_orb* h, int32_t on);
/* accumulated milliseconds and launch-sequence counts per stage since the last reset; synchronises the stream */
dvs_status dvs_orb_get_stage_times(dvs_orb* h, double* ms, int64_t* calls, int32_t reset);


dvs_status dvs_matcher_use_own_stream(dvs_matcher* m);


/* the handles inside a dvs_pipeline (lane 0's), owned by the pipeline; its match stream */
dvs_orb* dvs_pipeline_extractor(dvs_pipeline* p);      /* lane 0's */
dvs_matcher* dvs_pipeline_matcher(dvs_pipeline* p);
void* dvs_pipeline_match_stream(dvs_pipeline* p);

#ifdef __cplusplus
}
#endif
#endif /* DVSLAM_HIP_TEST_H */
