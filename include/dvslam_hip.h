/*
 * dvslam_hip.h — C-ABI of libdvslam_hip.so: MI355X (gfx950) implementation of the data-parallel hot
 * path of andrewkwolek/dynamic-visual-slam.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference has no FFI/plugin layer: its hot path is reached through three C++ call signatures
 * (SURVEY.md §8b).  Each group of entry points below replaces one of them; header-only C++ adapters
 * with the reference's own signatures live in include/dvslam/ (see INTEGRATION.md).
 *
 *   B1  ORB_SLAM3::ORBextractor::ORBextractor / operator()   include/dynamic_visual_slam/ORBextractor.hpp:50-60,
 *                                                            src/ORBextractor.cpp:409-469, 1086-1167
 *   B2  cv::BFMatcher(NORM_HAMMING).match call sites          src/frontend.cpp:220,614,1123; src/backend.cpp:222,1072
 *   B3  SlidingWindowBA / WeightedSquaredReprojectionError    include/dynamic_visual_slam/bundle_adjustment.hpp:469-593, 652-904
 *
 * Conventions: every function returns a dvs_status (0 = ok, < 0 = error; dvs_last_error() gives text);
 * nothing is allocated across the ABI — callers supply output buffers with explicit capacities;
 * a handle owns one HIP stream and is not thread-safe, distinct handles are independent.
 * "host" entry points take host pointers (they stage through pinned memory); "_device" entry points
 * take pointers into the handle's GPU memory space and only enqueue work on the handle's stream.
 */
#ifndef DVSLAM_HIP_H
#define DVSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t dvs_status;
enum {
  DVS_OK = 0,
  DVS_ERR_EMPTY = -1,        /* empty image: ORBextractor::operator() returns -1 (ORBextractor.cpp:1090-1091) */
  DVS_ERR_UNSUPPORTED = -2,  /* image size / parameters for which the reference divides by zero (nCols, nIni == 0) */
  DVS_ERR_CAPACITY = -3,     /* caller buffer too small */
  DVS_ERR_HIP = -4,          /* HIP runtime error (message in dvs_last_error) */
  DVS_ERR_NO_DEVICE = -5,    /* no gfx950 device visible: there is NO CPU fallback */
  DVS_ERR_ARG = -6           /* null pointer / negative size / bad enum */
};

const char* dvs_last_error(void);
int32_t dvs_device_count(void);
/* A plain HIP stream for the caller's side work (e.g. the multi-GPU boundary exchange), created by this library so that it
 * can be made right after the handles: HIP maps streams to hardware queues in creation order, and the path's concurrent
 * streams should be neighbours in that order ("Streams and hardware queues"). */
dvs_status dvs_stream_create(int32_t device, int32_t high_priority, void** out_stream);
dvs_status dvs_stream_synchronize(void* stream);
dvs_status dvs_stream_destroy(void* stream);
/* hipEvent_t helpers for callers without a HIP toolchain of their own (the scheduling hooks dvs_orb_set_after_fast_event /
 * dvs_orb_set_output_event take any hipEvent_t): create (timing disabled), destroy, host wait, record, stream wait */
dvs_status dvs_event_create(int32_t device, void** out_event);
dvs_status dvs_event_destroy(void* event);
dvs_status dvs_event_synchronize(void* event);
dvs_status dvs_event_create_timing(int32_t device, void** out_event);   /* an event that records a time stamp */
dvs_status dvs_event_elapsed_ms(void* start, void* stop, float* ms);    /* both created with _create_timing and completed */
dvs_status dvs_event_query(void* event, int32_t* done);   /* *done = 1 when everything recorded before the event has completed */
dvs_status dvs_event_record(void* event, void* stream);
dvs_status dvs_stream_wait_event(void* stream, void* event);
/* "gfx950" etc. for the given device, "" on error */
dvs_status dvs_device_arch(int32_t device, char* buf, int32_t cap);

/* ---- device memory / stream helpers so non-HIP hosts (tests, bench) can stay resident in HBM ---- */
dvs_status dvs_malloc(int32_t device, size_t bytes, void** out);
dvs_status dvs_free(int32_t device, void* p);
dvs_status dvs_memcpy_h2d(int32_t device, void* dst, const void* src, size_t bytes);
dvs_status dvs_memcpy_d2h(int32_t device, void* dst, const void* src, size_t bytes);
dvs_status dvs_memset(int32_t device, void* dst, int value, size_t bytes);

/* ======================================= B1: ORB extractor ===================================== */

typedef struct dvs_orb dvs_orb;

/* same field order and size (28 B) as cv::KeyPoint */
typedef struct dvs_keypoint {
  float x, y;       /* level-0 pixel coordinates (level coords * mvScaleFactor[octave], ORBextractor.cpp:1148-1150) */
  float size;       /* (int)(31 * mvScaleFactor[octave]) (ORBextractor.cpp:880,889) */
  float angle;      /* degrees in [0,360), intensity-centroid orientation (ORBextractor.cpp:76-103) */
  float response;   /* FAST-9/16 corner score */
  int32_t octave;   /* pyramid level */
  int32_t class_id; /* -1 */
} dvs_keypoint;

typedef struct dvs_orb_params {
  int32_t nfeatures;        /* ORBextractor ctor arg 1 (frontend.cpp:206: 1000) */
  float scale_factor;       /* arg 2 (1.2f) */
  int32_t nlevels;          /* arg 3 (8); 1..DVS_MAX_LEVELS */
  int32_t ini_th_fast;      /* arg 4 (20) */
  int32_t min_th_fast;      /* arg 5 (7) */
  int32_t gauss_kernel[7];  /* Q8 7-tap kernel of cv::GaussianBlur(7x7, sigma 2); all zeros = {18,34,48,56,48,34,18} */
  int32_t max_batch;        /* frames processed per launch sequence; 0 = 1 */
} dvs_orb_params;

#define DVS_MAX_LEVELS 16

dvs_status dvs_orb_create(const dvs_orb_params* params, int32_t device, dvs_orb** out);
void dvs_orb_destroy(dvs_orb* h);
/* capacity a caller must provide per frame: nfeatures + 3 * nlevels (a level may return quota + 2, ORBextractor.cpp:746-747) */
int32_t dvs_orb_max_keypoints(const dvs_orb* h);
/* enqueue on a caller-owned hipStream_t (e.g. torch's current stream) instead of the handle's own non-blocking
 * stream; NULL selects HIP's legacy default stream. */
dvs_status dvs_orb_set_stream(dvs_orb* h, void* hip_stream);
void* dvs_orb_get_stream(dvs_orb* h);
dvs_status dvs_orb_synchronize(dvs_orb* h);

/* float tables of the ctor (ORBextractor.cpp:414-445, 451-468); arrays of nlevels (umax: 16) entries, any may be NULL */
dvs_status dvs_orb_get_tables(const dvs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                              int32_t* features_per_level, int32_t* umax16);
dvs_status dvs_orb_level_size(const dvs_orb* h, int32_t rows, int32_t cols, int32_t level, int32_t* level_rows, int32_t* level_cols);

/* operator(): host image (8UC1, `step` bytes between rows) -> host keypoints + N x 32 descriptors.  *n_out = N. */
dvs_status dvs_orb_extract(dvs_orb* h, const uint8_t* gray, int32_t rows, int32_t cols, size_t step,
                           dvs_keypoint* kps, uint8_t* desc, int32_t capacity, int32_t* n_out);
/* nimg host images of identical size; outputs are [nimg][capacity] blocks, n_out[nimg] */
dvs_status dvs_orb_extract_batch(dvs_orb* h, const uint8_t* const* imgs, int32_t nimg, int32_t rows, int32_t cols, size_t step,
                                 dvs_keypoint* kps, uint8_t* desc, int32_t capacity, int32_t* n_out);
/* device-resident batch: frame f starts at d_imgs + f * frame_stride.  Asynchronous on the handle's stream. */
dvs_status dvs_orb_extract_batch_device(dvs_orb* h, const uint8_t* d_imgs, int32_t nimg, int32_t rows, int32_t cols,
                                        size_t step, size_t frame_stride, dvs_keypoint* d_kps, uint8_t* d_desc,
                                        int32_t capacity, int32_t* d_n_out);
/* ---- level-sharded extraction for SMALL batches on several GPUs (SURVEY.md §8e "Partitioning") --------------------------------
 * With fewer frames in flight than GPUs, frame sharding leaves GPUs idle; the stages after the pyramid are independent per
 * level (ORBextractor.cpp:787, 894, 1123), so every rank takes the same frames and a subset of the LEVELS: it rebuilds the
 * (cheap) level chain up to its highest level, runs FAST / quad-tree / blur / descriptors for its levels only and writes a
 * level-slotted block {int32 counts[nimg][nlevels]; dvs_keypoint[nimg][K]; uint8 desc[nimg][K][32]}, K = sum(quota_l + 4), in which
 * level l owns fixed slots.  One all-gather of the blocks (dvs_comm_all_gather) and dvs_orb_merge_levels_device then restore
 * the reference's level-major output on every rank — bit-identical to dvs_orb_extract_batch_device.  level_mask: bit l = this
 * rank owns level l (dvslam_amd.dist.level_shards balances them by pixel count). */
size_t dvs_orb_level_block_bytes(const dvs_orb* h, int32_t nimg);
dvs_status dvs_orb_extract_levels_device(dvs_orb* h, const uint8_t* d_imgs, int32_t nimg, int32_t rows, int32_t cols, size_t step,
                                         size_t frame_stride, uint32_t level_mask, uint8_t* d_block /* 16-byte aligned, level_block_bytes */);
/* d_blocks: [world][dvs_orb_level_block_bytes] as gathered; level_owner[nlevels] (host): the rank whose block holds level l */
dvs_status dvs_orb_merge_levels_device(dvs_orb* h, const uint8_t* d_blocks, int32_t world, const int32_t* level_owner, int32_t nimg,
                                       dvs_keypoint* d_kps, uint8_t* d_desc, int32_t capacity, int32_t* d_n_out);

/* the pyramid of the LAST extract call (mvImagePyramid is a public member, ORBextractor.hpp:84); blurred = 1: the blurred levels */
dvs_status dvs_orb_get_level(dvs_orb* h, int32_t frame, int32_t level, int32_t blurred, uint8_t* dst, int32_t cap_bytes);
/* ======================================= B2: Hamming matcher =================================== */

typedef struct dvs_matcher dvs_matcher;
dvs_status dvs_matcher_create(int32_t device, dvs_matcher** out);
/* the same on a caller-owned hipStream_t from the start (NULL = the legacy default stream): the handle then never creates a
 * stream of its own.  Every HIP stream is a hardware queue; idle ones are not free ("Streams and hardware queues") */
dvs_status dvs_matcher_create_on_stream(int32_t device, void* hip_stream, dvs_matcher** out);
void dvs_matcher_destroy(dvs_matcher* m);
dvs_status dvs_matcher_set_stream(dvs_matcher* m, void* hip_stream);
dvs_status dvs_matcher_synchronize(dvs_matcher* m);

/* BFMatcher(NORM_HAMMING).match(query, train): per query row the arg-min Hamming distance over train rows,
 * lowest train index on ties; train_idx = -1 and dist = INT32_MAX when nt == 0.  Rows are 32 bytes. Host pointers. */
dvs_status dvs_match_hamming(dvs_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt,
                             int32_t* train_idx, int32_t* dist);
/* npairs independent jobs, device-resident: job p uses rows [0, d_nq[p]) of d_q + p*q_stride_rows*32 against rows
 * [0, d_nt[p]) of d_t + p*t_stride_rows*32; outputs at d_idx/d_dist + p*q_stride_rows.  Asynchronous. */
dvs_status dvs_match_hamming_batch_device(dvs_matcher* m, const uint8_t* d_q, const int32_t* d_nq, int32_t q_stride_rows,
                                          const uint8_t* d_t, const int32_t* d_nt, int32_t t_stride_rows, int32_t npairs,
                                          int32_t* d_idx, int32_t* d_dist);
/* The frontend's pattern (frontend.cpp:1096: current frame against the previous one) over a device-resident run of frames:
 * frame p (rows [0, d_n[p]) of d_desc + p*stride_rows*32) is matched against frame p-1; frame 0 against the caller's
 * predecessor (d_prev_desc / d_prev_n: e.g. the last frame of the previous batch, wherever it lives — no copy), or against
 * nothing when both are NULL.  Outputs at d_idx/d_dist + p*stride_rows.  Asynchronous. */
dvs_status dvs_match_hamming_sequence_device(dvs_matcher* m, const uint8_t* d_desc, const int32_t* d_n, int32_t stride_rows,
                                             int32_t nframes, const uint8_t* d_prev_desc, const int32_t* d_prev_n, int32_t* d_idx,
                                             int32_t* d_dist);
/* backend.cpp:1068-1077 shape: every (query, train) pair with distance < max_dist, (query, train)-ordered int32
 * triplets (q, t, dist).  *n_pairs = total found (may exceed cap; only the first cap are written). Host pointers. */
dvs_status dvs_match_hamming_thresh(dvs_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt,
                                    int32_t max_dist, int32_t* pairs, int32_t cap, int32_t* n_pairs);

/* ======================= E: the multi-GPU exchange step ========================== */
/* One process per GPU; frames (or, for small batches, pyramid levels) are sharded over the ranks and extraction needs no
 * communication.  The match job (t, t-1) at a shard boundary needs the previous rank's last-frame descriptors: ONE
 * ncclAllGather of fixed-size blocks per step, over RCCL/xGMI, on the caller's stream.  The reference has no counterpart
 * (single process per node; frontend.cpp:1096 keeps prev_descriptors_ in host memory).  RCCL is dlopen'ed at first use.
 *
 * Bring-up: rank 0 calls dvs_comm_get_unique_id and hands the 128 bytes to the other ranks out of band (MPI_Bcast, a TCP
 * store, a file); every rank then calls dvs_comm_create (collective, blocks until all ranks arrived). */
typedef struct dvs_comm dvs_comm;
#define DVS_COMM_ID_BYTES 128
dvs_status dvs_comm_get_unique_id(uint8_t* id /* [DVS_COMM_ID_BYTES] */);
dvs_status dvs_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t* id, dvs_comm** out);
/* Loopback group for one-GPU rehearsals and tests: out[0 .. world) = `world` logical ranks of THIS process on ONE device, no RCCL.
 * Each rank must be driven by its own host thread with its own streams; a collective call (dvs_exchange_boundary,
 * dvs_comm_all_gather) blocks on the host until every rank of the group has made the same call (30 s, then it fails on all ranks),
 * then pulls the peers' blocks with device-to-device copies behind their events.  Same results as the RCCL communicator. */
#define DVS_COMM_MAX_LOOPBACK 16
dvs_status dvs_comm_create_loopback(int32_t device, int32_t world, dvs_comm** out /* [world] */);
/* Host-transport communicator: the same exchange step for a host that moves the blocks itself (MPI_Allgather, sockets, a
 * torch.distributed group).  `all_gather(user, send, recv, bytes)` gathers `bytes` from every rank into recv[rank * bytes] on HOST memory
 * and returns 0 on success; it is called in place (send == recv + this rank's slot).  With such a communicator dvs_exchange_boundary /
 * dvs_comm_all_gather take and return HOST pointers, ignore `stream`, run synchronously and touch no device: the rank logic — buffer
 * rotation, this rank's slot, the predecessor with its wrap-around to the previous call — is the code the RCCL communicator runs.
 * Not accepted by dvs_pipeline_attach_comm (its blocks are device memory). */
typedef int (*dvs_host_all_gather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
dvs_status dvs_comm_create_host(int32_t rank, int32_t world, dvs_host_all_gather_fn all_gather, void* user, dvs_comm** out);
int32_t dvs_comm_is_host(const dvs_comm* c);
/* the next dvs_exchange_boundary starts a new sequence (rank 0: no predecessor).  Call with the streams of earlier calls drained. */
dvs_status dvs_comm_reset_sequence(dvs_comm* c);
void dvs_comm_destroy(dvs_comm* c);
int32_t dvs_comm_rank(const dvs_comm* c);
int32_t dvs_comm_world(const dvs_comm* c);
int32_t dvs_comm_rccl_version(void);  /* ncclGetVersion code, 0 if RCCL is unavailable */
/* bytes of one rank's boundary block {descriptors[cap x 32], int32 n, padding to 64 B} */
size_t dvs_boundary_block_bytes(int32_t cap);
/* The exchange step, called once per GLOBAL BATCH with this rank's LAST frame of the batch.  Packs {d_desc_last (cap rows of 32 B,
 * 16-byte aligned), *d_n_last} into this rank's slot of a gather buffer owned by the communicator (three buffers used in turn,
 * allocated once per capacity), all-gathers in place on `stream`, and returns device pointers to the predecessor of this rank's
 * FIRST frame of the same batch in the global frame order: rank r >= 1 gets rank r - 1's block of this call, rank 0 gets the last
 * rank's block of the PREVIOUS call (both NULL on the first call: the sequence starts there).  The pointers stay valid until the
 * call after next.  Asynchronous. */
dvs_status dvs_exchange_boundary(dvs_comm* c, void* stream, const uint8_t* d_desc_last, const int32_t* d_n_last, int32_t cap,
                                 const uint8_t** d_prev_desc, const int32_t** d_prev_n);
/* plain all-gather of bytes_per_rank bytes per rank (level-sharded extraction gathers its per-level blocks with it) */
dvs_status dvs_comm_all_gather(dvs_comm* c, void* stream, const void* d_send, void* d_recv, size_t bytes_per_rank);

/* ======================= the streaming step: extract batch i + match batch i - 1 ================= */
/* The reference's frame loop (frontend.cpp:1084-1123: gray -> (*orb_extractor_)(...) -> orb_matcher_->match(current, previous)) for a
 * host that holds its frames in device memory B at a time.  ONE call per step enqueues the whole software-pipelined schedule that
 * bench.py times and tests/test_gpu_pipeline.py checks frame by frame against the oracle:
 *   - the extraction of batch i on the extractor's streams (next batch's pyramid beside FAST, blur beside the quad-tree, the
 *     descriptor stage deferred beside the NEXT step's FAST);
 *   - the B match jobs of batch i - 1 (frame t against t - 1; frame 0 against the last frame of batch i - 2, or against the frame a
 *     communicator's boundary exchange returns) on the match stream, released behind batch i's FAST;
 *   - `nsets` (>= 3 when pipelined) output sets in rotation, the extraction of step i + nsets gated on the last reader of set i.
 * pipelined = 0: the plain schedule — every batch's match behind its own extraction on one stream (nsets >= 1).
 * Small batches (lanes >= 2; automatic for batch <= DVS_PIPELINE_LANE_BATCH): the machine is mostly idle within one step and the
 * step is the latency of its kernel chain, so `lanes` independent extractor / matcher pairs, one stream each, take the steps in turn
 * — step i runs serially on lane i % lanes (pyramid -> FAST -> quad-tree -> blur -> descriptors -> its own match, no internal
 * forks), up to `lanes` steps in flight, ordered only where data flows: the match of batch i waits for batch i - 1's descriptors
 * (another lane), the extraction of step i + nsets for the readers of set i.  Results are those of any other schedule, bit for bit.
 * The handle owns extractors, matchers, streams, events and the output sets; nothing is allocated per step.  Not thread-safe. */
typedef struct dvs_pipeline dvs_pipeline;
typedef struct dvs_pipeline_params {
  dvs_orb_params orb;   /* max_batch is ignored (= batch) */
  int32_t batch;        /* B frames per step, tight rows: frame f at d_imgs + f * rows * cols */
  int32_t rows, cols;
  int32_t nsets;        /* output sets in rotation; 0 = 4 (lane schedule: two per lane); lanes are reduced to a divisor of nsets */
  int32_t pipelined;    /* 1: the software pipeline described above; 0: serial match */
  int32_t lanes;        /* pipelined only.  0 = by batch size (DVS_PIPELINE_MAX_LANES up to DVS_PIPELINE_LANE_BATCH frames, else 1); 1 = the
                           two-stream software pipeline; 2..DVS_PIPELINE_MAX_LANES = lanes.  The fourth lane's stream has the highest dispatch
                           priority: streams of one priority share four hardware queues with the process's default stream */
  int32_t quadtree_async; /* two-stream pipeline only.  1: the four-stream form — quad-tree on the extractor's auxiliary stream beside the next
                           step's FAST (dvs_orb_set_async_quadtree), descriptor stage on the match stream (dvs_orb_set_tail_stream), blur on the
                           main stream ahead of FAST; -1: off; 0 = by batch size (on for DVS_PIPELINE_LANE_BATCH < batch <= DVS_PIPELINE_ASYNC_BATCH:
                           measured +21 % at 8 frames per step, +13 % at 16, +3.5 % at 24, a tie at 28..32, -4 % at 64: profiles/r04_batch_sweep.json) */
} dvs_pipeline_params;
#define DVS_PIPELINE_ASYNC_BATCH 24
#define DVS_PIPELINE_MAX_LANES 4     /* four hardware queues run at a time on this part: a fifth lane collapses all of them (EXPERIMENTS.md) */
#define DVS_PIPELINE_LANE_BATCH 6    /* lanes = 0: batches up to this size run on lanes (four lanes against the four-stream form: +16 % at 5 frames, +14 % at 6, a tie at 7, -5 % at 8) */
/* results of one step's batch (device pointers into the handle's output set; valid until step + nsets is enqueued) */
typedef struct dvs_pipeline_set {
  const dvs_keypoint* d_kps;  /* [B][capacity] */
  const uint8_t* d_desc;      /* [B][capacity][32] */
  const int32_t* d_n;         /* [B] */
  const int32_t* d_idx;       /* [B][capacity] trainIdx of frame f's keypoints in frame f - 1 */
  const int32_t* d_dist;      /* [B][capacity] */
  void* ev_extracted;         /* hipEvent_t: keypoints / descriptors / counts complete */
  void* ev_matched;           /* hipEvent_t: the batch's match jobs complete (recorded when they are enqueued: one step late if pipelined) */
  int32_t capacity;
} dvs_pipeline_set;
enum { DVS_PIPELINE_NO_MATCH = 1 };   /* step flags: extraction only (per-stage timing passes) */
dvs_status dvs_pipeline_create(const dvs_pipeline_params* params, int32_t device, dvs_pipeline** out);
void dvs_pipeline_destroy(dvs_pipeline* p);
/* frames sharded contiguously over ranks: frame 0 of a batch is matched against the frame dvs_exchange_boundary
 * returns (one all-gather per global batch on the match stream) instead of the previous batch's last frame.  NULL detaches. */
dvs_status dvs_pipeline_attach_comm(dvs_pipeline* p, dvs_comm* comm);
/* step i: d_imgs = this step's batch, d_next_imgs = the batch the NEXT step will pass (its pyramid is built ahead), NULL if unknown */
dvs_status dvs_pipeline_step(dvs_pipeline* p, const uint8_t* d_imgs, const uint8_t* d_next_imgs, int32_t flags);
/* the match of the last extracted batch (the pipelined schedule runs it one step late); no-op for the serial schedule */
dvs_status dvs_pipeline_flush(dvs_pipeline* p);
dvs_status dvs_pipeline_synchronize(dvs_pipeline* p);
/* synchronise and restart the sequence at step 0 (the next batch has no predecessor) */
dvs_status dvs_pipeline_reset(dvs_pipeline* p);
int64_t dvs_pipeline_steps(const dvs_pipeline* p);   /* steps enqueued since creation / reset */
dvs_status dvs_pipeline_get_set(const dvs_pipeline* p, int64_t step, dvs_pipeline_set* out);
/* which schedule the handle runs */
int32_t dvs_pipeline_quadtree_async(const dvs_pipeline* p);   /* 1: the four-stream form is in use */
int32_t dvs_pipeline_nsets(const dvs_pipeline* p);    /* output sets in rotation */
int32_t dvs_pipeline_lanes(const dvs_pipeline* p);    /* 0: serial schedule, 1: two-stream software pipeline, >= 2: lanes */
/* ---- measurement (bench.py's per-stage report and roofline; results are unaffected) ----
 * The extractor's scheduling hooks that this step is composed of (announced next batch, after-FAST event, deferred outputs, reuse guard,
 * quad-tree / tail streams, single-stream handles) are internal to the library since round 5; libdvslam_hip_test.so exports them for the
 * tests that pin them one by one (include/dvslam_hip_test.h). */
enum { DVS_STAGE_PYRAMID = 0, DVS_STAGE_FAST = 1, DVS_STAGE_OCTREE = 2, DVS_STAGE_BLUR = 3, DVS_STAGE_DESCRIBE = 4, DVS_STAGE_COUNT = 5 };
/* 1: every kernel of an extraction alone on the main stream (no overlap inside a step: what per-kernel durations are measured with);
 * 0: the shipped schedule.  Synchronises; refused for the lane schedule (its handles have one stream for good). */
dvs_status dvs_pipeline_set_serialized(dvs_pipeline* p, int32_t on);
/* per-stage GPU time (hipEvents around every stage of lane 0's extractor): switch, then accumulated milliseconds and launch-sequence
 * counts per stage since the last reset (synchronises) */
dvs_status dvs_pipeline_stage_timing(dvs_pipeline* p, int32_t on);
dvs_status dvs_pipeline_get_stage_times(dvs_pipeline* p, double* ms /* [DVS_STAGE_COUNT] */, int64_t* calls /* [DVS_STAGE_COUNT] */, int32_t reset);

/* ======================= glue either side of the path =============== */
/* A dvs_matcher handle is the context (stream + scratch).  Host pointers unless the name says _device. */

/* cv::cvtColor(bgr, gray, COLOR_BGR2GRAY) on 8UC3 (frontend.cpp:1084).  variant 0 = OpenCV 4.x 15-bit coefficients
 * (B*3735 + G*19235 + R*9798 + 16384) >> 15, variant 1 = the 14-bit ones of older releases. */
dvs_status dvs_bgr_to_gray(dvs_matcher* ctx, const uint8_t* bgr, int32_t rows, int32_t cols, size_t step, uint8_t* gray, size_t gray_step,
                           int32_t variant);
dvs_status dvs_bgr_to_gray_device(dvs_matcher* ctx, const uint8_t* d_bgr, int32_t nimg, int32_t rows, int32_t cols, size_t step,
                                  size_t frame_stride, uint8_t* d_gray, size_t gray_step, size_t gray_frame_stride, int32_t variant);
/* filterDepth / isValidDepth (frontend.cpp:457-527): keep keypoints (and descriptor rows) whose pixel (std::round of pt) has
 * depth_u16 * 0.001f in [min_depth, max_depth]; order preserved; out_index = original indices (may be NULL). */
dvs_status dvs_filter_depth(dvs_matcher* ctx, const dvs_keypoint* kps, const uint8_t* desc, int32_t n, const uint16_t* depth, int32_t rows,
                            int32_t cols, size_t step_bytes, float min_depth, float max_depth, dvs_keypoint* out_kps, uint8_t* out_desc,
                            int32_t* out_index, int32_t* n_out);
/* nframes frames resident in HBM: frame f uses rows [0, d_n[f]) of the [nframes][stride_rows] blocks dvs_orb_extract_batch_device
 * wrote and the depth image at d_depth + f * frame_stride_bytes.  Asynchronous. */
dvs_status dvs_filter_depth_batch_device(dvs_matcher* ctx, const dvs_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                         int32_t stride_rows, int32_t nframes, const uint16_t* d_depth, int32_t rows, int32_t cols,
                                         size_t step_bytes, size_t frame_stride_bytes, float min_depth, float max_depth,
                                         dvs_keypoint* d_out_kps, uint8_t* d_out_desc, int32_t* d_out_index, int32_t* d_n_out);
/* frontend.cpp:1126-1132: matches with (float)distance < max_distance as (queryIdx, trainIdx, distance) int32 triplets */
dvs_status dvs_filter_matches(dvs_matcher* ctx, const int32_t* train_idx, const int32_t* dist, int32_t n, float max_distance,
                              int32_t* out_triplets, int32_t* n_out);
/* publishKeyframe (frontend.cpp:732-776): float back-projection with the depth image, keep 0.3 < Z < 3.0, world = R * p + t
 * (R row-major 3x3, double).  world_xyz[3 * n_out], out_index = keypoint indices (= the message's landmark_id).
 * A keypoint whose rounded position lies outside the depth image is dropped — here, in dvs_publish_keyframe* and in
 * dvs_filter_depth* alike (the reference reads the depth image unchecked at frontend.cpp:737). */
dvs_status dvs_backproject(dvs_matcher* ctx, const dvs_keypoint* kps, int32_t n, const uint16_t* depth, int32_t rows, int32_t cols,
                           size_t step_bytes, float fx, float fy, float cx, float cy, const double* R, const double* t, double* world_xyz,
                           int32_t* out_index, int32_t* n_out);
/* ---- Keyframe.msg on the wire (dynamic_visual_slam_interfaces/msg/{Keyframe,Landmark,Observation}.msg) ------------------------
 * The frontend publishes one Keyframe per keyframe on /frontend/keyframe (frontend.cpp:200, 699-790) and the backend consumes it;
 * rmw serialises it as little-endian CDR.  dvs_publish_keyframe* run publishKeyframe's loop (depth gate, back-projection,
 * landmark_id = keypoint index, float64 pixels, 32-byte descriptor) and write that CDR payload directly, so an adapter hands
 * the bytes to rclcpp::SerializedMessage / publish_serialized_message without touching 2000 keypoints on the host. */
typedef struct dvs_keyframe_header {
  int32_t stamp_sec;           /* header.stamp */
  uint32_t stamp_nanosec;
  const char* frame_id;        /* header.frame_id ("camera_link", frontend.cpp:727); <= 63 characters */
  uint64_t keyframe_id;        /* Keyframe.frame_id */
  double translation[3];       /* pose.translation x y z  (t_, optical frame) */
  double rotation_xyzw[4];     /* pose.rotation x y z w   (Eigen::Quaterniond(R_).normalized()) */
} dvs_keyframe_header;
/* payload bytes if all n keypoints pass the depth gate (the size of the buffer to provide) */
size_t dvs_keyframe_cdr_capacity(const char* header_frame_id, int32_t n);
/* device-resident inputs (extractor outputs, 16UC1 depth), payload into d_out; *d_out_size = bytes needed (> cap: nothing
 * written), *d_n_out = landmarks.  R (row-major 3x3), t: host.  Asynchronous on the context's stream. */
dvs_status dvs_publish_keyframe_device(dvs_matcher* ctx, const dvs_keyframe_header* hdr, const dvs_keypoint* d_kps, const uint8_t* d_desc,
                                       int32_t n, const uint16_t* d_depth, int32_t rows, int32_t cols, size_t step_bytes, float fx, float fy,
                                       float cx, float cy, const double* R, const double* t, uint8_t* d_out, size_t cap,
                                       uint64_t* d_out_size, int32_t* d_n_out);
/* host pointers in, payload in `out` */
dvs_status dvs_publish_keyframe(dvs_matcher* ctx, const dvs_keyframe_header* hdr, const dvs_keypoint* kps, const uint8_t* desc, int32_t n,
                                const uint16_t* depth, int32_t rows, int32_t cols, size_t step_bytes, float fx, float fy, float cx, float cy,
                                const double* R, const double* t, uint8_t* out, size_t cap, size_t* out_size, int32_t* n_landmarks);
/* the subscriber's side: a received payload as flat arrays (any output pointer may be NULL); hdr->frame_id points into
 * frame_id_buf.  Host code, no GPU.  DVS_ERR_CAPACITY (with the counts set) when an array holds fewer than cap_n entries. */
dvs_status dvs_keyframe_unpack_cdr(const uint8_t* buf, size_t len, dvs_keyframe_header* hdr, char* frame_id_buf, size_t frame_id_cap,
                                   uint64_t* landmark_ids, double* landmark_xyz, uint64_t* obs_landmark_ids, double* obs_pixels,
                                   uint8_t* obs_desc, int32_t cap_n, int32_t* n_landmarks, int32_t* n_observations);
/* ---- the frontend's robust-estimation stages, as batched-hypothesis kernels ---------------------------
 * Two forms.  dvs_find_fundamental_ransac / dvs_solve_pnp_ransac: the library's own minimal solvers (8-point, P3P) over a documented
 * deterministic sampler — NOT OpenCV's sample sequence or kernels (7-point, EPnP); they implement the same estimator (threshold, confidence, iteration cap, the
 * adaptive stopping rule RANSACUpdateNumIters, error measures) over a documented deterministic sampler (`seed`; csrc/ransac.hip),
 * and parity is stated as a tolerance on the inlier set and the pose.  Host pointers.
 *
 * cv::findFundamentalMat(pts1, pts2, mask, cv::FM_RANSAC, threshold = 2.0, confidence = 0.99) as frontend.cpp:635, 1146-1147
 * call it: pts n x 2 float, x2^T F x1 = 0; inlier_mask[n] = 1 where max of the two squared epipolar distances <= threshold^2
 * for the best model.  F9 (row-major, unit Frobenius norm) may be NULL.  n < 8: mask of zeros, *n_inliers = 0. */
dvs_status dvs_find_fundamental_ransac(dvs_matcher* ctx, const float* pts1, const float* pts2, int32_t n, double threshold, double confidence,
                                       int32_t max_iters /* OpenCV default 1000 */, uint64_t seed, double* F9, uint8_t* inlier_mask,
                                       int32_t* n_inliers);
/* nprob independent problems in ONE launch sequence (the replay's per-frame gates are pose-independent: tools/replay_tracking.py
 * runs them for all frames at once): problem b = correspondences [offsets[b], offsets[b + 1]) of the concatenated arrays, sampler
 * seed seeds[b]; outputs concatenated / indexed the same way.  Every problem gets exactly what the single call gives it. */
dvs_status dvs_find_fundamental_ransac_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets /* nprob + 1, offsets[0] = 0 */, const float* pts1,
                                             const float* pts2, double threshold, double confidence, int32_t max_iters, const uint64_t* seeds,
                                             double* F9 /* nprob x 9 or NULL */, uint8_t* inlier_mask /* offsets[nprob] */, int32_t* n_inliers /* nprob or NULL */);
/* cv::findFundamentalMat(pts1, pts2, mask, cv::FM_RANSAC, threshold, confidence) (frontend.cpp:635, 1146-1147) the way OpenCV 4.x
 * itself runs it, restated from the published algorithm (calib3d fundam.cpp / ptsetreg.cpp; csrc/ransac.hip k_f7_hypotheses,
 * k_lmeds_select): the sample sequence — ONE cv::RNG seeded with (uint64)-1, index = next() % n, drawn again while it repeats, whole
 * samples drawn again while their last point is collinear with two earlier ones — and the 7-point solver (two null vectors,
 * cv::solveCubic, one model per real root, F(3,3) = 1), errors compared as floats.  From 15 correspondences on RANSAC
 * (RANSACPointSetRegistrator: strictly better inlier count wins, adaptive stopping rule, no refit; *iterations = loop iterations run);
 * from 8 to 14 LMedS, as OpenCV switches (LMeDSPointSetRegistrator: a fixed 300 iterations at confidence 0.99, smallest MEDIAN error
 * wins, inliers within sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median) >= 0.001; fewer than 7 of them: F9 all zero — OpenCV
 * returns an empty matrix — with the mask still written).  F9 row-major with F[8] = 1 (0 where OpenCV sets it so), may be NULL.
 * OpenCV's maxIters default is 1000.  n < 8 (the reference never calls there, frontend.cpp:627): DVS_ERR_UNSUPPORTED.  PARITY UNPINNED
 * like everything else (no OpenCV in this image).  What cannot agree even in principle: a tie between two models of ONE sample,
 * whose order follows the null-space basis (OpenCV: SVD); and LMedS below 14 points, where every model's median is the error of one
 * of its own sample points — rounding noise — so that WHICH of the 300 samples wins (and becomes the 7-point inlier set) is decided
 * by the rounding of the solver at hand. */
dvs_status dvs_find_fundamental_cv(dvs_matcher* ctx, const float* pts1, const float* pts2, int32_t n, double threshold, double confidence,
                                   int32_t max_iters, double* F9, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iterations);
dvs_status dvs_find_fundamental_cv_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts1, const float* pts2, double threshold,
                                         double confidence, int32_t max_iters, double* F9, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iterations);
/* host only (works without a GPU): the sample sequence of the call above — iteration i drew idx[model_points i ..]; *found =
 * iterations that have a sample (cv::RNG((uint64)-1), uniform(0, n) = next() % n, repeats and collinear samples drawn again).
 * pts1 = pts2 = NULL: a callback WITHOUT checkSubset — the 5-point samples of dvs_solve_pnp_ransac_cv (model_points = 5). */
dvs_status dvs_cv_ransac_subsets(const float* pts1, const float* pts2, int32_t n, int32_t model_points, int32_t iterations, int32_t* idx, int32_t* found);
/* cv::solvePnPRansac(obj, img, K, noArray, rvec, tvec, false, iterations = 100, reproj_err = 4.0, confidence = 0.99, inliers)
 * (frontend.cpp:911-921; zero distortion): obj n x 3 float (camera frame of the previous image), img n x 2 float,
 * K4 = {fx, fy, cx, cy}.  P3P hypotheses, best by inlier count, Levenberg-Marquardt refinement on the inliers (the
 * SOLVEPNP_ITERATIVE step).  rvec3 = Rodrigues vector, tvec3: x_cam = R X + t.  inliers: ascending indices (capacity n, may be
 * NULL).  *success = 0 (and zeros) when no model found or n < 4. */
dvs_status dvs_solve_pnp_ransac(dvs_matcher* ctx, const float* pts3d, const float* pts2d, int32_t n, const double* K4, int32_t iterations,
                                double reproj_err, double confidence, uint64_t seed, double* rvec3, double* tvec3, int32_t* inliers,
                                int32_t* n_inliers, int32_t* success);
/* batch form: inliers concatenated like the points (problem b's ascending indices at inliers + offsets[b]); rvec3 / tvec3 nprob x 3 */
dvs_status dvs_solve_pnp_ransac_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts3d, const float* pts2d, const double* K4,
                                      int32_t iterations, double reproj_err, double confidence, const uint64_t* seeds, double* rvec3, double* tvec3,
                                      int32_t* inliers /* offsets[nprob] or NULL */, int32_t* n_inliers /* nprob or NULL */, int32_t* success /* nprob */);

/* cv::solvePnPRansac(points3d, points2d, K, no distortion, rvec, tvec, false, iterations, reproj_err, confidence, inliers) AS OPENCV 4.x
 * RUNS IT with its default flags for the reference's call (frontend.cpp:911-921: 100, 4.0, 0.99) — csrc/pnp_cv.h: the 5-point samples of
 * ONE cv::RNG((uint64)-1), the EPnP minimal solver, projectPoints + squared error + threshold in float, the adaptive iteration count
 * (RANSACUpdateNumIters, 5 model points), then solvePnP(SOLVEPNP_ITERATIVE) on the inliers of the best model (planar / DLT initialisation
 * + CvLevMarq, 20 iterations, FLT_EPSILON).  inliers: the best model's inlier indices in index order (capacity n), as OpenCV returns them.
 * success = 0 with a model: the refit could not be initialised, rvec / tvec are the RANSAC stage's (OpenCV returns false there too).
 * Fewer than 6 points: success 0, nothing else written (the reference returns before the call, frontend.cpp:900).  PARITY UNPINNED:
 * restated from the published sources; every SVD is a Jacobi eigen-decomposition here (same models to rounding).  On an EXACTLY planar
 * point set EPnP's fourth control point coincides with the centroid and M^T M gains a trivial three-dimensional null space that its
 * N <= 3 approximations cannot leave — in OpenCV as here; dvs_solve_pnp_ransac (P3P) has no such case. */
dvs_status dvs_solve_pnp_ransac_cv(dvs_matcher* ctx, const float* pts3d, const float* pts2d, int32_t n, const double* K4, int32_t iterations,
                                   double reproj_err, double confidence, double* rvec3, double* tvec3, int32_t* inliers, int32_t* n_inliers,
                                   int32_t* success, int32_t* iterations_run);
dvs_status dvs_solve_pnp_ransac_cv_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts3d, const float* pts2d, const double* K4,
                                         int32_t iterations, double reproj_err, double confidence, double* rvec3, double* tvec3, int32_t* inliers,
                                         int32_t* n_inliers, int32_t* success, int32_t* iterations_run);

/* Harris corner measure as cv::ORB scores keypoints (ORB::HARRIS_SCORE, the mode test_dbow2_integration.cpp:19 runs with:
 * OpenCV features2d orb.cpp HarrisResponses — integer 3x3 gradients over a block_size^2 window, response = (ab - c^2 - k(a+b)^2)
 * / (4 block_size 255)^4 in float; cv::ORB uses block_size 7, k 0.04).  x, y: integer pixel positions in this image (one pyramid
 * layer).  Points closer than block_size/2 + 1 to the border get 0 (OpenCV reads outside the layer there). block_size <= 8. */
dvs_status dvs_harris_responses(dvs_matcher* ctx, const uint8_t* img, int32_t rows, int32_t cols, size_t step, const int32_t* x,
                                const int32_t* y, int32_t n, int32_t block_size, float k, float* response);
dvs_status dvs_harris_responses_device(dvs_matcher* ctx, const uint8_t* d_img, int32_t rows, int32_t cols, size_t step, const int32_t* d_x,
                                       const int32_t* d_y, int32_t n, int32_t block_size, float k, float* d_response);
/* associateObservation + reprojectPoint (backend.cpp:1064-1173) for all observations of one category against a snapshot of
 * that category's landmarks (arrays in the database's iteration order): best[i] = index of the candidate with Hamming
 * distance < max_descriptor_distance and the smallest reprojection error < max_reprojection_distance (first on ties), or -1.
 * obs_px: nobs x (u, v) float; lm_xyz: nlm x 3 float (cv::Point3f); R, t: keyframe pose as extractPoseFromTransform gives it. */
dvs_status dvs_associate(dvs_matcher* ctx, const uint8_t* obs_desc, const float* obs_px, int32_t nobs, const uint8_t* lm_desc,
                         const float* lm_xyz, int32_t nlm, const double* R, const double* t, double fx, double fy, double cx, double cy,
                         double max_descriptor_distance, double max_reprojection_distance, int32_t* best);

/* The same, plus every observation's candidate list: cand_lm[cand_offsets[i] .. cand_offsets[i + 1]) = the landmarks with Hamming
 * distance < max_descriptor_distance for observation i, in landmark order (cand_offsets: nobs + 1 entries; *n_cand = total; if
 * it exceeds cand_cap nothing is written to cand_lm and DVS_ERR_CAPACITY is returned with *n_cand set).  The reference applies
 * associations one by one and re-triangulates the landmark after each (backend.cpp:758-777): include/dvslam/association.hpp
 * uses the lists to re-evaluate exactly the observations whose candidates moved, in observation order. */
dvs_status dvs_associate_candidates(dvs_matcher* ctx, const uint8_t* obs_desc, const float* obs_px, int32_t nobs, const uint8_t* lm_desc,
                                    const float* lm_xyz, int32_t nlm, const double* R, const double* t, double fx, double fy, double cx,
                                    double cy, double max_descriptor_distance, double max_reprojection_distance, int32_t* best,
                                    int64_t* cand_offsets, int32_t* cand_lm, int64_t cand_cap, int64_t* n_cand);

/* ======================================= B3: bundle adjustment ================================= */

typedef struct dvs_ba dvs_ba;

typedef struct dvs_ba_summary {
  int32_t termination;          /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE (ceres::TerminationType order) */
  int32_t num_successful_steps; /* OptimizationResult::iterations_completed (bundle_adjustment.hpp:862) */
  int32_t num_iterations;
  int32_t linear_solver;        /* who solved the normal equations: 1 = the device (dvs_ba_solve_device), 2 = the host (dvs_ba_solve) */
  double initial_cost, final_cost;
} dvs_ba_summary;

dvs_status dvs_ba_create(int32_t device, dvs_ba** out);
void dvs_ba_destroy(dvs_ba* h);
dvs_status dvs_ba_set_stream(dvs_ba* h, void* hip_stream);
dvs_status dvs_ba_synchronize(dvs_ba* h);

/* Problem in the optimiser's parameterisation (bundle_adjustment.hpp:92-165): K poses = world->camera quaternion
 * (w,x,y,z) + translation, L landmarks, R observations (cam_idx, lm_idx, uv).  pose_fixed / lm_fixed: 1 = constant block
 * (bundle_adjustment.hpp:781-785, 795-797).  Intrinsics and sigma exactly as passed (no sanity checks: backend.cpp:180). */
dvs_status dvs_ba_set_problem(dvs_ba* h, int32_t K, const double* q_wxyz, const double* t, int32_t L, const double* X,
                              int32_t R, const int32_t* cam_idx, const int32_t* lm_idx, const double* uv,
                              const uint8_t* pose_fixed, const uint8_t* lm_fixed,
                              double fx, double fy, double cx, double cy, double sigma_pixels, double huber_delta);
/* one evaluation at the current parameters: robustified cost 0.5*sum(rho), and (each nullable) loss-corrected residuals
 * R x 2, local pose Jacobians R x 2 x 6 (rotation tangent first, then translation), landmark Jacobians R x 2 x 3,
 * gradient (6K + 3L, fixed blocks zero).  Host output pointers. */
dvs_status dvs_ba_evaluate(dvs_ba* h, double* cost, double* residuals, double* J_pose, double* J_lm, double* grad);
/* raw (un-robustified) functor outputs as ceres::CostFunction::Evaluate delivers them: residuals R x 2, jacobians
 * wrt q (R x 2 x 4), t (R x 2 x 3), X (R x 2 x 3), row-major; any may be NULL. */
dvs_status dvs_ba_evaluate_raw(dvs_ba* h, double* residuals, double* J_q, double* J_t, double* J_X);
/* Gauss-Newton blocks: H_pp K x 6 x 6, H_ll L x 3 x 3, W R x 6 x 3 (pose-landmark block per observation), g (6K + 3L) */
dvs_status dvs_ba_normal_equations(dvs_ba* h, double* H_pp, double* H_ll, double* W, double* g, double* cost);
/* `iters` back-to-back evaluations (residuals + Jacobians + loss + reductions) with nothing copied to the host; for
 * throughput measurement.  Asynchronous. */
dvs_status dvs_ba_evaluate_device(dvs_ba* h, int32_t iters);
/* Levenberg-Marquardt + Schur complement with the trust-region schedule of ceres::Solve as configured at
 * bundle_adjustment.hpp:839-847.  Parameters are updated in place; read them back with dvs_ba_get_parameters. */
dvs_status dvs_ba_solve(dvs_ba* h, int32_t max_iterations, double function_tolerance, double gradient_tolerance,
                        double parameter_tolerance, dvs_ba_summary* summary);
/* The same solve with the linear algebra on the device (Jacobi scaling, LM diagonal, Schur complement, Cholesky of the
 * reduced camera system, back-substitution, model cost change, candidate point): one 64-byte status record crosses PCIe per
 * trial step instead of the W blocks.  Same trust-region decisions, fixed-order reductions; sums are associated differently
 * from dvs_ba_solve, so costs agree to rounding (tests: 1e-9 relative), not bit for bit.  Sliding-window shapes only:
 * <= 64 cameras, 1..16 of them free, a landmark observed at most once per camera; DVS_ERR_UNSUPPORTED otherwise. */
dvs_status dvs_ba_solve_device(dvs_ba* h, int32_t max_iterations, double function_tolerance, double gradient_tolerance,
                               double parameter_tolerance, dvs_ba_summary* summary);
/* Trust-region log of the last dvs_ba_solve / dvs_ba_solve_device (what ceres::Solver::Summary::iterations holds): one row of
 * 6 doubles per iteration = {radius the step was computed with, kind, cost change, model cost change, relative decrease,
 * candidate cost}; kind 0 = invalid step, 1 = accepted, 2 = rejected, 3 / 4 = parameter / function tolerance reached.
 * *n_rows = rows available; at most cap_rows are written (rows may be NULL). */
dvs_status dvs_ba_get_trace(const dvs_ba* h, double* rows, int32_t cap_rows, int32_t* n_rows);
dvs_status dvs_ba_get_parameters(dvs_ba* h, double* q_wxyz, double* t, double* X);
/* CameraPose::fromRt / toRt (bundle_adjustment.hpp:138-165, 192-212): caller-convention (R row-major 3x3, t) <->
 * optimiser (q_wxyz, translation).  Host arithmetic used by the SlidingWindowBA adapter. */
dvs_status dvs_ba_pose_from_rt(const double* R, const double* t, double* q_wxyz, double* trans);
dvs_status dvs_ba_pose_to_rt(const double* q_wxyz, const double* trans, double* R, double* t);

/* ============================ cv::ORB-compatible extractor ============================
 * Replaces cv::ORB::create(...)->detectAndCompute(image, noArray(), keypoints, descriptors)
 * (/root/reference/dynamic_visual_slam/test/test_dbow2_integration.cpp:19,38; OpenCV 4.x features2d/src/orb.cpp): pyramid from
 * level 0 with INTER_LINEAR_EXACT, FAST-9/16 with non-max suppression over whole levels, runByImageBorder(edge), retainBest
 * (std::nth_element + std::partition semantics: ties with the last kept response are all kept, the order is libstdc++'s), HARRIS
 * responses (7 x 7, k = 0.04), intensity-centroid angle, rBRIEF on the 7 x 7 Gaussian-blurred level.  Keypoints come back in
 * cv::KeyPoint layout (pt in level-0 coordinates, size = 31 * scale, response = HARRIS value or FAST score, octave = level).
 * Built: firstLevel = 0, WTA_K = 2, patchSize = 31 (cv::ORB's defaults), edge_threshold >= 19; anything else DVS_ERR_UNSUPPORTED.
 * Because retainBest keeps ties, the count is data dependent: DVS_ERR_CAPACITY (with *n_out = rows needed) when capacity is
 * too small; an empty image returns DVS_OK with *n_out = 0 (detectAndCompute returns without detecting). */
typedef struct dvs_cvorb dvs_cvorb;
typedef struct dvs_cvorb_params {
  int32_t nfeatures;       /* 500 */
  float scale_factor;      /* 1.2f */
  int32_t nlevels;         /* 8 */
  int32_t edge_threshold;  /* 31 */
  int32_t first_level;     /* 0 */
  int32_t wta_k;           /* 2 */
  int32_t score_type;      /* 0 = cv::ORB::HARRIS_SCORE, 1 = FAST_SCORE */
  int32_t patch_size;      /* 31 */
  int32_t fast_threshold;  /* 20 */
} dvs_cvorb_params;
dvs_status dvs_cvorb_create(const dvs_cvorb_params* params, int32_t device, dvs_cvorb** out);
void dvs_cvorb_destroy(dvs_cvorb* h);
dvs_status dvs_cvorb_detect_and_compute(dvs_cvorb* h, const uint8_t* gray, int32_t rows, int32_t cols, size_t step_bytes, dvs_keypoint* kps,
                                        uint8_t* desc /* capacity x 32 */, int32_t capacity, int32_t* n_out);
/* parity introspection: level `level` of the last call's pyramid (blurred = 1: after the Gaussian), tight rows */
dvs_status dvs_cvorb_get_level(dvs_cvorb* h, int32_t level, int32_t blurred, uint8_t* dst, int32_t cap_bytes, int32_t* w, int32_t* hgt);

#ifdef __cplusplus
}
#endif
#endif /* DVSLAM_HIP_H */
