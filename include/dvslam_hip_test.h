/*
 * dvslam_hip_test.h — test hooks of libdvslam_hip_test.so (the product library built with -DDVS_TEST_HOOKS; `make test-lib`).
 * NOT part of the product ABI: lib/libdvslam_hip.so does not export any of these.  They expose internals that the parity tests pin
 * one by one — the libstdc++ std::sort / nth_element / partition replicas, the glibc sinf / cosf restatement, the geometry tables,
 * the PnP stage's host-compiled minimal solvers — and one scheduling aid (a kernel that holds a stream for a bounded time).
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

#ifdef __cplusplus
}
#endif
#endif /* DVSLAM_HIP_TEST_H */
