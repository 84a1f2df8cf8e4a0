// A C++ host of the benchmarked streaming step (include/dvslam/streaming_pipeline.hpp over dvs_pipeline_*): what a maintainer of the
// reference's frontend (frontend.cpp:1084-1123) writes once frames arrive B at a time in device memory.  No Python, no torch, no HIP
// toolchain: plain g++ against the C-ABI.
//
//   pipeline_stream <frames.bin> <B> <rows> <cols> <nfeatures> <NB> <steps> <nsets> <out.bin> [world] [lanes]
//
// frames.bin: NB global batches of world * B gray frames (tight rows).  world == 1: one pipeline.  world > 1: `world` logical ranks
// of this process on one device (dvs_comm_create_loopback), one host thread + pipeline + communicator each, frames sharded
// contiguously, the boundary frame through dvs_exchange_boundary (attach).  Runs `steps` pipelined steps rotating over the NB
// batches, flush, synchronize, and writes for every rank and each of the last min(nsets, steps) steps:
//   int32 n[B]; dvs_keypoint kps[B][cap]; uint8 desc[B][cap][32]; int32 idx[B][cap]; int32 dist[B][cap]
// (rank-major, then step-major) behind a header {int32 world, B, cap, first_step, nsteps_written}.  tests/test_cpp_pipeline.py
// compares the bytes with the Python caller of the same C-ABI and with the oracle.  Prints the host time spent in step() per step.
// Exit code 3 = no GPU (the build container).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "dvslam/streaming_pipeline.hpp"

static std::vector<uint8_t> read_file(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)sz);
  if (std::fread(b.data(), 1, b.size(), f) != b.size()) { std::perror("fread"); std::exit(2); }
  std::fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (dvs_device_count() < 1) { std::printf("no device: streaming_pipeline.hpp compiled, nothing run\n"); return 3; }
  if (argc < 10) { std::fprintf(stderr, "usage: %s frames.bin B rows cols nfeatures NB steps nsets out.bin [world]\n", argv[0]); return 2; }
  const int B = std::atoi(argv[2]), rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), nf = std::atoi(argv[5]), NB = std::atoi(argv[6]);
  const int steps = std::atoi(argv[7]), world = argc > 10 ? std::atoi(argv[10]) : 1;
  int nsets = std::atoi(argv[8]);   // 0 = the schedule's default
  const int lanes = argc > 11 ? std::atoi(argv[11]) : 0;   // 0 = by batch size, 1 = two-stream software pipeline, 2..4 = lane schedule
  const std::vector<uint8_t> frames = read_file(argv[1]);
  const size_t frameBytes = (size_t)rows * cols, shard = frameBytes * B, global = shard * world;
  if (frames.size() != global * NB) { std::fprintf(stderr, "frames.bin: %zu bytes, expected %zu\n", frames.size(), global * NB); return 2; }

  try {
    // device-resident input: rank r's shard of every global batch
    std::vector<std::vector<uint8_t*>> d_img(world, std::vector<uint8_t*>(NB, nullptr));
    for (int r = 0; r < world; r++)
      for (int g = 0; g < NB; g++) {
        void* p = nullptr;
        if (dvs_malloc(0, shard, &p) != DVS_OK || dvs_memcpy_h2d(0, p, frames.data() + global * g + shard * r, shard) != DVS_OK) throw std::runtime_error(dvs_last_error());
        d_img[r][g] = (uint8_t*)p;
      }
    // handles first, communicators after (streams are hardware queues: created back to back)
    std::vector<std::unique_ptr<dvslam::StreamingPipeline>> pipes;
    for (int r = 0; r < world; r++) pipes.emplace_back(new dvslam::StreamingPipeline(B, rows, cols, nf, 1.2f, 8, 20, 7, 0, nsets, true, lanes));
    nsets = pipes[0]->nsets();
    std::vector<dvs_comm*> comms(world, nullptr);
    if (world > 1) {
      if (dvs_comm_create_loopback(0, world, comms.data()) != DVS_OK) throw std::runtime_error(dvs_last_error());
      for (int r = 0; r < world; r++) {
        if (dvs_comm_rank(comms[r]) != r || dvs_comm_world(comms[r]) != world) throw std::runtime_error("loopback rank / world");
        pipes[r]->attach(comms[r]);
      }
    }
    std::vector<double> host_ms(world, 0.0);
    auto w0 = std::chrono::steady_clock::now();
    std::vector<std::string> errors(world);
    auto run = [&](int r) {
      try {
        dvslam::StreamingPipeline& pipe = *pipes[r];
        if (world == 1) {   // warm-up outside the timing: the first step of every extractor allocates its workspace
          for (int i = 0; i < 2 * nsets; i++) pipe.step(d_img[r][i % NB], d_img[r][(i + 1) % NB]);
          pipe.flush();
          pipe.reset();     // synchronises; the sequence restarts at step 0
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; i++) pipe.step(d_img[r][i % NB], d_img[r][(i + 1) % NB]);
        host_ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
        if (world == 1) w0 = t0;
        pipe.flush();
        pipe.synchronize();
      } catch (const std::exception& e) { errors[r] = e.what(); }
    };
    w0 = std::chrono::steady_clock::now();
    if (world == 1) run(0);
    else {
      std::vector<std::thread> th;
      for (int r = 0; r < world; r++) th.emplace_back(run, r);
      for (auto& t : th) t.join();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    for (int r = 0; r < world; r++) if (!errors[r].empty()) { std::fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str()); return 1; }

    const int nw = steps < nsets ? steps : nsets, first = steps - nw, cap = pipes[0]->capacity();
    FILE* f = std::fopen(argv[9], "wb");
    if (!f) { std::perror(argv[9]); return 2; }
    const int32_t hdr[5] = {world, B, cap, first, nw};
    std::fwrite(hdr, 4, 5, f);
    std::vector<int32_t> n, idx, dist; std::vector<dvs_keypoint> kps; std::vector<uint8_t> desc;
    for (int r = 0; r < world; r++)
      for (int i = first; i < steps; i++) {
        pipes[r]->download(i, n, kps, desc, &idx, &dist);
        std::fwrite(n.data(), 4, n.size(), f);
        std::fwrite(kps.data(), sizeof(dvs_keypoint), kps.size(), f);
        std::fwrite(desc.data(), 1, desc.size(), f);
        std::fwrite(idx.data(), 4, idx.size(), f);
        std::fwrite(dist.data(), 4, dist.size(), f);
      }
    std::fclose(f);
    std::printf("pipeline_stream ok: world %d, lanes %d, %d steps of %d frames %dx%d, host %.4f ms per step() (rank 0), %.3f ms wall per step incl. drain\n", world, pipes[0]->lanes(), steps,
                B, cols, rows, host_ms[0], wall_ms / steps);
    pipes.clear();
    for (dvs_comm* c : comms) if (c) dvs_comm_destroy(c);
    for (auto& v : d_img) for (uint8_t* p : v) dvs_free(0, p);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "pipeline_stream: %s\n", e.what());
    return 1;
  }
  return 0;
}
