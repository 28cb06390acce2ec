// ilcc_corners_mgpu -- the multi-GPU form of the path in C++ (north_star: "host code stays C++ ... independent frames
// shard embarrassingly across the GPUs of one node with a single RCCL gather of per-frame corners over xGMI").
// One process, one host thread per GPU, one ilcc_handle per thread.  Frames (bag x click pairs are independent:
// /root/reference/ilcc2/test/get_lidar_corners.cpp:130-211 loops over them and overwrites every member per frame) are
// split into contiguous shards of ceil(F / G); every rank runs the whole path on its shard (H2D copy on the batch's own
// stream, records packed on the GPU by K9) and ONE ncclGather (rccl.h) ships the fixed-size records from each GPU's HBM
// to rank 0's, which checks tag + content word of every record and writes the process_data files.
//   ilcc_corners_mgpu [--accept-ambiguous] [--accept-low-coverage] <yaml|-> <out_prefix> <n_gpus, 0 = all> {<cloud.bin> <cx> <cy> <cz>}...
// A corner file is written for a frame under the library's accept rule (include/ilcc_hip.h, "accepting a frame": status ==
// ILCC_OK and ILCC_FLAG_LOW_COVERAGE clear -- the operator's `r` key of the reference, LidarCornersEst.cpp:415-441), with the same
// two switches as ilcc_corners (ilcc_corners_cli.cpp) and the class mirror: --accept-ambiguous also takes ILCC_AMBIGUOUS
// records, --accept-low-coverage takes flagged ones.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "ilcc_hip.h"

namespace {

struct Frame {
  std::vector<float> xyzi;
  float click[3];
};

bool load(const std::string& path, std::vector<float>* out) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in.is_open()) return false;
  const std::streamsize bytes = in.tellg();
  in.seekg(0);
  out->resize((size_t)bytes / 16 * 4);
  in.read(reinterpret_cast<char*>(out->data()), (std::streamsize)(out->size() * 4));
  return true;
}

// the check word K9 stores in header slot 17 (sharding.record_check is the same fold)
uint32_t record_check(const float* rec, uint32_t n_corners, uint32_t tag) {
  uint32_t x = 0;
  for (uint32_t c = 0; c < 3 * n_corners; ++c) {
    uint32_t b;
    std::memcpy(&b, rec + ILCC_RECORD_HEADER + c, 4);
    x ^= b * (2u * c + 1u);
  }
  uint32_t t = x ^ ((tag & 0xFFFFFFu) * 0x9E3779B1u);
  return (t ^ (t >> 24)) & 0xFFFFFFu;
}

struct RankJob {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  const std::vector<Frame>* frames = nullptr;
  ilcc_params params{};
  uint32_t per = 0, n_corners = 0, width = 0;
  std::vector<float> gathered;   // rank 0: world * per * width
  int status = 0;
};

// One rank.  A rank NEVER returns before the collective: a failure of its local work (ilcc_create, a bad frame, an
// out-of-memory) is remembered, its records keep the status -1 padding, and it still calls ncclGather -- otherwise the
// other ranks' threads would wait in the collective for ever and main() would hang in join().  Only when a rank cannot
// even set up its device buffers does it abort its communicator (which fails the peers' collective instead of hanging
// it).  Every resource is released on every path.
int run_rank(RankJob* j) {
  int failed = 0;
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    std::fprintf(stderr, "rank %d: %s: %s\n", j->rank, what, hipGetErrorString(e));
    failed = 1;
    return false;
  };
  const uint32_t F = (uint32_t)j->frames->size();
  const uint32_t lo = std::min(F, (uint32_t)j->rank * j->per), hi = std::min(F, lo + j->per);
  const uint32_t n_local = hi - lo;
  const size_t rec_floats = (size_t)j->per * j->width;
  hipStream_t stream = nullptr;
  float *d_rec = nullptr, *d_all = nullptr, *h_xyzi = nullptr, *h_click = nullptr;
  ilcc_handle* h = nullptr;

  // ---- buffers the collective needs
  bool ready = hip_ok(hipSetDevice(j->device), "hipSetDevice") &&
               hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate") &&
               hip_ok(hipMalloc((void**)&d_rec, rec_floats * 4), "hipMalloc(records)") &&
               (j->rank != 0 || hip_ok(hipMalloc((void**)&d_all, rec_floats * 4 * (size_t)j->world), "hipMalloc(gathered)"));
  if (ready) {
    // shards are padded to `per` records; a padding record carries status -1
    std::vector<float> pad(rec_floats, 0.f);
    for (uint32_t f = 0; f < j->per; ++f) pad[(size_t)f * j->width] = -1.f;
    ready = hip_ok(hipMemcpy(d_rec, pad.data(), rec_floats * 4, hipMemcpyHostToDevice), "hipMemcpy(padding)");
  }

  // test hook of THIS driver (never read by the library): make one rank's local work fail, to show that the process then
  // exits with an error instead of hanging in the collective
  const char* fail_rank = std::getenv("ILCC_MGPU_FAIL_RANK");
  if (fail_rank && std::atoi(fail_rank) == j->rank) {
    std::fprintf(stderr, "rank %d: local work failed (ILCC_MGPU_FAIL_RANK)\n", j->rank);
    failed = 1;
  }

  // ---- this rank's shard: failures are recorded, never returned from
  if (ready && !failed && n_local > 0) {
    std::vector<uint64_t> off(n_local + 1, 0);
    for (uint32_t f = 0; f < n_local; ++f) off[f + 1] = off[f] + (*j->frames)[lo + f].xyzi.size() / 4;
    // page-locked: the batch's H2D copy then runs as a DMA on its stream
    if (hip_ok(hipHostMalloc((void**)&h_xyzi, std::max<uint64_t>(off[n_local], 1) * 16, hipHostMallocDefault), "hipHostMalloc(cloud)") &&
        hip_ok(hipHostMalloc((void**)&h_click, (size_t)n_local * 12, hipHostMallocDefault), "hipHostMalloc(clicks)")) {
      for (uint32_t f = 0; f < n_local; ++f) {
        const Frame& fr = (*j->frames)[lo + f];
        std::memcpy(h_xyzi + off[f] * 4, fr.xyzi.data(), fr.xyzi.size() * 4);
        std::memcpy(h_click + 3 * f, fr.click, 12);
      }
      h = ilcc_create(j->device, &j->params, n_local, std::max<uint64_t>(off[n_local], 1));
      if (!h) {
        std::fprintf(stderr, "rank %d: ilcc_create: %s\n", j->rank, ilcc_last_error(nullptr));
        failed = 1;
      } else {
        std::vector<ilcc_result> res(n_local);
        int32_t ticket = -1;
        int32_t st = ilcc_submit_batch(h, h_xyzi, off.data(), n_local, h_click, &ticket);
        if (st == ILCC_OK) st = ilcc_wait_records_device(h, ticket, res.data(), d_rec, j->n_corners, /*tag_base=*/lo);
        if (st != ILCC_OK) {
          std::fprintf(stderr, "rank %d: %s %s\n", j->rank, ilcc_strerror(st), ilcc_last_error(h));
          failed = 1;
        }
      }
    }
  }

  // ---- the path's only collective: every rank that has its buffers takes part, failed or not
  if (ready) {
    const ncclResult_t e = ncclGather(d_rec, d_all, rec_floats, ncclFloat, /*root=*/0, j->comm, stream);
    if (e != ncclSuccess) {
      std::fprintf(stderr, "rank %d: ncclGather: %s\n", j->rank, ncclGetErrorString(e));
      failed = 1;
    }
    (void)hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (j->rank == 0 && !failed) {
      j->gathered.resize(rec_floats * (size_t)j->world);
      (void)hip_ok(hipMemcpy(j->gathered.data(), d_all, j->gathered.size() * 4, hipMemcpyDeviceToHost), "hipMemcpy(gathered)");
    }
  } else {
    // no buffers to gather from / into: fail the peers' collective instead of letting it wait for this rank
    (void)ncclCommAbort(j->comm);
    j->comm = nullptr;
  }

  if (h) ilcc_destroy(h);
  if (h_xyzi) (void)hipHostFree(h_xyzi);
  if (h_click) (void)hipHostFree(h_click);
  if (d_all) (void)hipFree(d_all);
  if (d_rec) (void)hipFree(d_rec);
  if (stream) (void)hipStreamDestroy(stream);
  return failed;
}

}  // namespace

int main(int argc, char** argv) {
  (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);   // host-side, before the first HIP call (include/ilcc_hip.h)
  bool accept_ambiguous = false, accept_low_coverage = false;
  const char* prog = argv[0];
  while (argc > 1 && std::strncmp(argv[1], "--", 2) == 0) {
    if (std::strcmp(argv[1], "--accept-ambiguous") == 0) accept_ambiguous = true;
    else if (std::strcmp(argv[1], "--accept-low-coverage") == 0) accept_low_coverage = true;
    else {
      std::fprintf(stderr, "unknown option: %s\n", argv[1]);
      return 2;
    }
    ++argv;
    --argc;
  }
  if (argc < 8 || (argc - 4) % 4 != 0) {
    std::fprintf(stderr, "usage: %s [--accept-ambiguous] [--accept-low-coverage] <yaml|-> <out_prefix> <n_gpus, 0 = all> "
                         "{<cloud.bin> <cx> <cy> <cz>}...\n", prog);
    return 2;
  }
  const std::string yaml = argv[1], prefix = argv[2];
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device: there is no CPU fallback\n");
    return 1;
  }
  int world = std::atoi(argv[3]);
  if (world <= 0 || world > ndev) world = ndev;
  std::vector<Frame> frames((size_t)(argc - 4) / 4);
  for (size_t f = 0; f < frames.size(); ++f) {
    char** a = argv + 4 + 4 * f;
    if (!load(a[0], &frames[f].xyzi)) {
      std::fprintf(stderr, "can not open %s\n", a[0]);
      return 1;
    }
    for (int k = 0; k < 3; ++k) frames[f].click[k] = (float)std::atof(a[1 + k]);
  }
  ilcc_params params;
  ilcc_default_params(&params);
  if (yaml != "-" && ilcc_set_chessboard_param(&params, yaml.c_str()) != ILCC_OK) {
    std::fprintf(stderr, "can not open %s\n", yaml.c_str());
    return 1;
  }
  const uint32_t n_corners = (uint32_t)((params.board_w - 1) * (params.board_h - 1));
  const uint32_t F = (uint32_t)frames.size(), per = (F + (uint32_t)world - 1) / (uint32_t)world;

  std::vector<int> devs(world);
  for (int r = 0; r < world; ++r) devs[r] = r;
  std::vector<ncclComm_t> comms(world);
  if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) {
    std::fprintf(stderr, "ncclCommInitAll failed\n");
    return 1;
  }
  std::vector<RankJob> jobs(world);
  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) {
    RankJob& j = jobs[r];
    j.rank = r;
    j.world = world;
    j.device = devs[r];
    j.comm = comms[r];
    j.frames = &frames;
    j.params = params;
    j.per = per;
    j.n_corners = n_corners;
    j.width = (uint32_t)ILCC_RECORD_HEADER + 3u * n_corners;
    threads.emplace_back([&j]() { j.status = run_rank(&j); });
  }
  for (std::thread& t : threads) t.join();
  for (RankJob& j : jobs)
    if (j.comm) (void)ncclCommDestroy(j.comm);   // (a rank that aborted its communicator has cleared the pointer)
  int n_failed = 0;
  for (const RankJob& j : jobs)
    if (j.status != 0) {
      std::fprintf(stderr, "rank %d (device %d) failed\n", j.rank, j.device);
      ++n_failed;
    }
  if (n_failed) return 1;

  // rank 0: every frame's record must sit at its own position (tag = global frame index) with intact contents
  const RankJob& root = jobs[0];
  uint32_t written = 0, rejected = 0;
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t r = f / per, k = f - r * per;
    const float* rec = root.gathered.data() + ((size_t)r * per + k) * root.width;
    if ((uint32_t)rec[16] != (f & 0xFFFFFFu) || (uint32_t)rec[17] != record_check(rec, n_corners, f)) {
      std::fprintf(stderr, "gathered record %u fails its tag / content check\n", f);
      return 1;
    }
    const int status = (int)rec[0], nc = (int)rec[1];
    const uint32_t flags = (uint32_t)rec[18];
    const std::string file = prefix + "_lidar_" + std::to_string(f + 1) + ".txt";   // get_lidar_corners.cpp:197
    // the accept rule of include/ilcc_hip.h (what LidarCornersEst::get_corners of the class mirror applies)
    const bool accepted = (status == ILCC_OK || (accept_ambiguous && status == ILCC_AMBIGUOUS)) &&
                          (accept_low_coverage || !(flags & ILCC_FLAG_LOW_COVERAGE));
    bool ok = false;
    if (accepted) {
      ok = ilcc_save_corners2txt(rec + ILCC_RECORD_HEADER, (uint32_t)nc, file.c_str()) == ILCC_OK;
      written += ok ? 1u : 0u;
    } else if (status == ILCC_OK || status == ILCC_AMBIGUOUS) {
      ++rejected;
    }
    std::printf("frame %u rank %u status %d flags %u corners %d margin %.6g file %s\n", f + 1, r, status, flags, nc, (double)rec[15],
                ok ? file.c_str() : "-");
  }
  std::printf("gathered %u records from %d GPU(s) with one ncclGather; %u files written\n", F, world, written);
  if (rejected)
    std::printf("%u frame(s) with corners rejected by the accept rule (ambiguous basin / low coverage: --accept-ambiguous, "
                "--accept-low-coverage)\n", rejected);
  return 0;
}
