// ilcc_corners_mgpu -- the multi-GPU form of the path in C++ (north_star: "host code stays C++ ... independent frames
// shard embarrassingly across the GPUs of one node with a single RCCL gather of per-frame corners over xGMI").
// One process, one host thread per GPU, one ilcc_handle per thread.  Frames (bag x click pairs are independent:
// /root/reference/ilcc2/test/get_lidar_corners.cpp:130-211 loops over them and overwrites every member per frame) are
// split into contiguous shards of ceil(F / G); every rank runs the whole path on its shard (H2D copy on the batch's own
// stream, records packed on the GPU by K9) and ONE ncclGather (rccl.h) ships the fixed-size records from each GPU's HBM
// to rank 0's, which checks tag + content word of every record and writes the process_data files.
//   ilcc_corners_mgpu <yaml|-> <out_prefix> <n_gpus, 0 = all> {<cloud.bin> <cx> <cy> <cz>}...
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

#define CHECK_HIP(e)                                                            \
  do {                                                                          \
    hipError_t e_ = (e);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));              \
      return 1;                                                                 \
    }                                                                           \
  } while (0)
#define CHECK_NCCL(e)                                                           \
  do {                                                                          \
    ncclResult_t e_ = (e);                                                      \
    if (e_ != ncclSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #e, ncclGetErrorString(e_));             \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

struct RankJob {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  const std::vector<Frame>* frames = nullptr;
  ilcc_params params{};
  uint32_t per = 0, n_corners = 0, width = 0;
  std::vector<float> gathered;   // rank 0: world * per * width
  int status = 0;
};

int run_rank(RankJob* j) {
  CHECK_HIP(hipSetDevice(j->device));
  const uint32_t F = (uint32_t)j->frames->size();
  const uint32_t lo = std::min(F, (uint32_t)j->rank * j->per), hi = std::min(F, lo + j->per);
  const uint32_t n_local = hi - lo;
  hipStream_t stream;
  CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  float *d_rec = nullptr, *d_all = nullptr;
  const size_t rec_floats = (size_t)j->per * j->width;
  CHECK_HIP(hipMalloc((void**)&d_rec, rec_floats * 4));
  if (j->rank == 0) CHECK_HIP(hipMalloc((void**)&d_all, rec_floats * 4 * (size_t)j->world));
  // shards are padded to `per` records; a padding record carries status -1
  std::vector<float> pad(rec_floats, 0.f);
  for (uint32_t f = 0; f < j->per; ++f) pad[(size_t)f * j->width] = -1.f;
  CHECK_HIP(hipMemcpy(d_rec, pad.data(), rec_floats * 4, hipMemcpyHostToDevice));
  if (n_local > 0) {
    std::vector<uint64_t> off(n_local + 1, 0);
    for (uint32_t f = 0; f < n_local; ++f) off[f + 1] = off[f] + (*j->frames)[lo + f].xyzi.size() / 4;
    float *h_xyzi = nullptr, *h_click = nullptr;   // page-locked: the batch's H2D copy then runs as a DMA on its stream
    CHECK_HIP(hipHostMalloc((void**)&h_xyzi, std::max<uint64_t>(off[n_local], 1) * 16, hipHostMallocDefault));
    CHECK_HIP(hipHostMalloc((void**)&h_click, (size_t)n_local * 12, hipHostMallocDefault));
    for (uint32_t f = 0; f < n_local; ++f) {
      const Frame& fr = (*j->frames)[lo + f];
      std::memcpy(h_xyzi + off[f] * 4, fr.xyzi.data(), fr.xyzi.size() * 4);
      std::memcpy(h_click + 3 * f, fr.click, 12);
    }
    ilcc_handle* h = ilcc_create(j->device, &j->params, n_local, std::max<uint64_t>(off[n_local], 1));
    if (!h) {
      std::fprintf(stderr, "rank %d: ilcc_create: %s\n", j->rank, ilcc_last_error(nullptr));
      return 1;
    }
    std::vector<ilcc_result> res(n_local);
    int32_t ticket = -1;
    int32_t st = ilcc_submit_batch(h, h_xyzi, off.data(), n_local, h_click, &ticket);
    if (st == ILCC_OK) st = ilcc_wait_records_device(h, ticket, res.data(), d_rec, j->n_corners, /*tag_base=*/lo);
    if (st != ILCC_OK) {
      std::fprintf(stderr, "rank %d: %s %s\n", j->rank, ilcc_strerror(st), ilcc_last_error(h));
      return 1;
    }
    ilcc_destroy(h);
    (void)hipHostFree(h_xyzi);
    (void)hipHostFree(h_click);
  }
  // the path's only collective
  CHECK_NCCL(ncclGather(d_rec, d_all, rec_floats, ncclFloat, /*root=*/0, j->comm, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  if (j->rank == 0) {
    j->gathered.resize(rec_floats * (size_t)j->world);
    CHECK_HIP(hipMemcpy(j->gathered.data(), d_all, j->gathered.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_all);
  }
  (void)hipFree(d_rec);
  (void)hipStreamDestroy(stream);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8 || (argc - 4) % 4 != 0) {
    std::fprintf(stderr, "usage: %s <yaml|-> <out_prefix> <n_gpus, 0 = all> {<cloud.bin> <cx> <cy> <cz>}...\n", argv[0]);
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
  for (ncclComm_t c : comms) (void)ncclCommDestroy(c);
  for (const RankJob& j : jobs)
    if (j.status != 0) return 1;

  // rank 0: every frame's record must sit at its own position (tag = global frame index) with intact contents
  const RankJob& root = jobs[0];
  uint32_t written = 0;
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t r = f / per, k = f - r * per;
    const float* rec = root.gathered.data() + ((size_t)r * per + k) * root.width;
    if ((uint32_t)rec[16] != (f & 0xFFFFFFu) || (uint32_t)rec[17] != record_check(rec, n_corners, f)) {
      std::fprintf(stderr, "gathered record %u fails its tag / content check\n", f);
      return 1;
    }
    const int status = (int)rec[0], nc = (int)rec[1];
    const std::string file = prefix + "_lidar_" + std::to_string(f + 1) + ".txt";   // get_lidar_corners.cpp:197
    bool ok = false;
    if (status == ILCC_OK) {
      ok = ilcc_save_corners2txt(rec + ILCC_RECORD_HEADER, (uint32_t)nc, file.c_str()) == ILCC_OK;
      written += ok ? 1u : 0u;
    }
    std::printf("frame %u rank %u status %d corners %d margin %.6g file %s\n", f + 1, r, status, nc, (double)rec[15],
                ok ? file.c_str() : "-");
  }
  std::printf("gathered %u records from %d GPU(s) with one ncclGather; %u files written\n", F, world, written);
  return 0;
}
