// bag_reader.cpp -- ROS bag format 2.0 + sensor_msgs/PointCloud2 without ROS (include/ilcc_ingest.h).
// Replaces rosbag::Bag / rosbag::View / instantiate<> as used by
// /root/reference/ilcc2/test/get_lidar_corners.cpp:136-155.  Host only; little-endian host.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ilcc_ingest.h"
#include "ilcc_internal.h"

namespace ilcc {
void set_global_error(const std::string& s);   // ilcc_api.cpp
}

namespace {

constexpr const char* kPointCloud2Md5 = "1158d486dd51d683ce2f1be655c3c181";

struct Fields {
  std::map<std::string, std::string> kv;
  bool has(const char* k) const { return kv.count(k) != 0; }
  template <typename T>
  bool get(const char* k, T* v) const {
    auto it = kv.find(k);
    if (it == kv.end() || it->second.size() != sizeof(T)) return false;
    std::memcpy(v, it->second.data(), sizeof(T));
    return true;
  }
  std::string str(const char* k) const {
    auto it = kv.find(k);
    return it == kv.end() ? std::string() : it->second;
  }
};

// <len:u32><name>=<value> ... over exactly `n` bytes
bool parse_fields(const uint8_t* p, uint64_t n, Fields* f) {
  uint64_t at = 0;
  while (at < n) {
    if (at + 4 > n) return false;
    uint32_t len;
    std::memcpy(&len, p + at, 4);
    at += 4;
    if (len == 0 || at + len > n) return false;
    const uint8_t* eq = (const uint8_t*)std::memchr(p + at, '=', len);
    if (!eq) return false;
    f->kv[std::string((const char*)p + at, eq - (p + at))] = std::string((const char*)eq + 1, (p + at + len) - (eq + 1));
    at += len;
  }
  return true;
}

struct Record {
  Fields hdr;
  const uint8_t* data = nullptr;
  uint32_t data_len = 0;
  uint8_t op = 0;
};

// one record out of a memory range; returns bytes consumed or 0
uint64_t read_record(const uint8_t* p, uint64_t n, Record* r) {
  if (n < 4) return 0;
  uint32_t hl;
  std::memcpy(&hl, p, 4);
  if ((uint64_t)hl + 8 > n) return 0;
  if (!parse_fields(p + 4, hl, &r->hdr)) return 0;
  std::memcpy(&r->data_len, p + 4 + hl, 4);
  if ((uint64_t)hl + 8 + r->data_len > n) return 0;
  r->data = p + 8 + hl;
  if (!r->hdr.get("op", &r->op)) return 0;
  return (uint64_t)hl + 8 + r->data_len;
}

struct File {
  FILE* f = nullptr;
  uint64_t size = 0;   // set once after fopen: every length field of the file is checked against it BEFORE it sizes a buffer
  ~File() {
    if (f) std::fclose(f);
  }
  bool measure() {
    if (fseeko(f, 0, SEEK_END) != 0) return false;
    const off_t e = ftello(f);
    if (e < 0) return false;
    size = (uint64_t)e;
    return true;
  }
  // [pos, pos + n) lies inside the file (no wrap-around)
  bool holds(uint64_t pos, uint64_t n) const { return pos <= size && n <= size - pos; }
  bool read_at(uint64_t pos, void* dst, uint64_t n) {
    if (!holds(pos, n)) return false;
    if (fseeko(f, (off_t)pos, SEEK_SET) != 0) return false;
    return std::fread(dst, 1, n, f) == n;
  }
  // record header + data length at pos (data itself not loaded)
  bool record_at(uint64_t pos, Fields* hdr, uint32_t* data_len, uint64_t* data_pos) {
    uint32_t hl;
    if (!read_at(pos, &hl, 4) || hl > (1u << 20)) return false;
    std::vector<uint8_t> h(hl);
    if (hl && !read_at(pos + 4, h.data(), hl)) return false;
    if (!parse_fields(h.data(), hl, hdr)) return false;
    if (!read_at(pos + 4 + hl, data_len, 4)) return false;
    *data_pos = pos + 8 + hl;
    return holds(*data_pos, *data_len);   // a data length that runs past the end of the file is a truncated / forged record
  }
};

typedef int (*bz2_fn)(char*, unsigned int*, char*, unsigned int, int, int);
bz2_fn bz2_decompress() {
  static bz2_fn fn = []() -> bz2_fn {
    for (const char* name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL))
        if (void* s = dlsym(h, "BZ2_bzBuffToBuffDecompress")) return (bz2_fn)s;
    }
    return nullptr;
  }();
  return fn;
}

// LZ4 frame format (what roslz4 writes) through the system liblz4's frame API
struct Lz4 {
  typedef size_t (*create_fn)(void**, unsigned);
  typedef size_t (*free_fn)(void*);
  typedef size_t (*dec_fn)(void*, void*, size_t*, const void*, size_t*, const void*);
  typedef unsigned (*iserr_fn)(size_t);
  create_fn create = nullptr;
  free_fn release = nullptr;
  dec_fn dec = nullptr;
  iserr_fn iserr = nullptr;
  bool ok() const { return create && release && dec && iserr; }
};
const Lz4& lz4() {
  static Lz4 L = []() {
    Lz4 l;
    for (const char* name : {"liblz4.so.1", "liblz4.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        l.create = (Lz4::create_fn)dlsym(h, "LZ4F_createDecompressionContext");
        l.release = (Lz4::free_fn)dlsym(h, "LZ4F_freeDecompressionContext");
        l.dec = (Lz4::dec_fn)dlsym(h, "LZ4F_decompress");
        l.iserr = (Lz4::iserr_fn)dlsym(h, "LZ4F_isError");
        if (l.ok()) break;
      }
    }
    return l;
  }();
  return L;
}

bool inflate_chunk(const std::string& compression, std::vector<uint8_t>& raw, uint32_t size, std::vector<uint8_t>* out,
                   std::string* err) {
  if (compression == "none") {
    if ((uint64_t)size != raw.size()) {   // the chunk header's `size` is the uncompressed length: for "none" it IS the data length
      *err = "uncompressed chunk whose size field disagrees with its data length";
      return false;
    }
    out->swap(raw);
    return true;
  }
  // compressed chunk: bz2 / lz4 cannot expand more than ~255x / ~255x of their input; refuse declared sizes beyond that
  // bound (and beyond 1 GiB) instead of letting a 30-byte file ask for 4 GiB
  if ((uint64_t)size > (1ull << 30) || (uint64_t)size > 1024ull + 1024ull * (uint64_t)raw.size()) {
    *err = "chunk declares an implausible uncompressed size";
    return false;
  }
  out->resize(size);
  if (compression == "bz2") {
    bz2_fn fn = bz2_decompress();
    if (!fn) {
      *err = "bz2 chunk but libbz2 is not loadable";
      return false;
    }
    unsigned int dl = size;
    if (fn((char*)out->data(), &dl, (char*)raw.data(), (unsigned int)raw.size(), 0, 0) != 0 || dl != size) {
      *err = "bz2 chunk does not decompress";
      return false;
    }
    return true;
  }
  if (compression == "lz4") {
    const Lz4& L = lz4();
    if (!L.ok()) {
      *err = "lz4 chunk but liblz4's frame API is not loadable";
      return false;
    }
    void* ctx = nullptr;
    if (L.iserr(L.create(&ctx, 100))) {
      *err = "lz4 context";
      return false;
    }
    size_t in_at = 0, out_at = 0;
    bool good = true;
    while (in_at < raw.size()) {
      size_t dn = size - out_at, sn = raw.size() - in_at;
      const size_t rc = L.dec(ctx, out->data() + out_at, &dn, raw.data() + in_at, &sn, nullptr);
      if (L.iserr(rc) || (dn == 0 && sn == 0)) {
        good = false;
        break;
      }
      out_at += dn;
      in_at += sn;
      if (rc == 0) break;
    }
    L.release(ctx);
    if (!good || out_at != size) {
      *err = "lz4 chunk does not decompress";
      return false;
    }
    return true;
  }
  *err = "unknown chunk compression '" + compression + "'";
  return false;
}

struct ChunkRef {
  uint64_t pos = 0, start = 0;
  uint32_t order = 0;
};

uint64_t time_key(uint64_t ros_time) {   // wire: sec (low 32), nsec (high 32) -> sortable
  return ((ros_time & 0xffffffffull) << 32) | (ros_time >> 32);
}

int32_t fail(int32_t code, const std::string& what) {
  ilcc::set_global_error(what);
  return code;
}

}  // namespace

extern "C" {

static int32_t bag_first_message_impl(const char* bag_path, const char* topic, const char* md5sum, uint8_t* msg, uint64_t cap,
                                      uint64_t* msg_bytes) {
  if (!bag_path || !topic || !msg_bytes || (!msg && cap)) return fail(ILCC_BAD_ARGUMENT, "null argument");
  const std::string want_md5 = md5sum ? md5sum : kPointCloud2Md5;
  *msg_bytes = 0;
  File file;
  file.f = std::fopen(bag_path, "rb");
  if (!file.f) return fail(ILCC_IO_ERROR, std::string("cannot open ") + bag_path);
  if (!file.measure()) return fail(ILCC_IO_ERROR, "cannot determine the size of the bag file");
  char magic[13];
  if (!file.read_at(0, magic, 13) || std::memcmp(magic, "#ROSBAG V2.0\n", 13) != 0)
    return fail(ILCC_IO_ERROR, "not a ROS bag V2.0");
  Fields bh;
  uint32_t dl;
  uint64_t dpos;
  uint8_t op = 0;
  if (!file.record_at(13, &bh, &dl, &dpos) || !bh.get("op", &op) || op != 0x03)
    return fail(ILCC_IO_ERROR, "bag header record missing");
  uint64_t index_pos = 0;
  uint32_t conn_count = 0, chunk_count = 0;
  bh.get("index_pos", &index_pos);
  bh.get("conn_count", &conn_count);
  bh.get("chunk_count", &chunk_count);
  if (index_pos == 0) return fail(ILCC_IO_ERROR, "bag is unindexed (rosbag refuses it too: run rosbag reindex)");

  // index section: connection records, then chunk infos
  std::map<uint32_t, bool> conn_ok;   // connections on the topic -> md5 matches
  std::vector<ChunkRef> chunks;
  uint64_t pos = index_pos;
  const uint64_t n_index_records = (uint64_t)conn_count + (uint64_t)chunk_count;   // (a u32 sum could wrap)
  if (n_index_records > file.size / 8) return fail(ILCC_IO_ERROR, "bag header declares more index records than the file can hold");
  for (uint64_t k = 0; k < n_index_records; ++k) {
    Fields h;
    if (!file.record_at(pos, &h, &dl, &dpos) || !h.get("op", &op)) return fail(ILCC_IO_ERROR, "index section truncated");
    std::vector<uint8_t> d(dl);
    if (dl && !file.read_at(dpos, d.data(), dl)) return fail(ILCC_IO_ERROR, "index section truncated");
    if (op == 0x07) {
      uint32_t conn;
      Fields ch;
      if (!h.get("conn", &conn) || !parse_fields(d.data(), dl, &ch)) return fail(ILCC_IO_ERROR, "bad connection record");
      if (h.str("topic") == topic) {   // rosbag::TopicQuery compares the connection's topic (record header)
        const std::string md5 = ch.str("md5sum");
        conn_ok[conn] = (md5 == want_md5) || md5 == "*" || want_md5 == "*";
      }
    } else if (op == 0x06) {
      ChunkRef c;
      uint32_t count = 0;
      if (!h.get("chunk_pos", &c.pos) || !h.get("start_time", &c.start) || !h.get("count", &count))
        return fail(ILCC_IO_ERROR, "bad chunk info record");
      c.start = time_key(c.start);
      bool has = false;
      for (uint32_t i = 0; i + 1 <= count && (uint64_t)(i + 1) * 8 <= dl; ++i) {
        uint32_t conn, n;
        std::memcpy(&conn, d.data() + 8 * i, 4);
        std::memcpy(&n, d.data() + 8 * i + 4, 4);
        if (n && conn_ok.count(conn)) has = true;
      }
      c.order = (uint32_t)chunks.size();
      if (has) chunks.push_back(c);
    } else {
      return fail(ILCC_IO_ERROR, "unexpected record in the index section");
    }
    pos = dpos + dl;
  }
  std::stable_sort(chunks.begin(), chunks.end(), [](const ChunkRef& a, const ChunkRef& b) { return a.start < b.start; });

  // View order = message time; a chunk can hold an earlier message only if it starts no later than the best so far
  bool found = false;
  uint64_t best_time = ~0ull;
  std::vector<uint8_t> best;
  for (const ChunkRef& c : chunks) {
    if (found && c.start > best_time) break;
    Fields h;
    if (!file.record_at(c.pos, &h, &dl, &dpos) || !h.get("op", &op) || op != 0x05) return fail(ILCC_IO_ERROR, "chunk record missing");
    uint32_t size = 0;
    h.get("size", &size);
    std::vector<uint8_t> raw(dl), plain;
    if (dl && !file.read_at(dpos, raw.data(), dl)) return fail(ILCC_IO_ERROR, "chunk truncated");
    std::string err;
    if (!inflate_chunk(h.str("compression"), raw, size, &plain, &err)) return fail(ILCC_IO_ERROR, err);
    uint64_t at = 0;
    while (at < plain.size()) {
      Record r;
      const uint64_t used = read_record(plain.data() + at, plain.size() - at, &r);
      if (!used) return fail(ILCC_IO_ERROR, "bad record inside a chunk");
      at += used;
      if (r.op != 0x02) continue;
      uint32_t conn;
      uint64_t t;
      if (!r.hdr.get("conn", &conn) || !r.hdr.get("time", &t)) return fail(ILCC_IO_ERROR, "bad message record");
      auto it = conn_ok.find(conn);
      if (it == conn_ok.end() || !it->second) continue;   // other topic, or instantiate<>() would return NULL
      const uint64_t tk = time_key(t);
      if (!found || tk < best_time) {
        found = true;
        best_time = tk;
        best.assign(r.data, r.data + r.data_len);
      }
    }
  }
  if (!found) return fail(ILCC_BAD_ARGUMENT, std::string("no message of that type on topic ") + topic);
  *msg_bytes = best.size();
  if (best.size() > cap) return fail(ILCC_CAPACITY, "message larger than the buffer");
  std::memcpy(msg, best.data(), best.size());
  return ILCC_OK;
}

// No exception crosses the C-ABI: allocation failures and anything else a malformed file provokes become a status.
int32_t ilcc_bag_first_message(const char* bag_path, const char* topic, const char* md5sum, uint8_t* msg, uint64_t cap,
                               uint64_t* msg_bytes) {
  try {
    return bag_first_message_impl(bag_path, topic, md5sum, msg, cap, msg_bytes);
  } catch (const std::bad_alloc&) {
    return fail(ILCC_IO_ERROR, "out of memory while reading the bag");
  } catch (const std::exception& e) {
    return fail(ILCC_IO_ERROR, std::string("bag reader: ") + e.what());
  } catch (...) {
    return fail(ILCC_IO_ERROR, "bag reader: unknown failure");
  }
}

int32_t ilcc_pointcloud2_parse(const uint8_t* m, uint64_t n, ilcc_pointcloud2_layout* out) {
  if (!m || !out) return fail(ILCC_BAD_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof(*out));
  out->off_x = out->off_y = out->off_z = out->off_intensity = ILCC_FIELD_ABSENT;
  uint64_t at = 0;
  bool ok = true;
  auto u32 = [&]() -> uint32_t {
    uint32_t v = 0;
    if (at + 4 > n) { ok = false; return 0; }
    std::memcpy(&v, m + at, 4);
    at += 4;
    return v;
  };
  auto u8 = [&]() -> uint8_t {
    if (at + 1 > n) { ok = false; return 0; }
    return m[at++];
  };
  auto str = [&]() -> std::string {
    const uint32_t l = u32();
    if (!ok || at + l > n) { ok = false; return std::string(); }
    std::string s((const char*)m + at, l);
    at += l;
    return s;
  };
  out->seq = u32();
  out->stamp_sec = u32();
  out->stamp_nsec = u32();
  const std::string frame = str();
  std::snprintf(out->frame_id, sizeof(out->frame_id), "%s", frame.c_str());
  out->height = u32();
  out->width = u32();
  out->n_fields = u32();
  if (!ok || out->n_fields > 4096) return fail(ILCC_BAD_ARGUMENT, "not a PointCloud2 message");
  for (uint32_t k = 0; k < out->n_fields && ok; ++k) {
    const std::string name = str();
    const uint32_t offset = u32();
    const uint8_t datatype = u8();
    const uint32_t count = u32();
    // pcl::FieldMatches<PointXYZI, tag>: name, datatype FLOAT32 (7) and count 1 must all agree
    if (datatype != 7 || count != 1) continue;
    if (name == "x") out->off_x = offset;
    else if (name == "y") out->off_y = offset;
    else if (name == "z") out->off_z = offset;
    else if (name == "intensity") out->off_intensity = offset;
  }
  out->is_bigendian = u8();
  out->point_step = u32();
  out->row_step = u32();
  out->data_bytes = u32();
  out->data_offset = at;
  if (!ok || at + out->data_bytes > n) return fail(ILCC_BAD_ARGUMENT, "PointCloud2 message truncated");
  at += out->data_bytes;
  out->is_dense = u8();
  if (!ok) return fail(ILCC_BAD_ARGUMENT, "PointCloud2 message truncated");
  const uint64_t pts = (uint64_t)out->height * out->width;
  if (pts) {
    if ((uint64_t)out->width * out->point_step > out->row_step ||
        (uint64_t)(out->height - 1) * out->row_step + (uint64_t)out->width * out->point_step > out->data_bytes)
      return fail(ILCC_BAD_ARGUMENT, "PointCloud2 steps do not fit data[]");
    for (uint32_t off : {out->off_x, out->off_y, out->off_z, out->off_intensity})
      if (off != ILCC_FIELD_ABSENT && (uint64_t)off + 4 > out->point_step)
        return fail(ILCC_BAD_ARGUMENT, "PointCloud2 field outside point_step");
  }
  return ILCC_OK;
}

}  // extern "C"
