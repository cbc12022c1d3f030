// HDF5 subset reader for Caffe weight snapshots (see caffe/util/hdf5_lite.hpp).  Written from the published HDF5 File
// Format Specification (version 1.1 structures: superblock v0 / v1, v1 B-tree group nodes, symbol-table nodes, local heaps,
// version-1 object headers; header messages 0x0001 dataspace, 0x0003 datatype, 0x0008 layout, 0x0010 continuation,
// 0x0011 symbol table).
#include "caffe/util/hdf5_lite.hpp"

#include <cstring>
#include <sstream>

namespace caffe {
namespace h5lite {

namespace {
const unsigned char kSignature[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
const int kMaxDepth = 32;            // B-tree levels / continuation chains: far above any real file, stops cycles
template <typename T>
std::string Str(const T& v) { std::ostringstream o; o << v; return o.str(); }
}  // namespace

bool File::Fail(const std::string& what) {
  if (err_.empty()) err_ = "HDF5: " + what;
  ok_ = false;
  return false;
}

bool File::Need(uint64_t off, uint64_t n, const char* what) {
  if (off > n_ || n > n_ - off) return Fail(std::string("truncated or corrupt file (") + what + " at offset " + Str(off) + ")");
  return true;
}

uint64_t File::U(uint64_t off, int nbytes) const {
  uint64_t v = 0;
  for (int i = nbytes - 1; i >= 0; --i) v = (v << 8) | p_[off + i];
  return v;
}

// an address field: relative to the base address; all ones = undefined
bool File::Addr(uint64_t off, uint64_t* a, const char* what) {
  if (!Need(off, so_, what)) return false;
  const uint64_t v = U(off, so_);
  const uint64_t undef = so_ == 8 ? ~0ull : ((1ull << (8 * so_)) - 1ull);
  if (v == undef) { *a = ~0ull; return true; }
  if (v > n_ || base_ > n_ - v) return Fail(std::string("address out of range (") + what + ")");
  *a = base_ + v;
  return true;
}

File::File(const std::string& bytes)
    : p_(reinterpret_cast<const unsigned char*>(bytes.data())), n_(bytes.size()), ok_(true), so_(8), sl_(8), base_(0),
      root_header_(0), budget_(0) {
  // the superblock sits at 0 or at 512 << k (a user block may precede it)
  uint64_t sb = ~0ull;
  for (uint64_t off = 0; off + 8 <= n_; off = off ? off * 2 : 512)
    if (std::memcmp(p_ + off, kSignature, 8) == 0) { sb = off; break; }
  if (sb == ~0ull) { Fail("not an HDF5 file (no superblock signature)"); return; }
  if (!Need(sb, 24, "superblock")) return;
  const int version = p_[sb + 8];
  if (version > 1) {
    Fail("superblock version " + Str(version) + " (a file written with libver=latest) is outside the subset this reader supports; "
         "Caffe writes version 0");
    return;
  }
  so_ = p_[sb + 13];
  sl_ = p_[sb + 14];
  if ((so_ != 4 && so_ != 8) || (sl_ != 4 && sl_ != 8)) { Fail("unsupported size of offsets / lengths"); return; }
  uint64_t o = sb + 24 + (version == 1 ? 4 : 0);
  if (!Need(o, 4ull * so_ + 2ull * so_ + 24, "superblock")) return;
  base_ = 0;
  uint64_t b;
  if (!Addr(o, &b, "base address")) return;
  base_ = b;
  o += 4ull * so_;                                  // base, free-space info, end of file, driver info
  // root group symbol table entry: link name offset, object header address, cache type, reserved, scratch pad
  if (!Addr(o + so_, &root_header_, "root group object header")) return;
  if (root_header_ == ~0ull) Fail("root group has no object header");
}

bool File::Messages(uint64_t header, std::vector<Message>* out) {
  out->clear();
  if (!Need(header, 16, "object header")) return false;
  if (std::memcmp(p_ + header, "OHDR", 4) == 0)
    return Fail("version-2 object header (a file written with libver=latest) is outside the subset this reader supports");
  if (p_[header] != 1) return Fail("object header version " + Str((int)p_[header]) + " at offset " + Str(header));
  const unsigned total = (unsigned)U(header + 2, 2);
  uint64_t chunk = header + 16, chunk_bytes = U(header + 8, 4);
  std::vector<std::pair<uint64_t, uint64_t> > pending;      // continuation blocks (offset, bytes)
  for (int guard = 0; guard < kMaxDepth * 8; ++guard) {
    if (!Need(chunk, chunk_bytes, "object header messages")) return false;
    uint64_t o = chunk;
    const uint64_t end = chunk + chunk_bytes;
    while (o + 8 <= end && out->size() < total) {
      Message m;
      m.type = (int)U(o, 2);
      m.size = U(o + 2, 2);
      m.offset = o + 8;
      if (m.size > end - m.offset) return Fail("header message runs past its chunk");
      if (m.type == 0x0010) {                                 // object header continuation: offset, length
        uint64_t a;
        if (m.size < (uint64_t)so_ + sl_) return Fail("short continuation message");
        if (!Addr(m.offset, &a, "continuation block")) return false;
        if (a == ~0ull) return Fail("continuation block has an undefined address");
        pending.push_back(std::make_pair(a, U(m.offset + so_, sl_)));
      }
      out->push_back(m);
      o = m.offset + m.size;
    }
    if (pending.empty() || out->size() >= total) return true;
    chunk = pending.front().first;
    chunk_bytes = pending.front().second;
    pending.erase(pending.begin());
  }
  return Fail("object header continuation chain too long");
}

bool File::SymbolTable(uint64_t group_header, uint64_t* btree, uint64_t* heap) {
  std::vector<Message> ms;
  if (!Messages(group_header, &ms)) return false;
  for (size_t i = 0; i < ms.size(); ++i) {
    if (ms[i].type == 0x0011) {
      if (ms[i].size < 2ull * so_) return Fail("short symbol table message");
      return Addr(ms[i].offset, btree, "group B-tree") && Addr(ms[i].offset + so_, heap, "group local heap");
    }
    if (ms[i].type == 0x0002 || ms[i].type == 0x0006)
      return Fail("link-message (\"new style\") group: written with libver=latest, outside the subset this reader supports");
  }
  return Fail("object at offset " + Str(group_header) + " is not a group");
}

bool File::WalkBtree(uint64_t node, uint64_t heap_data, uint64_t heap_size, int depth,
                     std::vector<std::pair<std::string, uint64_t> >* links) {
  if (depth > kMaxDepth) return Fail("group B-tree too deep");
  if (budget_ == 0) return Fail("group B-tree visits more nodes than a file of this size can hold (cycle?)");
  --budget_;
  const uint64_t hdr = 8 + 2ull * so_;
  if (!Need(node, hdr, "B-tree node")) return false;
  if (std::memcmp(p_ + node, "TREE", 4) != 0) return Fail("bad B-tree node signature");
  if (p_[node + 4] != 0) return Fail("B-tree node is not a group node");
  const int level = p_[node + 5];
  const unsigned used = (unsigned)U(node + 6, 2);
  if (!Need(node + hdr, (uint64_t)used * (sl_ + so_) + sl_, "B-tree entries")) return false;
  for (unsigned i = 0; i < used; ++i) {
    uint64_t child;
    if (!Addr(node + hdr + (uint64_t)i * (sl_ + so_) + sl_, &child, "B-tree child")) return false;
    if (child == ~0ull) return Fail("B-tree child has an undefined address");
    if (level > 0) {
      if (!WalkBtree(child, heap_data, heap_size, depth + 1, links)) return false;
      continue;
    }
    // symbol table node
    if (budget_ == 0) return Fail("group B-tree visits more nodes than a file of this size can hold (cycle?)");
    --budget_;
    if (!Need(child, 8, "symbol table node")) return false;
    if (std::memcmp(p_ + child, "SNOD", 4) != 0) return Fail("bad symbol table node signature");
    const unsigned nsym = (unsigned)U(child + 6, 2);
    const uint64_t esz = 2ull * so_ + 24;
    if (!Need(child + 8, nsym * esz, "symbol table entries")) return false;
    for (unsigned s = 0; s < nsym; ++s) {
      const uint64_t e = child + 8 + s * esz;
      const uint64_t name_off = U(e, so_);
      uint64_t obj;
      if (!Addr(e + so_, &obj, "symbol table entry")) return false;
      if (name_off >= heap_size) return Fail("link name outside the local heap");
      const unsigned char* s0 = p_ + heap_data + name_off;
      const void* z = std::memchr(s0, 0, heap_size - name_off);
      if (!z) return Fail("unterminated link name");
      links->push_back(std::make_pair(std::string(reinterpret_cast<const char*>(s0), (const char*)z - (const char*)s0), obj));
    }
  }
  return true;
}

bool File::ListGroup(uint64_t group_header, std::vector<std::pair<std::string, uint64_t> >* links) {
  links->clear();
  if (!ok_) return false;
  uint64_t btree, heap;
  if (!SymbolTable(group_header, &btree, &heap)) return false;
  if (btree == ~0ull || heap == ~0ull) return Fail("group without a B-tree / heap");
  if (!Need(heap, 8 + 2ull * sl_ + so_, "local heap")) return false;
  if (std::memcmp(p_ + heap, "HEAP", 4) != 0) return Fail("bad local heap signature");
  const uint64_t heap_size = U(heap + 8, sl_);
  uint64_t heap_data;
  if (!Addr(heap + 8 + 2ull * sl_, &heap_data, "local heap data")) return false;
  if (heap_data == ~0ull || !Need(heap_data, heap_size, "local heap data")) return Fail("local heap data out of range");
  budget_ = n_ / 32 + 16;                // every node occupies more than 32 bytes: a walk that needs more has a cycle
  return WalkBtree(btree, heap_data, heap_size, 0, links);
}

bool File::Find(uint64_t group_header, const std::string& name, uint64_t* object_header, bool* found) {
  std::vector<std::pair<std::string, uint64_t> > links;
  *found = false;
  if (!ListGroup(group_header, &links)) return false;
  for (size_t i = 0; i < links.size(); ++i)
    if (links[i].first == name) { *object_header = links[i].second; *found = true; break; }
  return true;
}

bool File::ReadDatasetInfo(uint64_t object_header, Dataset* ds) {
  *ds = Dataset();
  if (!ok_) return false;
  std::vector<Message> ms;
  if (!Messages(object_header, &ms)) return false;
  bool have_space = false, have_type = false, have_layout = false;
  for (size_t i = 0; i < ms.size(); ++i) {
    const Message& m = ms[i];
    const uint64_t o = m.offset;
    if (m.type == 0x0001) {                                   // dataspace
      if (m.size < 8) return Fail("short dataspace message");
      const int version = p_[o], rank = p_[o + 1];
      if (version != 1 && version != 2) return Fail("dataspace message version " + Str(version));
      if (version == 2 && p_[o + 3] == 2) return Fail("null dataspace");
      const uint64_t d0 = o + (version == 1 ? 8 : 4);
      if (rank > 32 || (uint64_t)rank * sl_ > m.size - (d0 - o)) return Fail("dataspace rank / size mismatch");
      for (int r = 0; r < rank; ++r) {
        const uint64_t d = U(d0 + (uint64_t)r * sl_, sl_);
        if (d > (1ull << 40)) return Fail("dataset dimension out of range");
        ds->dims.push_back((long long)d);
      }
      have_space = true;
    } else if (m.type == 0x0003) {                            // datatype
      if (m.size < 8) return Fail("short datatype message");
      ds->type_class = p_[o] & 0x0f;
      ds->type_size = (int)U(o + 4, 4);
      ds->big_endian = (p_[o + 1] & 1) != 0;
      if (ds->type_class == 0) {
        ds->is_signed = (p_[o + 1] & 8) != 0;
        if (ds->type_size != 1 && ds->type_size != 2 && ds->type_size != 4 && ds->type_size != 8)
          return Fail("fixed-point type of " + Str(ds->type_size) + " bytes");
      } else if (ds->type_class == 1) {
        if (m.size < 20) return Fail("short floating-point datatype message");
        const int esize = p_[o + 13], msize = p_[o + 15];
        const bool f32 = ds->type_size == 4 && esize == 8 && msize == 23, f64 = ds->type_size == 8 && esize == 11 && msize == 52;
        if (!f32 && !f64) return Fail("floating-point type other than IEEE binary32 / binary64");
        if (p_[o + 1] & 0x40) return Fail("VAX byte order");
      } else {
        // hdf5_load_nd_dataset_helper (util/hdf5.cpp:32-59): only H5T_FLOAT / H5T_INTEGER are accepted
        return Fail("unsupported datatype class " + Str(ds->type_class) + " (only H5T_FLOAT / H5T_INTEGER, util/hdf5.cpp:32-59)");
      }
      have_type = true;
    } else if (m.type == 0x0008) {                            // data layout
      if (m.size < 2) return Fail("short layout message");
      const int version = p_[o];
      if (version == 3 || version == 4) {
        const int cls = p_[o + 1];
        if (cls == 0) {
          if (m.size < 4) return Fail("short compact layout");
          ds->data_bytes = U(o + 2, 2);
          ds->data_offset = o + 4;
          if (ds->data_bytes > m.size - 4) return Fail("compact data runs past its message");
        } else if (cls == 1) {
          if (m.size < 2ull + so_ + sl_) return Fail("short contiguous layout");
          if (!Addr(o + 2, &ds->data_offset, "dataset storage")) return false;
          ds->data_bytes = U(o + 2 + so_, sl_);
        } else {
          return Fail("chunked / virtual dataset layout (Caffe writes contiguous datasets) is outside the subset this reader supports");
        }
      } else if (version == 1 || version == 2) {
        if (m.size < 8) return Fail("short layout message");
        const int rank = p_[o + 1], cls = p_[o + 2];
        if (cls != 1) return Fail("layout version " + Str(version) + " with a non-contiguous class");
        if (m.size < 8ull + so_ + 4ull * rank) return Fail("short layout message");
        if (!Addr(o + 8, &ds->data_offset, "dataset storage")) return false;
        ds->data_bytes = ~0ull;                               // implied by dataspace x datatype (checked below)
      } else {
        return Fail("data layout message version " + Str(version));
      }
      have_layout = true;
    } else if (m.type == 0x000B) {
      return Fail("filtered (compressed) dataset is outside the subset this reader supports");
    }
  }
  if (!have_space || !have_type || !have_layout) return Fail("object at offset " + Str(object_header) + " is not a dataset");
  if (ds->count() > Dataset::kMaxCount) return Fail("dataset has more than INT_MAX elements (or a dimension product that overflows)");
  const uint64_t want = (uint64_t)ds->count() * (uint64_t)ds->type_size;
  if (ds->data_bytes == ~0ull) ds->data_bytes = want;
  if (want == 0) return true;
  if (ds->data_offset == ~0ull) return Fail("dataset has no storage allocated (never written)");
  if (ds->data_bytes < want) return Fail("dataset storage smaller than its dataspace");
  return Need(ds->data_offset, want, "dataset data");
}

bool File::ReadFloats(const Dataset& ds, float* out) {
  if (!ok_) return false;
  const long long n = ds.count();
  const unsigned char* src = p_ + ds.data_offset;
  const int sz = ds.type_size;
  if (!Need(ds.data_offset, (uint64_t)n * sz, "dataset data")) return false;
  for (long long i = 0; i < n; ++i) {
    unsigned char b[8];
    for (int k = 0; k < sz; ++k) b[k] = ds.big_endian ? src[i * sz + (sz - 1 - k)] : src[i * sz + k];    // -> little endian
    if (ds.type_class == 1) {
      if (sz == 4) { std::memcpy(out + i, b, 4); }
      else { double d; std::memcpy(&d, b, 8); out[i] = (float)d; }
    } else {
      uint64_t u = 0;
      for (int k = sz - 1; k >= 0; --k) u = (u << 8) | b[k];
      if (ds.is_signed) {
        int64_t s = (int64_t)u;
        if (sz < 8 && (u >> (8 * sz - 1))) s = (int64_t)(u | (~0ull << (8 * sz)));
        out[i] = (float)s;
      } else {
        out[i] = (float)u;
      }
    }
  }
  return true;
}

}  // namespace h5lite
}  // namespace caffe
