// A reader for the subset of the HDF5 file format that Caffe's weight snapshots use (Net::ToHDF5, net.cpp:868-918 ->
// hdf5_save_nd_dataset, util/hdf5.cpp:95-142: H5Gcreate2 + H5LTmake_dataset_float/double): old-style groups (symbol table:
// v1 B-tree + local heap + SNOD leaves), version-1 object headers, contiguous or compact datasets of fixed- or floating-point
// numbers.  That is what every libhdf5 writes by default (libver "earliest"), 1.8 and 1.10 alike.  No libhdf5 dependency:
// the library is absent from the deployment image, and the reference only ever calls H5Gopen2 / H5Lexists /
// H5LTget_dataset_{ndims,info} / H5LTread_dataset_float on these files (net.cpp:806-848, util/hdf5.cpp:9-74).
//
// Everything read from the file is bounds-checked; anything outside the subset (superblock v2+, object header v2, chunked or
// filtered datasets, link-message groups) is reported by name, never guessed at.
#ifndef MSCNN_CAFFE_UTIL_HDF5_LITE_HPP_
#define MSCNN_CAFFE_UTIL_HDF5_LITE_HPP_

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace caffe {
namespace h5lite {

struct Dataset {
  std::vector<long long> dims;
  int type_class = -1;          // 0 fixed-point, 1 floating-point (H5T_INTEGER / H5T_FLOAT)
  int type_size = 0;            // bytes per element
  bool big_endian = false, is_signed = false;
  uint64_t data_offset = 0;     // file offset of the raw data (contiguous or compact)
  uint64_t data_bytes = 0;
  // element count, saturated at kMaxCount + 1 (a crafted dataspace must not wrap the product: ReadDatasetInfo refuses anything above
  // kMaxCount = INT_MAX, the limit of Blob's int shapes)
  static constexpr long long kMaxCount = 2147483647LL;
  long long count() const {
    long long c = 1;
    for (size_t i = 0; i < dims.size(); ++i) {
      if (dims[i] < 0) return kMaxCount + 1;
      if (dims[i] == 0) return 0;
      if (c > kMaxCount / dims[i]) return kMaxCount + 1;
      c *= dims[i];
    }
    return c;
  }
};

class File {
 public:
  // `bytes` must outlive the File.  Throws nothing: every method returns false and sets error().
  explicit File(const std::string& bytes);
  bool ok() const { return ok_; }
  const std::string& error() const { return err_; }

  uint64_t root() const { return root_header_; }
  // links of the group whose object header is at `group_header`: (name, object header address), in the file's B-tree order
  bool ListGroup(uint64_t group_header, std::vector<std::pair<std::string, uint64_t> >* links);
  // H5Lexists + H5Gopen2 / dataset open: address of the object header the link `name` of that group points at
  bool Find(uint64_t group_header, const std::string& name, uint64_t* object_header, bool* found);
  bool ReadDatasetInfo(uint64_t object_header, Dataset* ds);
  // H5LTread_dataset_float: the elements converted to float (fixed- and floating-point sources, either byte order)
  bool ReadFloats(const Dataset& ds, float* out);

 private:
  struct Message { int type; uint64_t offset, size; };
  bool Fail(const std::string& what);
  bool Need(uint64_t off, uint64_t n, const char* what);
  uint64_t U(uint64_t off, int nbytes) const;         // little-endian unsigned, caller has checked the range
  bool Addr(uint64_t off, uint64_t* a, const char* what);
  bool Messages(uint64_t header, std::vector<Message>* out);
  bool SymbolTable(uint64_t group_header, uint64_t* btree, uint64_t* heap);
  bool WalkBtree(uint64_t node, uint64_t heap_data, uint64_t heap_size, int depth,
                 std::vector<std::pair<std::string, uint64_t> >* links);

  const unsigned char* p_;
  uint64_t n_;
  bool ok_;
  std::string err_;
  int so_, sl_;                 // size of offsets / lengths
  uint64_t base_, root_header_, budget_;
};

}  // namespace h5lite
}  // namespace caffe
#endif
