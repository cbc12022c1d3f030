// Minimal glog-style CHECK/LOG for the Caffe-compatible host layer.
// The reference aborts the process on CHECK failure (glog LOG(FATAL), device_alternate.hpp:48-67).
// Here a failed CHECK throws caffe::FatalError: uncaught it terminates the process exactly like glog
// would; the C API wrappers (src/capi.cpp) catch it and hand the text to the foreign caller.
#ifndef MSCNN_CAFFE_UTIL_LOGGING_HPP_
#define MSCNN_CAFFE_UTIL_LOGGING_HPP_

#include <cstdio>
#include <sstream>
#include <stdexcept>
#include <string>

namespace caffe {

class FatalError : public std::runtime_error {
 public:
  explicit FatalError(const std::string& what) : std::runtime_error(what) {}
};

namespace logging {
int& verbosity();   // 0: warnings+errors (default), 1: INFO
class Fatal {
 public:
  Fatal(const char* file, int line, const char* cond) { os_ << file << ":" << line << "] Check failed: " << cond << " "; }
  [[noreturn]] ~Fatal() noexcept(false) { throw FatalError(os_.str()); }
  std::ostream& stream() { return os_; }
 private:
  std::ostringstream os_;
};
class Message {
 public:
  Message(const char* file, int line, int level) : level_(level) { os_ << file << ":" << line << "] "; }
  ~Message() {
    if (level_ > 0 || verbosity() > 0) std::fprintf(stderr, "%c %s\n", "IWE"[level_], os_.str().c_str());
  }
  std::ostream& stream() { return os_; }
 private:
  int level_;
  std::ostringstream os_;
};
struct Voidify { void operator&(std::ostream&) {} };
}  // namespace logging
}  // namespace caffe

#define CHECK(cond) (cond) ? (void)0 : ::caffe::logging::Voidify() & ::caffe::logging::Fatal(__FILE__, __LINE__, #cond).stream()
#define CHECK_OP(a, b, op) CHECK((a)op(b)) << "(" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) CHECK_OP(a, b, <)
#define CHECK_LE(a, b) CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) CHECK_OP(a, b, >)
#define CHECK_GE(a, b) CHECK_OP(a, b, >=)
#define LOG_INFO ::caffe::logging::Message(__FILE__, __LINE__, 0).stream()
#define LOG_WARNING ::caffe::logging::Message(__FILE__, __LINE__, 1).stream()
#define LOG_ERROR ::caffe::logging::Message(__FILE__, __LINE__, 2).stream()
#define LOG_FATAL ::caffe::logging::Fatal(__FILE__, __LINE__, "LOG(FATAL)").stream()
#define LOG(severity) LOG_##severity
#define DLOG(severity) LOG_##severity
#define NOT_IMPLEMENTED LOG(FATAL) << "Not Implemented Yet"

#endif
