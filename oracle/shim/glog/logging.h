// oracle/shim: glog's CHECK/LOG surface.  LOG(FATAL)/failed CHECK throw (the harness reports them) instead of abort().
#pragma once
// the real glog / boost headers pull these in transitively; the reference sources rely on that
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
namespace google {
inline void InitGoogleLogging(const char*) {}
inline void InstallFailureSignalHandler() {}
struct FatalThrow {
  std::ostringstream os;
  FatalThrow(const char* f, int l, const char* c) { os << f << ":" << l << "] " << c << " "; }
  std::ostream& stream() { return os; }
  [[noreturn]] ~FatalThrow() noexcept(false) { throw std::runtime_error(os.str()); }
};
struct NullStream { NullStream& self() { return *this; } template <class T> NullStream& operator<<(const T&) { return *this; } NullStream& operator<<(std::ostream& (*)(std::ostream&)) { return *this; } };
struct Voidify { void operator&(std::ostream&) {} void operator&(NullStream&) {} };
}
#define SHIM_FATAL(cond) ::google::FatalThrow(__FILE__, __LINE__, cond).stream()
#define CHECK(c) (c) ? (void)0 : ::google::Voidify() & SHIM_FATAL("Check failed: " #c)
#define CHECK_OP(a, b, op) ((a)op(b)) ? (void)0 : ::google::Voidify() & SHIM_FATAL("Check failed: " #a " " #op " " #b) << "(" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) CHECK_OP(a, b, <)
#define CHECK_LE(a, b) CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) CHECK_OP(a, b, >)
#define CHECK_GE(a, b) CHECK_OP(a, b, >=)
#define CHECK_NOTNULL(p) (p)
#define DCHECK(c) CHECK(c)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define LOG_INFO ::google::NullStream().self()
#define LOG_WARNING ::google::NullStream().self()
#define LOG_ERROR std::cerr
#define LOG_FATAL SHIM_FATAL("LOG(FATAL)")
#define LOG(sev) LOG_##sev
#define DLOG(sev) LOG_##sev
#define VLOG(n) ::google::NullStream().self()
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::google::Voidify() & LOG_##sev
#define LOG_EVERY_N(sev, n) LOG_##sev
#define LOG_FIRST_N(sev, n) LOG_##sev
