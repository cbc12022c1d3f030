// oracle/shim: the slice of boost.thread the reference's layer.cpp / common.cpp use
#pragma once
#include <memory>
#include <mutex>
#include <thread>
namespace boost {
class mutex : public std::mutex {};
template <typename T>
class thread_specific_ptr {
 public:
  T* get() const { return slot().get(); }
  void reset(T* p) { slot().reset(p); }
  T* operator->() const { return get(); }
 private:
  static std::unique_ptr<T>& slot() { static thread_local std::unique_ptr<T> s; return s; }
};
class thread { public: typedef std::thread::id id; };
}
