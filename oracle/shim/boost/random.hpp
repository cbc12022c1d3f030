// oracle/shim: boost.random names used by caffe/util/rng.hpp and math_functions.cpp (training-only paths)
#pragma once
#include <random>
namespace boost {
typedef std::mt19937 mt19937;
template <typename T = double> using uniform_real = std::uniform_real_distribution<T>;
template <typename T = int> using uniform_int = std::uniform_int_distribution<T>;
template <typename T = double> using normal_distribution = std::normal_distribution<T>;
template <typename T = double> using bernoulli_distribution = std::bernoulli_distribution;
template <class Engine, class Dist>
class variate_generator {
 public:
  variate_generator(Engine e, Dist d) : e_(e), d_(d) {}
  typename Dist::result_type operator()() { return d_(*e_); }
 private:
  Engine e_;
  Dist d_;
};
namespace random { using boost::mt19937; }
}
