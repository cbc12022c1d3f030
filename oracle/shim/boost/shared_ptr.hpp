// oracle/shim: boost::shared_ptr -> std::shared_ptr (test infrastructure for compiling the reference's own sources)
#pragma once
#include <memory>
namespace boost {
using std::shared_ptr;
using std::dynamic_pointer_cast;
using std::static_pointer_cast;
using std::weak_ptr;
}
