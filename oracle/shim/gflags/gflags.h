#pragma once
#define GFLAGS_GFLAGS_H_
namespace gflags { inline void ParseCommandLineFlags(int*, char***, bool) {} }
namespace google { using gflags::ParseCommandLineFlags; }
