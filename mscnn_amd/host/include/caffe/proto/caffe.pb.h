// The reference includes "caffe/proto/caffe.pb.h" (protoc output); here it is hand-written.
#include "caffe/proto/caffe_param.hpp"
