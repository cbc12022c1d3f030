// Umbrella header, same path as the reference's include/caffe/caffe.hpp.
#ifndef MSCNN_CAFFE_CAFFE_HPP_
#define MSCNN_CAFFE_CAFFE_HPP_
#include "caffe/blob.hpp"
#include "caffe/common.hpp"
#include "caffe/layer.hpp"
#include "caffe/layer_factory.hpp"
#include "caffe/layers/mscnn_layers.hpp"
#include "caffe/net.hpp"
#include "caffe/proto/caffe_param.hpp"
#endif
