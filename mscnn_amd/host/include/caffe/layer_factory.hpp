// LayerRegistry: string -> creator map keyed by the prototxt `type:` (mirror of
// include/caffe/layer_factory.hpp:58-137).  REGISTER_LAYER_CLASS(Type) registers TypeLayer<float>.
#ifndef MSCNN_CAFFE_LAYER_FACTORY_HPP_
#define MSCNN_CAFFE_LAYER_FACTORY_HPP_

#include <map>
#include <string>
#include <vector>

#include "caffe/common.hpp"
#include "caffe/layer.hpp"

namespace caffe {

template <typename Dtype>
class LayerRegistry {
 public:
  typedef shared_ptr<Layer<Dtype> > (*Creator)(const LayerParameter&);
  typedef std::map<string, Creator> CreatorRegistry;

  static CreatorRegistry& Registry() {
    static CreatorRegistry* g_registry_ = new CreatorRegistry();
    return *g_registry_;
  }
  static void AddCreator(const string& type, Creator creator) {
    CreatorRegistry& registry = Registry();
    CHECK_EQ(registry.count(type), 0u) << "Layer type " << type << " already registered.";
    registry[type] = creator;
  }
  static shared_ptr<Layer<Dtype> > CreateLayer(const LayerParameter& param) {
    const string& type = param.type();
    CreatorRegistry& registry = Registry();
    CHECK_EQ(registry.count(type), 1u) << "Unknown layer type: " << type << " (known types: " << LayerTypeListString() << ")";
    return registry[type](param);
  }
  static vector<string> LayerTypeList() {
    vector<string> layer_types;
    for (typename CreatorRegistry::iterator iter = Registry().begin(); iter != Registry().end(); ++iter) layer_types.push_back(iter->first);
    return layer_types;
  }

 private:
  LayerRegistry() {}
  static string LayerTypeListString() {
    string s;
    for (const string& t : LayerTypeList()) s += (s.empty() ? "" : ", ") + t;
    return s;
  }
};

template <typename Dtype>
class LayerRegisterer {
 public:
  LayerRegisterer(const string& type, shared_ptr<Layer<Dtype> > (*creator)(const LayerParameter&)) {
    LayerRegistry<Dtype>::AddCreator(type, creator);
  }
};

#define REGISTER_LAYER_CREATOR(type, creator) static LayerRegisterer<float> g_creator_f_##type(#type, creator<float>)

#define REGISTER_LAYER_CLASS(type)                                                \
  template <typename Dtype>                                                       \
  shared_ptr<Layer<Dtype> > Creator_##type##Layer(const LayerParameter& param) {  \
    return shared_ptr<Layer<Dtype> >(new type##Layer<Dtype>(param));              \
  }                                                                               \
  REGISTER_LAYER_CREATOR(type, Creator_##type##Layer)

}  // namespace caffe
#endif
