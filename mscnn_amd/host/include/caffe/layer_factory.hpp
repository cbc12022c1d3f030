// The layer factory of the drop-in surface: prototxt `type:` string -> constructor.  What user code touches keeps the reference's
// names (include/caffe/layer_factory.hpp:58-137): LayerRegistry<Dtype>::{Creator, CreatorRegistry, Registry, AddCreator, CreateLayer,
// LayerTypeList}, LayerRegisterer<Dtype>, REGISTER_LAYER_CLASS(Type), REGISTER_LAYER_CREATOR(type, creator) -- a reference-style user
// layer (tests/boundary/*.cpp) registers itself with the same one line.  The implementation is this build's own: one table per Dtype
// in a function-local static (constructed on first use, so registration from static initialisers of any translation unit is safe),
// looked up with find(); only `float` is ever registered (the reference also instantiates `double`, common.hpp:41-44 -- no deploy net
// uses it).
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
    static CreatorRegistry table;
    return table;
  }
  // one creator per type string; a second registration of the same string is a programming error, reported by name
  static void AddCreator(const string& type, Creator creator) {
    const bool fresh = Registry().insert(std::make_pair(type, creator)).second;
    CHECK(fresh) << "Layer type " << type << " already registered.";
  }
  static shared_ptr<Layer<Dtype> > CreateLayer(const LayerParameter& param) {
    const typename CreatorRegistry::const_iterator hit = Registry().find(param.type());
    if (hit == Registry().end()) {
      string known;
      for (const string& t : LayerTypeList()) known += (known.empty() ? "" : ", ") + t;
      LOG(FATAL) << "Unknown layer type: " << param.type() << " (known types: " << known << ")";
    }
    return hit->second(param);
  }
  static vector<string> LayerTypeList() {
    vector<string> types;
    types.reserve(Registry().size());
    for (const auto& entry : Registry()) types.push_back(entry.first);      // (std::map: already in alphabetical order)
    return types;
  }

 private:
  LayerRegistry();      // a namespace of statics: never instantiated
};

// an object whose construction registers: what the macros below place at namespace scope
template <typename Dtype>
class LayerRegisterer {
 public:
  LayerRegisterer(const string& type, typename LayerRegistry<Dtype>::Creator creator) { LayerRegistry<Dtype>::AddCreator(type, creator); }
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
