# oracle/_ref: the reference's OWN hot-path sources, compiled where they lie under $(REF) against oracle/shim
# (glog / gflags / boost / protobuf / cblas stand-ins), plus oracle/ref_harness.cpp (C entry points for the tests).
# Nothing from $(REF) is copied into this repository; outputs go to oracle/_ref/ only (git-ignored).
REF_SRCS := blob.cpp syncedmem.cpp layer.cpp common.cpp util/math_functions.cpp util/im2col.cpp \
            layers/box_output_layer.cpp layers/roi_pooling_layer.cpp layers/roi_align_layer.cpp layers/eltwise_layer.cpp layers/decode_bbox_layer.cpp \
            layers/conv_layer.cpp layers/base_conv_layer.cpp layers/deconv_layer.cpp layers/pooling_layer.cpp \
            layers/relu_layer.cpp layers/neuron_layer.cpp layers/inner_product_layer.cpp layers/concat_layer.cpp \
            layers/softmax_layer.cpp layers/split_layer.cpp
REF_OBJS := $(addprefix _ref/obj/,$(subst /,_,$(REF_SRCS:.cpp=.o)))
MKL_DIR  ?= /opt/conda/lib
REF_CXXFLAGS := -O2 -ffp-contract=off -fPIC -std=c++11 -DCPU_ONLY -Ishim -I$(REF)/include -w

ref: _ref/libmscnn_ref.so _ref/kitti_eval_ref

# the devkit evaluator the reference ships (one self-contained file): the pin of mscnn_amd/host/tools/kitti_eval.cpp
_ref/kitti_eval_ref: $(REF)/examples/kitti_result/eval/evaluate_object.cpp
	@mkdir -p _ref
	$(CXX) -O2 -w -o $@ $<

SHIM_HDRS := $(shell find shim -name '*.h' -o -name '*.hpp')

define REF_RULE
_ref/obj/$(subst /,_,$(1:.cpp=.o)): $(REF)/src/caffe/$(1) $(SHIM_HDRS)
	@mkdir -p _ref/obj
	$(CXX) $(REF_CXXFLAGS) -c $$< -o $$@
endef
$(foreach s,$(REF_SRCS),$(eval $(call REF_RULE,$(s))))

_ref/obj/harness.o: ref_harness.cpp $(SHIM_HDRS)
	@mkdir -p _ref/obj
	$(CXX) $(REF_CXXFLAGS) -c $< -o $@

_ref/libmscnn_ref.so: $(REF_OBJS) _ref/obj/harness.o
	$(CXX) -shared -fPIC -o $@ $^ -L$(MKL_DIR) -l:libmkl_rt.so.1 -Wl,-rpath,$(MKL_DIR) -lpthread
