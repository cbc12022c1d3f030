"""A .caffemodel written by the protobuf runtime itself (protobuf-python, the library the reference's pycaffe uses to save
models) -- an encoder this repository did not write, for the reader in mscnn_amd/host/src/net.cpp.

protoc is not available here, so the slice of caffe.proto the reader consumes is declared at run time as a
FileDescriptorProto (same field numbers, labels, types and packed options as src/caffe/proto/caffe.proto:6-21, 65-95,
311-329): BlobShape, BlobProto, LayerParameter{name, type, bottom, top, phase, blobs}, NetParameter{name, layer}."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_classes = None


def _field(msg, name, number, ftype, label=_F.LABEL_OPTIONAL, type_name=None, packed=False):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed:
        f.options.packed = True


def classes():
    global _classes
    if _classes is not None:
        return _classes
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "caffe_subset.proto"
    fd.package = "caffe"
    fd.syntax = "proto2"
    shape = fd.message_type.add(); shape.name = "BlobShape"
    _field(shape, "dim", 1, _F.TYPE_INT64, _F.LABEL_REPEATED, packed=True)
    blob = fd.message_type.add(); blob.name = "BlobProto"
    _field(blob, "shape", 7, _F.TYPE_MESSAGE, type_name=".caffe.BlobShape")
    _field(blob, "data", 5, _F.TYPE_FLOAT, _F.LABEL_REPEATED, packed=True)
    _field(blob, "diff", 6, _F.TYPE_FLOAT, _F.LABEL_REPEATED, packed=True)
    _field(blob, "double_data", 8, _F.TYPE_DOUBLE, _F.LABEL_REPEATED, packed=True)
    _field(blob, "double_diff", 9, _F.TYPE_DOUBLE, _F.LABEL_REPEATED, packed=True)
    for i, nm in enumerate(("num", "channels", "height", "width"), start=1):
        _field(blob, nm, i, _F.TYPE_INT32)
    layer = fd.message_type.add(); layer.name = "LayerParameter"
    _field(layer, "name", 1, _F.TYPE_STRING)
    _field(layer, "type", 2, _F.TYPE_STRING)
    _field(layer, "bottom", 3, _F.TYPE_STRING, _F.LABEL_REPEATED)
    _field(layer, "top", 4, _F.TYPE_STRING, _F.LABEL_REPEATED)
    _field(layer, "phase", 10, _F.TYPE_INT32)                # enum Phase in caffe.proto: same wire type
    _field(layer, "loss_weight", 5, _F.TYPE_FLOAT, _F.LABEL_REPEATED)      # NOT packed in caffe.proto: fixed32 per element
    _field(layer, "blobs", 7, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".caffe.BlobProto")
    v1 = fd.message_type.add(); v1.name = "V1LayerParameter"        # caffe.proto:1358-1361, 1405, 1441 (deprecated, still loadable)
    _field(v1, "bottom", 2, _F.TYPE_STRING, _F.LABEL_REPEATED)
    _field(v1, "top", 3, _F.TYPE_STRING, _F.LABEL_REPEATED)
    _field(v1, "name", 4, _F.TYPE_STRING)
    _field(v1, "type", 5, _F.TYPE_INT32)                     # enum LayerType: same wire type
    _field(v1, "blobs", 6, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".caffe.BlobProto")
    _field(v1, "blobs_lr", 7, _F.TYPE_FLOAT, _F.LABEL_REPEATED)
    net = fd.message_type.add(); net.name = "NetParameter"
    _field(net, "name", 1, _F.TYPE_STRING)
    _field(net, "layers", 2, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".caffe.V1LayerParameter")
    _field(net, "layer", 100, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".caffe.LayerParameter")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    _classes = {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe." + n))
                for n in ("BlobShape", "BlobProto", "LayerParameter", "V1LayerParameter", "NetParameter")}
    return _classes


def serialize(layers, name="from_protobuf_python", v1=False):
    """layers: [(name, type, [(array, mode)])] with mode in {"shape", "legacy", "double"} -> bytes (NetParameter).
    v1: the deprecated `layers = 2` / V1LayerParameter form (the published VGG-16 weights MS-CNN starts from use it)."""
    import numpy as np
    C = classes()
    net = C["NetParameter"](); net.name = name
    for lname, ltype, blobs in layers:
        if v1:
            lp = net.layers.add(); lp.name, lp.type = lname, 4          # CONVOLUTION = 4
            lp.blobs_lr.extend([1.0, 2.0])
        else:
            lp = net.layer.add(); lp.name, lp.type, lp.phase = lname, ltype, 1
            lp.loss_weight.append(1.0)
        lp.bottom.append("x"); lp.top.append(lname)
        for arr, mode in blobs:
            a = np.ascontiguousarray(arr, np.float32)
            bp = lp.blobs.add()
            if mode == "legacy":          # pre-BlobShape models: num/channels/height/width index from the END (blob.cpp:392-406)
                dims = [1] * (4 - a.ndim) + list(a.shape)
                bp.num, bp.channels, bp.height, bp.width = dims
            else:
                bp.shape.dim.extend(a.shape)
            if mode == "double":
                bp.double_data.extend(a.reshape(-1).astype(np.float64).tolist())
            else:
                bp.data.extend(a.reshape(-1).tolist())
    return net.SerializeToString()
