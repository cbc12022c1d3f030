"""ctypes binding of libmscnn_caffe.so (include/mscnn_net.h): the Caffe-compatible C++ runtime.

Mirrors the slice of matcaffe / pycaffe the reference's drivers use (caffe.Net(prototxt, 'test'), net.forward,
blob get/set; examples/kitti_car/run_mscnn_detection.m:24,72-79).  No CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmscnn_caffe.so")
_lib = None


class NetError(RuntimeError):
    pass


class DetectParams(C.Structure):
    _fields_ = [("cls_id", C.c_int), ("bbox_mean", C.c_float * 4), ("bbox_std", C.c_float * 4), ("proposal_thr", C.c_float),
                ("ratio_h", C.c_double), ("ratio_w", C.c_double), ("org_h", C.c_double), ("org_w", C.c_double),
                ("nms_overlap", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NetError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        C.CDLL(os.path.join(_HERE, "libmscnn_hip.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB_PATH)
        for f in ("mscnn_net_last_error", "mscnn_net_layer_name", "mscnn_net_layer_type", "mscnn_net_layer_bottom",
                  "mscnn_net_layer_top", "mscnn_net_layer_kernel", "mscnn_net_layer_dtype", "mscnn_net_layer_param_text", "mscnn_net_blob_name", "mscnn_net_output_name"):
            getattr(L, f).restype = C.c_char_p
        L.mscnn_net_layer_flops.restype = C.c_double
        L.mscnn_net_layer_executed_flops.restype = C.c_double
        L.mscnn_net_layer_calibration_err.restype = C.c_double
        L.mscnn_net_layer_ms.restype = C.c_float
        L.mscnn_net_blob_device_ptr.restype = C.c_void_p
        L.mscnn_net_destroy.restype = None
        L.mscnn_net_detect_pack_bytes.restype = C.c_size_t
        vp, ci, cs = C.c_void_p, C.c_int, C.c_char_p
        sig = {
            "mscnn_net_create_from_file": [cs, ci, vp], "mscnn_net_create_from_string": [cs, ci, vp], "mscnn_net_destroy": [vp],
            "mscnn_net_create_from_string_ex": [cs, ci, C.c_uint, vp],
            "mscnn_net_layer_executed_flops": [vp, ci], "mscnn_net_set_conv_profiling": [vp, ci], "mscnn_net_layer_stage_ms": [vp, ci, vp],
            "mscnn_net_set_precision": [vp, cs], "mscnn_net_layer_dtype": [vp, ci],
            "mscnn_net_set_conv_algo": [vp, ci, ci], "mscnn_net_set_inner_product_algo": [vp, ci, ci], "mscnn_net_set_conv_tuning": [vp, ci, ci, ci, ci],
            "mscnn_net_calibrate_numerics": [vp, C.c_double, vp], "mscnn_net_set_numerics_watch": [vp, ci, C.c_double],
            "mscnn_net_numerics_watch_state": [vp, vp, vp, ci], "mscnn_net_layer_calibration_err": [vp, ci],
            "mscnn_net_set_auto_calibrate": [vp, C.c_double], "mscnn_net_set_chain_fusion": [vp, ci], "mscnn_net_chain_pairs": [vp, vp, vp, ci], "mscnn_net_auto_calibrate_state": [vp, vp, vp, ci],
            "mscnn_net_load_caffemodel": [vp, cs], "mscnn_net_set_stream": [vp], "mscnn_net_num_layers": [vp],
            "mscnn_net_layer_name": [vp, ci], "mscnn_net_layer_type": [vp, ci], "mscnn_net_layer_index": [vp, cs],
            "mscnn_net_layer_num_bottoms": [vp, ci], "mscnn_net_layer_num_tops": [vp, ci], "mscnn_net_layer_bottom": [vp, ci, ci],
            "mscnn_net_layer_top": [vp, ci, ci], "mscnn_net_layer_num_params": [vp, ci], "mscnn_net_layer_param_shape": [vp, ci, ci, vp, vp],
            "mscnn_net_layer_fused_away": [vp, ci], "mscnn_net_layer_param_text": [vp, ci], "mscnn_net_layer_kernel": [vp, ci], "mscnn_net_layer_flops": [vp, ci],
            "mscnn_net_num_blobs": [vp], "mscnn_net_blob_name": [vp, ci], "mscnn_net_blob_shape": [vp, cs, vp, vp],
            "mscnn_net_num_inputs": [vp], "mscnn_net_num_outputs": [vp], "mscnn_net_output_name": [vp, ci],
            "mscnn_net_set_param": [vp, ci, ci, vp, C.c_size_t], "mscnn_net_get_param": [vp, ci, ci, vp, C.c_size_t],
            "mscnn_net_set_blob": [vp, cs, vp, C.c_size_t], "mscnn_net_set_blob_device": [vp, cs, vp, C.c_size_t],
            "mscnn_net_set_image": [vp, cs, vp, ci, ci, ci, vp],
            "mscnn_net_get_blob": [vp, cs, vp, C.c_size_t, vp], "mscnn_net_blob_device_ptr": [vp, cs],
            "mscnn_net_forward": [vp], "mscnn_net_forward_from_to": [vp, ci, ci], "mscnn_net_reshape": [vp],
            "mscnn_net_set_layer_timing": [vp, ci], "mscnn_net_layer_ms": [vp, ci],
            "mscnn_net_detect": [vp, vp, vp, vp, ci, vp, vp],
            "mscnn_net_detect_cascade": [vp, vp, C.c_float, cs, cs, cs, vp, vp, ci, vp, vp],
            "mscnn_net_detect_pack_bytes": [ci], "mscnn_net_detect_device": [vp, vp, ci, vp],
            "mscnn_net_unpack_detections": [vp, ci, vp, vp, vp, vp],
            "mscnn_net_handoff_state": [vp, vp],
            "mscnn_net_detect_begin": [vp, vp, ci], "mscnn_net_detect_end": [vp, vp, vp, ci, vp, vp],
            "mscnn_net_reshape_input": [vp, cs, vp, ci], "mscnn_net_detect_image": [vp, vp, ci, vp, vp, ci, vp, vp],
        }
        for name, args in sig.items():
            getattr(L, name).argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise NetError(lib().mscnn_net_last_error().decode())


class Net:
    """caffe.Net(prototxt, 'test') on one MI355X."""

    def __init__(self, prototxt_path=None, prototxt_text=None, device=0, fusion=None):
        """fusion: None = the process default (on unless MSCNN_NO_FUSE=1), True / False = explicit."""
        self._h = C.c_void_p()
        if prototxt_text is None and fusion is not None:
            prototxt_text = open(prototxt_path).read()
        if prototxt_text is not None and fusion is not None:
            _check(lib().mscnn_net_create_from_string_ex(prototxt_text.encode(), device, 0 if fusion else 1, C.byref(self._h)))
        elif prototxt_text is not None:
            _check(lib().mscnn_net_create_from_string(prototxt_text.encode(), device, C.byref(self._h)))
        else:
            _check(lib().mscnn_net_create_from_file(str(prototxt_path).encode(), device, C.byref(self._h)))
        L = lib()
        n = L.mscnn_net_num_layers(self._h)
        self.layer_names = [L.mscnn_net_layer_name(self._h, i).decode() for i in range(n)]
        self.layer_types = [L.mscnn_net_layer_type(self._h, i).decode() for i in range(n)]
        self.blob_names = [L.mscnn_net_blob_name(self._h, i).decode() for i in range(L.mscnn_net_num_blobs(self._h))]
        self.outputs = [L.mscnn_net_output_name(self._h, i).decode() for i in range(L.mscnn_net_num_outputs(self._h))]

    def __del__(self):
        try:
            if self._h:
                lib().mscnn_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- graph ----
    def layer_bottoms(self, i):
        return [lib().mscnn_net_layer_bottom(self._h, i, k).decode() for k in range(lib().mscnn_net_layer_num_bottoms(self._h, i))]

    def layer_tops(self, i):
        return [lib().mscnn_net_layer_top(self._h, i, k).decode() for k in range(lib().mscnn_net_layer_num_tops(self._h, i))]

    def param_shapes(self, i):
        out = []
        for p in range(lib().mscnn_net_layer_num_params(self._h, i)):
            dims = (C.c_int * 8)(); nd = C.c_int()
            _check(lib().mscnn_net_layer_param_shape(self._h, i, p, dims, C.byref(nd)))
            out.append(tuple(dims[:nd.value]))
        return out

    def blob_shape(self, name):
        dims = (C.c_int * 8)(); nd = C.c_int()
        _check(lib().mscnn_net_blob_shape(self._h, name.encode(), dims, C.byref(nd)))
        return tuple(dims[:nd.value])

    def layer_param_text(self, i):
        return lib().mscnn_net_layer_param_text(self._h, i).decode()

    def fused_away(self, i):
        return bool(lib().mscnn_net_layer_fused_away(self._h, i))

    def layer_kernel(self, i):
        return lib().mscnn_net_layer_kernel(self._h, i).decode()

    def layer_flops(self, i):
        return lib().mscnn_net_layer_flops(self._h, i)

    def layer_executed_flops(self, i):
        return lib().mscnn_net_layer_executed_flops(self._h, i)

    def set_conv_profiling(self, on):
        _check(lib().mscnn_net_set_conv_profiling(self._h, int(on)))

    def layer_stage_ms(self, i):
        """(input transform, MFMA GEMM kernels, output transform) ms of layer i's last forward; zeros for non-conv layers."""
        out = (C.c_float * 3)()
        lib().mscnn_net_layer_stage_ms(self._h, i, out)
        return tuple(out)

    def set_precision(self, dtype):
        """"f32" (default, reference parity), "f16" (fp16 MFMA operands, fp32 accumulate: BASELINE config 5) or "f16x3" (the
        Winograd plane GEMMs on the fp16 pipe with exactly split operands: fp32-grade, same parity gates as "f32")."""
        _check(lib().mscnn_net_set_precision(self._h, dtype.encode()))

    def layer_dtype(self, i):
        return lib().mscnn_net_layer_dtype(self._h, i).decode()

    def set_conv_algo(self, layer, algo):
        i = layer if isinstance(layer, int) else self.layer_names.index(layer)
        _check(lib().mscnn_net_set_conv_algo(self._h, i, algo))

    def set_inner_product_algo(self, layer, algo):
        """0 = auto (fc6-class shapes on the plane-GEMM kernel), 1 = always gemm.hip's stream-K kernel; layer -1 = all."""
        i = layer if isinstance(layer, int) else self.layer_names.index(layer)
        _check(lib().mscnn_net_set_inner_product_algo(self._h, i, algo))

    def set_conv_tuning(self, layer, variant=0, grid=0, flags=0):
        i = layer if isinstance(layer, int) else self.layer_names.index(layer)
        _check(lib().mscnn_net_set_conv_tuning(self._h, i, variant, grid, flags))

    def calibrate_numerics(self, tol=5e-5):
        """After a forward on representative input: Winograd layers that stray more than tol from the direct kernel fall
        back to it.  Returns ({layer name: measured error}, [names switched])."""
        n = C.c_int()
        before = [self.layer_kernel(i) for i in range(len(self.layer_names))]
        _check(lib().mscnn_net_calibrate_numerics(self._h, tol, C.byref(n)))
        errs = {self.layer_names[i]: lib().mscnn_net_layer_calibration_err(self._h, i)
                for i in range(len(self.layer_names)) if before[i].startswith("winograd")}
        switched = [nm for nm, e in errs.items() if not e <= tol]
        assert len(switched) == n.value
        return errs, switched

    def set_auto_calibrate(self, tol=5e-5):
        """The first-forward check of every Winograd layer against the direct kernel is ON by default (5e-5); tol = 0 opts out,
        tol > 0 re-arms it with that tolerance."""
        _check(lib().mscnn_net_set_auto_calibrate(self._h, tol))

    def set_chain_fusion(self, on=True):
        """Chains of same-resolution F(4x4,3x3) convolutions keep the blob between two members out of HBM (on by default); off = every
        blob is written by every forward.  Reading such a blob re-runs its producer on demand -- same bytes either way."""
        _check(lib().mscnn_net_set_chain_fusion(self._h, int(on)))

    def chain_pairs(self):
        """[(producer layer name, consumer layer name | None)]: convolutions whose top may stay unwritten while a forward runs
        (consumer None: the top is read by its fused 2x2 pooling only)."""
        a, b = (C.c_int * 64)(), (C.c_int * 64)()
        k = lib().mscnn_net_chain_pairs(self._h, a, b, 64)
        return [(self.layer_names[a[i]], self.layer_names[b[i]] if b[i] >= 0 else None) for i in range(min(k, 64))]

    def auto_calibrate_state(self):
        """(first-forward checks done so far, [names of the layers they sent to the direct kernel], {layer: last measured error})."""
        n, sw = C.c_int(), (C.c_int * 64)()
        k = lib().mscnn_net_auto_calibrate_state(self._h, C.byref(n), sw, 64)
        errs = {nm: lib().mscnn_net_layer_calibration_err(self._h, i) for i, nm in enumerate(self.layer_names)}
        return n.value, [self.layer_names[sw[i]] for i in range(min(k, 64))], {nm: e for nm, e in errs.items() if e > 0}

    def set_numerics_watch(self, period, tol=5e-5):
        """Every period-th forward re-checks one Winograd layer (round robin) against the direct kernel on the live frame."""
        _check(lib().mscnn_net_set_numerics_watch(self._h, period, tol))

    def numerics_watch_state(self):
        """(layer checks done, [names of the layers the watch sent to the direct kernel])."""
        n, sw = C.c_int(), (C.c_int * 64)()
        k = lib().mscnn_net_numerics_watch_state(self._h, C.byref(n), sw, 64)
        return n.value, [self.layer_names[sw[i]] for i in range(min(k, 64))]

    # ---- weights ----
    def set_param(self, layer, p, arr):
        i = layer if isinstance(layer, int) else self.layer_names.index(layer)
        a = np.ascontiguousarray(arr, np.float32)
        _check(lib().mscnn_net_set_param(self._h, i, p, a.ctypes.data_as(C.c_void_p), a.size))

    def get_param(self, layer, p):
        i = layer if isinstance(layer, int) else self.layer_names.index(layer)
        shape = self.param_shapes(i)[p]
        a = np.empty(shape, np.float32)
        _check(lib().mscnn_net_get_param(self._h, i, p, a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def load_caffemodel(self, path):
        _check(lib().mscnn_net_load_caffemodel(self._h, str(path).encode()))

    # ---- data ----
    def set_blob(self, name, arr):
        if hasattr(arr, "is_cuda"):   # torch CUDA tensor: device -> device
            assert arr.is_cuda and arr.is_contiguous()
            _check(lib().mscnn_net_set_blob_device(self._h, name.encode(), C.c_void_p(arr.data_ptr()), arr.numel()))
            return
        a = np.ascontiguousarray(arr, np.float32)
        _check(lib().mscnn_net_set_blob(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def set_image(self, name, img_rgb_u8, mean_bgr=None):
        """uint8 [H, W, 3] RGB frame (numpy array or torch CUDA tensor) -> resized / BGR / mean-subtracted input blob."""
        m = (C.c_float * 3)(*mean_bgr) if mean_bgr is not None else None
        if hasattr(img_rgb_u8, "is_cuda"):
            assert img_rgb_u8.is_cuda and img_rgb_u8.is_contiguous() and str(img_rgb_u8.dtype) == "torch.uint8"
            h, w, _ = img_rgb_u8.shape
            _check(lib().mscnn_net_set_image(self._h, name.encode(), C.c_void_p(img_rgb_u8.data_ptr()), 1, h, w, m))
            return
        a = np.ascontiguousarray(img_rgb_u8, np.uint8)
        h, w, _ = a.shape
        _check(lib().mscnn_net_set_image(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), 0, h, w, m))

    def get_blob(self, name):
        a = np.empty(self.blob_shape(name), np.float32)
        cnt = C.c_size_t()
        _check(lib().mscnn_net_get_blob(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size, C.byref(cnt)))
        return a

    def forward(self, start=0, end=-1):
        if start == 0 and end == -1:
            _check(lib().mscnn_net_forward(self._h))
        else:
            _check(lib().mscnn_net_forward_from_to(self._h, start, end))

    def reshape(self):
        _check(lib().mscnn_net_reshape(self._h))

    def reshape_input(self, name, shape):
        """net.blobs[name].reshape(*shape); net.reshape() -- another batch / frame size on the same net (Caffe order N, C, H, W)."""
        dims = (C.c_int * len(shape))(*[int(v) for v in shape])
        _check(lib().mscnn_net_reshape_input(self._h, name.encode(), dims, len(shape)))

    def handoff_state(self):
        """(stream-K hand-off time-outs this net has answered by re-running the frame, whole-tile scheduling forced for the process)."""
        forced = C.c_int()
        n = lib().mscnn_net_handoff_state(self._h, C.byref(forced))
        return n, bool(forced.value)

    def set_layer_timing(self, on):
        lib().mscnn_net_set_layer_timing(self._h, int(on))

    def layer_ms(self):
        return [lib().mscnn_net_layer_ms(self._h, i) for i in range(len(self.layer_names))]

    @staticmethod
    def _params(cls_id, ratios, org_hw, bbox_mean, bbox_std, proposal_thr, nms_overlap):
        p = DetectParams()
        p.cls_id = cls_id
        for k in range(4):
            p.bbox_mean[k] = bbox_mean[k]; p.bbox_std[k] = bbox_std[k]
        p.proposal_thr = proposal_thr
        p.ratio_h, p.ratio_w = ratios
        p.org_h, p.org_w = org_hw
        p.nms_overlap = nms_overlap
        return p

    def detect_device(self, cap, cls_id, ratios, org_hw, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2),
                      proposal_thr=-10.0, nms_overlap=0.5):
        """Final stage into the fixed-capacity DEVICE pack (multi-GPU gather input, include/mscnn_dist.h); asynchronous.
        Returns the device address of the pack (detect_pack_bytes(cap) bytes)."""
        p = self._params(cls_id, ratios, org_hw, bbox_mean, bbox_std, proposal_thr, nms_overlap)
        ptr = C.c_void_p()
        _check(lib().mscnn_net_detect_device(self._h, C.byref(p), cap, C.byref(ptr)))
        return ptr.value

    def detect_begin(self, cap, cls_id, ratios, org_hw, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2), proposal_thr=-10.0,
                     nms_overlap=0.5):
        """mscnn_net_detect_begin: final stage of the forward just done + an asynchronous copy of its pack to pinned host memory; go
        on with the next frame and collect with detect_end (at most two frames in flight)."""
        p = self._params(cls_id, ratios, org_hw, bbox_mean, bbox_std, proposal_thr, nms_overlap)
        _check(lib().mscnn_net_detect_begin(self._h, C.byref(p), cap))

    def detect_end(self, cap):
        """mscnn_net_detect_end: (dets [D, 5] float64, ids [D] int32, R) of the oldest frame in flight."""
        rows = max(cap, 1)
        if getattr(self, "_end_rows", 0) != rows:
            self._end_dets = np.zeros((rows, 5), np.float64); self._end_ids = np.zeros(rows, np.int32); self._end_rows = rows
        D = C.c_int(); R = C.c_int()
        _check(lib().mscnn_net_detect_end(self._h, self._end_dets.ctypes.data_as(C.c_void_p), self._end_ids.ctypes.data_as(C.c_void_p), cap,
                                          C.byref(D), C.byref(R)))
        return self._end_dets[:D.value].copy(), self._end_ids[:D.value].copy(), R.value

    def detect_cascade(self, bbox_blob, prob_blob, proposal_blob, cls_id, ratios, org_hw, det_thr=0.0, nms_overlap=0.5, cap=4096):
        """Final stage of the cascade drivers (run_cascademscnn.m:84-127) for one cascade output; returns (dets, ids, R)."""
        p = self._params(cls_id, ratios, org_hw, (0, 0, 0, 0), (1, 1, 1, 1), 0.0, nms_overlap)
        dets = np.zeros((cap, 5), np.float64); ids = np.zeros(cap, np.int32)
        D = C.c_int(); R = C.c_int()
        _check(lib().mscnn_net_detect_cascade(self._h, C.byref(p), det_thr, bbox_blob.encode(), prob_blob.encode(), proposal_blob.encode(),
                                              dets.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), cap, C.byref(D), C.byref(R)))
        return dets[:D.value].copy(), ids[:D.value].copy(), R.value

    def detect_image(self, image, cls_id, ratios, org_hw, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2), proposal_thr=-10.0,
                     nms_overlap=0.5, cap=4096):
        """The final stage for image `image` of a batched forward; returns (dets[D,5], rows of the net's ROI blobs[D], the image's R)."""
        p = self._params(cls_id, ratios, org_hw, bbox_mean, bbox_std, proposal_thr, nms_overlap)
        dets = np.zeros((cap, 5), np.float64); ids = np.zeros(cap, np.int32)
        D = C.c_int(); R = C.c_int()
        _check(lib().mscnn_net_detect_image(self._h, C.byref(p), image, dets.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), cap,
                                            C.byref(D), C.byref(R)))
        return dets[:D.value].copy(), ids[:D.value].copy(), R.value

    def detect(self, cls_id, ratios, org_hw, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2), proposal_thr=-10.0,
               nms_overlap=0.5, cap=4096):
        """Final detection stage on the device; returns (dets[D,5] float64 [x y w h prob], roi ids[D], R)."""
        p = DetectParams()
        p.cls_id = cls_id
        for k in range(4):
            p.bbox_mean[k] = bbox_mean[k]; p.bbox_std[k] = bbox_std[k]
        p.proposal_thr = proposal_thr
        p.ratio_h, p.ratio_w = ratios
        p.org_h, p.org_w = org_hw
        p.nms_overlap = nms_overlap
        dets = np.zeros((cap, 5), np.float64); ids = np.zeros(cap, np.int32)
        D = C.c_int(); R = C.c_int()
        _check(lib().mscnn_net_detect(self._h, C.byref(p), dets.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), cap,
                                      C.byref(D), C.byref(R)))
        return dets[:D.value].copy(), ids[:D.value].copy(), R.value


def detect_pack_bytes(cap):
    return lib().mscnn_net_detect_pack_bytes(cap)


def unpack_detections(pack_host, cap):
    """One host copy of a detection pack (bytes / ctypes buffer / numpy uint8) -> (dets[D,5] float64, ids[D] int32, R)."""
    buf = np.frombuffer(pack_host, np.uint8) if not isinstance(pack_host, np.ndarray) else pack_host
    assert buf.nbytes >= detect_pack_bytes(cap)
    buf = np.ascontiguousarray(buf)
    dets = np.zeros((max(cap, 1), 5), np.float64); ids = np.zeros(max(cap, 1), np.int32)
    D = C.c_int(); R = C.c_int()
    _check(lib().mscnn_net_unpack_detections(buf.ctypes.data_as(C.c_void_p), cap, dets.ctypes.data_as(C.c_void_p),
                                             ids.ctypes.data_as(C.c_void_p), C.byref(D), C.byref(R)))
    return dets[:D.value].copy(), ids[:D.value].copy(), R.value
