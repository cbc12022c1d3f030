"""ctypes binding of libmscnn_hip.so (include/mscnn_hip.h) for Python callers.

Tensors are torch CUDA (ROCm) tensors; torch is used only as the device allocator and stream
owner -- every computation is done by the hand-written HIP kernels behind the C ABI.
There is no CPU fallback: a missing library or a host tensor raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmscnn_hip.so")
_lib = None


class MscnnError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "Cin", "H", "W", "Cout", "Kh", "Kw", "pad_h", "pad_w",
                                       "stride_h", "stride_w", "group", "relu", "algo", "tune_variant", "tune_grid", "tune_flags")]


# mscnn_conv_algo
AMAX_SLOTS = 1024
ALGO_AUTO, ALGO_DIRECT, ALGO_WINO_F2, ALGO_WINO_F3, ALGO_F16, ALGO_WINO_F3_X3, ALGO_WINO_F4 = 0, 1, 2, 3, 4, 5, 6


MAX_HEADS = 16


class BoxOutputDesc(C.Structure):
    _fields_ = [("num_heads", C.c_int), ("num", C.c_int), ("channels", C.c_int),
                ("head_h", C.c_int * MAX_HEADS), ("head_w", C.c_int * MAX_HEADS),
                ("field_w", C.c_float * MAX_HEADS), ("field_h", C.c_float * MAX_HEADS),
                ("downsample_rate", C.c_float * MAX_HEADS),
                ("fg_thr", C.c_float), ("iou_thr", C.c_float), ("nms_mode", C.c_int),
                ("field_whr", C.c_float), ("field_xyr", C.c_float),
                ("max_nms_num", C.c_int), ("max_post_nms_num", C.c_int), ("min_size", C.c_float),
                ("do_bbox_norm", C.c_int), ("bbox_mean", C.c_float * 4), ("bbox_std", C.c_float * 4)]


class DetectionsDesc(C.Structure):
    _fields_ = [("ncls", C.c_int), ("cls_id", C.c_int), ("bbox_mean", C.c_float * 4), ("bbox_std", C.c_float * 4),
                ("proposal_thr", C.c_float), ("ratio_h", C.c_double), ("ratio_w", C.c_double),
                ("org_h", C.c_double), ("org_w", C.c_double), ("nms_overlap", C.c_double)]


NMS_MODES = {"IOU": 0, "IOMU": 1, "IOFU": 2}


def lib():
    """Load libmscnn_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MscnnError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.mscnn_last_error.restype = C.c_char_p
        L.mscnn_version.restype = C.c_char_p
        L.mscnn_conv2d_plan_kernel.restype = C.c_char_p
        L.mscnn_conv2d_plan_kernel.argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_publishes_amax.argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_set_amax_io.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mscnn_conv2d_plan_dtype.restype = C.c_char_p
        L.mscnn_conv2d_plan_dtype.argtypes = [C.c_void_p]
        L.mscnn_inner_product_f16_supported.argtypes = [C.c_int, C.c_int]
        L.mscnn_inner_product_x3_supported.argtypes = [C.c_int, C.c_int]
        L.mscnn_inner_product_x3_packed_bytes.restype = C.c_size_t
        L.mscnn_inner_product_x3_packed_bytes.argtypes = [C.c_int, C.c_int]
        L.mscnn_inner_product_x3_workspace_bytes.restype = C.c_size_t
        L.mscnn_inner_product_x3_workspace_bytes.argtypes = [C.c_int] * 3
        L.mscnn_inner_product_x3_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mscnn_inner_product_x3_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mscnn_inner_product_wg_supported.argtypes = [C.c_int] * 3
        L.mscnn_inner_product_wg_packed_bytes.restype = C.c_size_t
        L.mscnn_inner_product_wg_packed_bytes.argtypes = [C.c_int, C.c_int]
        L.mscnn_inner_product_wg_workspace_bytes.restype = C.c_size_t
        L.mscnn_inner_product_wg_workspace_bytes.argtypes = [C.c_int] * 3
        L.mscnn_inner_product_wg_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mscnn_inner_product_wg_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]
        L.mscnn_max_rel_diff_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]
        L.mscnn_sum_squares_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.mscnn_max_rel_diff_strided_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_double,
                                                     C.c_void_p, C.c_void_p]
        L.mscnn_store_words_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mscnn_wgemm_handoff_event.restype = C.c_ulonglong
        L.mscnn_wgemm_handoff_event.argtypes = []
        L.mscnn_wgemm_force_whole_tiles.restype = None
        L.mscnn_wgemm_force_whole_tiles.argtypes = [C.c_int]
        L.mscnn_wgemm_whole_tiles_forced.argtypes = []
        L.mscnn_debug_inner_product_rows.restype = None
        L.mscnn_debug_inner_product_rows.argtypes = [C.c_int]
        L.mscnn_debug_wgemm_handoff_fault.restype = None
        L.mscnn_debug_wgemm_handoff_fault.argtypes = [C.c_int, C.c_uint]
        L.mscnn_inner_product_pack_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mscnn_inner_product_fwd_f16.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        for f in ("mscnn_conv2d_plan_flops", "mscnn_conv2d_plan_executed_flops"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.mscnn_conv2d_plan_stage_ms.argtypes = [C.c_void_p, C.c_void_p]
        for f in ("mscnn_conv2d_packed_weight_bytes", "mscnn_conv2d_workspace_bytes"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_destroy.argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_destroy.restype = None
        L.mscnn_conv2d_plan_set_batch.argtypes = [C.c_void_p, C.c_int]
        L.mscnn_conv2d_pack_weights.argtypes = [C.c_void_p] * 4
        L.mscnn_conv2d_fwd_f32.argtypes = [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p]
        L.mscnn_conv2d_fwd_pool_f32.argtypes = [C.c_void_p] * 8 + [C.c_size_t, C.c_void_p]
        L.mscnn_conv2d_plan_can_pool.argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_can_fuse_roipool.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mscnn_conv2d_plan_can_chain.argtypes = [C.c_void_p, C.c_void_p]
        L.mscnn_conv2d_plan_can_pool_only.argtypes = [C.c_void_p]
        L.mscnn_conv2d_fwd_chain_f32.argtypes = [C.c_void_p] * 8 + [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mscnn_conv2d_roipool_workspace_bytes.restype = C.c_size_t
        L.mscnn_conv2d_roipool_workspace_bytes.argtypes = [C.c_void_p] + [C.c_int] * 4
        L.mscnn_conv2d_fwd_roipool_pair_f32.argtypes = ([C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_float, C.c_float, C.c_float]
                                                        + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p])
        L.mscnn_roipool_maps_bytes.restype = C.c_size_t
        L.mscnn_roipool_maps_bytes.argtypes = [C.c_int] * 4
        L.mscnn_roipool_maps_build_f32.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        L.mscnn_conv2d_plan_weight_layout.argtypes = [C.c_void_p]
        L.mscnn_conv2d_plan_weight_layout.restype = C.c_ulonglong
        L.mscnn_relu_fwd_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
        L.mscnn_pool2d_fwd_f32.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 11 + [C.c_void_p]
        L.mscnn_inner_product_fwd_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        L.mscnn_concat_channels_f32.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
        L.mscnn_deconv_depthwise_fwd_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 10 + [C.c_void_p]
        L.mscnn_softmax_fwd_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.mscnn_roipool_fwd_f32.argtypes = ([C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_float, C.c_float] + [C.c_int] * 2
                                            + [C.c_void_p])
        L.mscnn_roipool_pair_fwd_f32.argtypes = ([C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int]
                                                 + [C.c_void_p])
        L.mscnn_roialign_fwd_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_float, C.c_float, C.c_void_p]
        L.mscnn_eltwise_fwd_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.mscnn_boxoutput_workspace_bytes.restype = C.c_size_t
        L.mscnn_boxoutput_workspace_bytes.argtypes = [C.c_void_p]
        L.mscnn_boxoutput_max_rows.argtypes = [C.c_void_p]
        L.mscnn_boxoutput_fwd_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mscnn_nms_workspace_bytes.restype = C.c_size_t
        L.mscnn_nms_workspace_bytes.argtypes = [C.c_int]
        L.mscnn_nms_greedy_f32.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]
        L.mscnn_decodebbox_fwd_f32.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mscnn_detections_workspace_bytes.restype = C.c_size_t
        L.mscnn_detections_workspace_bytes.argtypes = [C.c_int]
        L.mscnn_detections_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MscnnError(f"mscnn status {rc}: {lib().mscnn_last_error().decode()}")


def _dev(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise MscnnError("libmscnn_hip has no CPU path: tensor must live on the GPU")
    if not t.is_contiguous():
        raise MscnnError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ConvPlan:
    """Owns a mscnn_conv_plan plus its packed weights and stream-K workspace (device memory)."""

    def __init__(self, N, Cin, H, W, Cout, Kh, Kw, pad=(0, 0), stride=(1, 1), group=1, relu=False, device="cuda",
                 algo=ALGO_AUTO, tune_variant=0, tune_grid=0, tune_flags=0):
        self.desc = ConvDesc(N, Cin, H, W, Cout, Kh, Kw, pad[0], pad[1], stride[0], stride[1], group, int(relu),
                             algo, tune_variant, tune_grid, tune_flags)
        self._p = C.c_void_p()
        _check(lib().mscnn_conv2d_plan_create(C.byref(self.desc), C.byref(self._p)))
        self.device = device
        self.packed = None
        self.ws = None
        self.w = None
        self._alloc()

    def _alloc(self):
        pb = lib().mscnn_conv2d_packed_weight_bytes(self._p)
        wb = lib().mscnn_conv2d_workspace_bytes(self._p)
        if pb and (self.packed is None or self.packed.numel() * 4 < pb):
            self.packed = torch.empty(pb // 4, dtype=torch.float32, device=self.device)
        if wb and (self.ws is None or self.ws.numel() * 4 < wb):
            self.ws = torch.empty(wb // 4, dtype=torch.float32, device=self.device)

    @property
    def kernel(self):
        return lib().mscnn_conv2d_plan_kernel(self._p).decode()

    @property
    def dtype(self):
        return lib().mscnn_conv2d_plan_dtype(self._p).decode()

    @property
    def flops(self):
        return lib().mscnn_conv2d_plan_flops(self._p)

    @property
    def executed_flops(self):
        return lib().mscnn_conv2d_plan_executed_flops(self._p)

    def set_profiling(self, on=True):
        _check(lib().mscnn_conv2d_plan_set_profiling(self._p, int(on)))

    def stage_ms(self):
        """(input transform, MFMA GEMM kernels, output transform) of the last forward, HIP events on its stream."""
        out = (C.c_float * 3)()
        _check(lib().mscnn_conv2d_plan_stage_ms(self._p, out))
        return tuple(out)

    def out_shape(self):
        d = self.desc
        return (d.N, d.Cout, (d.H + 2 * d.pad_h - d.Kh) // d.stride_h + 1, (d.W + 2 * d.pad_w - d.Kw) // d.stride_w + 1)

    def set_batch(self, N):
        before, had = lib().mscnn_conv2d_plan_weight_layout(self._p), self.packed
        _check(lib().mscnn_conv2d_plan_set_batch(self._p, N))
        self.desc.N = N
        self._alloc()
        if getattr(self, "w", None) is not None and self.packed is not None and \
                (lib().mscnn_conv2d_plan_weight_layout(self._p) != before or self.packed is not had):
            self.pack(self.w)      # another kernel family (or a new buffer): the packed weights must be rebuilt

    def pack(self, w):
        self.w = w
        if self.packed is not None:
            _check(lib().mscnn_conv2d_pack_weights(self._p, _dev(w), _dev(self.packed), _stream()))

    @property
    def publishes_amax(self):
        return bool(lib().mscnn_conv2d_plan_publishes_amax(self._p))

    def set_amax_io(self, in_bound=None, out_amax=None):
        """max |x| hand-over (mscnn_conv2d_plan_set_amax_io): int32 device tensors of AMAX_SLOTS float bit patterns."""
        self._amax = (in_bound, out_amax)      # keep them alive
        _check(lib().mscnn_conv2d_plan_set_amax_io(self._p, _dev(in_bound), _dev(out_amax)))

    @property
    def can_pool(self):
        return bool(lib().mscnn_conv2d_plan_can_pool(self._p))

    def forward(self, x, bias=None, out=None, pool_out=None):
        """pool_out: tensor [N, Cout, ceil(Ho/2), ceil(Wo/2)] that receives the fused MAX 2x2 / stride 2 pooling."""
        if out is None:
            out = torch.empty(self.out_shape(), dtype=torch.float32, device=x.device)
        wsb = self.ws.numel() * 4 if self.ws is not None else 0
        _check(lib().mscnn_conv2d_fwd_pool_f32(self._p, _dev(x), _dev(self.w), _dev(self.packed), _dev(bias), _dev(out),
                                               _dev(pool_out), _dev(self.ws), wsb, _stream()))
        return out

    @property
    def can_pool_only(self):
        return bool(lib().mscnn_conv2d_plan_can_pool_only(self._p))

    def can_chain(self, nxt):
        return bool(lib().mscnn_conv2d_plan_can_chain(self._p, nxt._p))

    def forward_chain(self, x, nxt, bias=None, out=None, pool_out=None, write_y=True):
        """mscnn_conv2d_fwd_chain_f32: x None = this plan's planes were prepared (it was the `nxt` of the previous call);
        nxt given = its planes are written by this plan's output stage (y only if write_y); nxt None = ordinary output."""
        if out is None and write_y:
            out = torch.empty(self.out_shape(), dtype=torch.float32, device=self.ws.device)
        wsb = self.ws.numel() * 4
        _check(lib().mscnn_conv2d_fwd_chain_f32(self._p, nxt._p if nxt is not None else None, _dev(x), _dev(self.packed), _dev(bias),
                                                _dev(out) if write_y else None, _dev(pool_out), _dev(self.ws), wsb,
                                                _dev(nxt.ws) if nxt is not None else None, nxt.ws.numel() * 4 if nxt is not None else 0,
                                                _stream()))
        return out

    def can_fuse_roipool(self, Cc, pooled_h, pooled_w):
        return bool(lib().mscnn_conv2d_plan_can_fuse_roipool(self._p, Cc, pooled_h, pooled_w))

    def forward_roipool_pair(self, feat, rois, spatial_scale, pad_a, pad_b, bias=None, out=None, maps=None):
        """ROIPooling x 2 (pad_a -> channels [0, C), pad_b -> [C, 2C)) fused into this convolution's Winograd input stage
        (mscnn_conv2d_fwd_roipool_pair_f32): y = conv(concat(roipool(feat, pad_a), roipool(feat, pad_b)))."""
        N, Cc, H, W = feat.shape
        need = lib().mscnn_conv2d_roipool_workspace_bytes(self._p, N, Cc, H, W)
        if self.ws is None or self.ws.numel() * 4 < need:
            self.ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=feat.device)
        if out is None:
            out = torch.empty(self.out_shape(), dtype=torch.float32, device=feat.device)
        _check(lib().mscnn_conv2d_fwd_roipool_pair_f32(self._p, _dev(feat), _dev(maps), N, Cc, H, W, _dev(rois), spatial_scale, pad_a, pad_b,
                                                       _dev(self.packed), _dev(bias), _dev(out), _dev(self.ws), self.ws.numel() * 4,
                                                       _stream()))
        return out

    def __del__(self):
        try:
            if self._p:
                lib().mscnn_conv2d_plan_destroy(self._p)
                self._p = None
        except Exception:
            pass


def conv2d(x, w, bias=None, pad=(0, 0), stride=(1, 1), group=1, relu=False, algo=ALGO_AUTO):
    N, Cin, H, W = x.shape
    Cout, _, Kh, Kw = w.shape
    plan = ConvPlan(N, Cin, H, W, Cout, Kh, Kw, pad, stride, group, relu, device=x.device, algo=algo)
    plan.pack(w)
    y = plan.forward(x, bias)
    torch.cuda.current_stream().synchronize()   # plan buffers die with the plan
    return y


def relu(x, slope=0.0, inplace=False):
    y = x if inplace else torch.empty_like(x)
    _check(lib().mscnn_relu_fwd_f32(_dev(x), _dev(y), x.numel(), slope, _stream()))
    return y


def sum_squares(x):
    """mscnn_sum_squares_f32: a float64 device scalar holding sum x^2."""
    out = torch.zeros(1, dtype=torch.float64, device=x.device)
    _check(lib().mscnn_sum_squares_f32(_dev(x), x.numel(), _dev(out), _stream()))
    return out


def max_rel_diff(a, ref, floor=1.0):
    """mscnn_max_rel_diff_f32: max |a - ref| / max(floor, |ref|) (+inf when a NaN is involved), a float32 device scalar."""
    out = torch.zeros(1, dtype=torch.float32, device=a.device)
    _check(lib().mscnn_max_rel_diff_f32(_dev(a), _dev(ref), a.numel(), floor, _dev(out), _stream()))
    return out


def max_rel_diff_strided(a, a_stride, ref, ref_stride, planes, run, sumsq, sumsq_count):
    """mscnn_max_rel_diff_strided_f32: the same over `planes` runs of `run` floats (plane p at a + p * a_stride / ref + p * ref_stride),
    floor = max(1, sqrt(sumsq[0] / sumsq_count)) read on the device from `sumsq` (a float64 device scalar)."""
    out = torch.zeros(1, dtype=torch.float32, device=a.device)
    _check(lib().mscnn_max_rel_diff_strided_f32(_dev(a), a_stride, _dev(ref), ref_stride, planes, run, _dev(sumsq), float(sumsq_count), _dev(out),
                                                _stream()))
    return out


def store_words(dst, values):
    """mscnn_store_words_i32: dst[0 .. len(values)) = values (1 .. 4 int32 words) in one launch."""
    arr = (C.c_int * len(values))(*[int(v) for v in values])
    _check(lib().mscnn_store_words_i32(_dev(dst), arr, len(values), _stream()))


def pool_out_dim(i, k, p, s):
    return lib().mscnn_pool_out_dim(i, k, p, s)


def pool2d(x, kernel=(2, 2), pad=(0, 0), stride=(2, 2), method="MAX"):
    N, Cc, H, W = x.shape
    Ho, Wo = pool_out_dim(H, kernel[0], pad[0], stride[0]), pool_out_dim(W, kernel[1], pad[1], stride[1])
    y = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    _check(lib().mscnn_pool2d_fwd_f32(_dev(x), _dev(y), N, Cc, H, W, kernel[0], kernel[1], pad[0], pad[1],
                                      stride[0], stride[1], 0 if method == "MAX" else 1, _stream()))
    return y


def inner_product_wg_supported(M, N, K):
    return bool(lib().mscnn_inner_product_wg_supported(M, N, K))


def inner_product_wg(x, w, bias=None, relu=False, wt=None, out=None):
    """InnerProduct on the plane-GEMM kernel (mscnn_inner_product_wg_*); wt: the transposed weights from a previous call (returned)."""
    M = x.shape[0]
    K = x.numel() // M
    Nn = w.shape[0]
    if wt is None:
        wt = torch.empty((K, Nn), dtype=torch.float32, device=x.device)
        _check(lib().mscnn_inner_product_wg_pack(_dev(w), _dev(wt), Nn, K, _stream()))
    wb = lib().mscnn_inner_product_wg_workspace_bytes(M, Nn, K)
    ws = torch.empty((wb + 3) // 4, dtype=torch.float32, device=x.device)
    y = out if out is not None else torch.empty((M, Nn), dtype=torch.float32, device=x.device)
    _check(lib().mscnn_inner_product_wg_fwd(_dev(x), _dev(wt), _dev(bias), _dev(y), M, Nn, K, int(relu), _dev(ws), wb, _stream()))
    torch.cuda.current_stream().synchronize()
    return y, wt


def inner_product(x, w, bias=None, relu=False):
    M = x.shape[0]
    K = x.numel() // max(M, 1) if M else w.shape[1]
    Nn = w.shape[0]
    y = torch.empty((M, Nn), dtype=torch.float32, device=x.device)
    _check(lib().mscnn_inner_product_fwd_f32(_dev(x), _dev(w), _dev(bias), _dev(y), M, Nn, K, int(relu), _stream()))
    return y


def inner_product_x3(x, w, bias=None, relu=False, packed=None):
    """Split-fp16 InnerProduct (fp32-grade): w [N][K] is packed into exactly split fp16 hi + lo units, x is split on the device
    with a scale from its measured max |x|; three fp16 MFMAs per operand pair, fp32 accumulate."""
    M, K = x.shape[0], int(x.numel() // max(x.shape[0], 1))
    Nn = w.shape[0]
    if not lib().mscnn_inner_product_x3_supported(Nn, K):
        raise MscnnError(f"inner_product x3 needs N >= 128 and K % 32 == 0 (N={Nn}, K={K})")
    if packed is None:
        packed = torch.empty(lib().mscnn_inner_product_x3_packed_bytes(Nn, K), dtype=torch.uint8, device=x.device)
        _check(lib().mscnn_inner_product_x3_pack(_dev(w), _dev(packed), Nn, K, _stream()))
    wb = lib().mscnn_inner_product_x3_workspace_bytes(M, Nn, K)
    ws = torch.empty(wb, dtype=torch.uint8, device=x.device)
    y = torch.empty((M, Nn), dtype=torch.float32, device=x.device)
    _check(lib().mscnn_inner_product_x3_fwd(_dev(x), _dev(packed), _dev(bias), _dev(y), M, Nn, K, int(relu), None, _dev(ws), wb, _stream()))
    return y


def inner_product_f16(x, w, bias=None, relu=False):
    """fp16-operand InnerProduct: w is converted once to fp16 (here per call), x on the fly, fp32 accumulate."""
    M = x.shape[0]
    Nn, K = w.shape[0], w[0].numel()
    if not lib().mscnn_inner_product_f16_supported(Nn, K):
        raise MscnnError(f"inner_product f16 needs N >= 64 and K % 8 == 0 (N={Nn}, K={K})")
    w16 = torch.empty(Nn * K, dtype=torch.float16, device=x.device)
    _check(lib().mscnn_inner_product_pack_f16(_dev(w), _dev(w16), Nn, K, _stream()))
    y = torch.empty((M, Nn), dtype=torch.float32, device=x.device)
    _check(lib().mscnn_inner_product_fwd_f16(_dev(x), _dev(w16), _dev(bias), _dev(y), M, Nn, K, int(relu), _stream()))
    return y


def concat_channels(xs):
    N = xs[0].shape[0]
    inner = xs[0][0, 0].numel()
    ctot = sum(t.shape[1] for t in xs)
    y = torch.empty((N, ctot) + tuple(xs[0].shape[2:]), dtype=torch.float32, device=xs[0].device)
    off = 0
    for t in xs:
        _check(lib().mscnn_concat_channels_f32(_dev(t), _dev(y), N, t.shape[1], inner, ctot, off, _stream()))
        off += t.shape[1]
    return y


def deconv_depthwise(x, w, bias=None, pad=(0, 0), stride=(1, 1)):
    N, Cc, H, W = x.shape
    Kh, Kw = w.shape[2], w.shape[3]
    Ho, Wo = stride[0] * (H - 1) + Kh - 2 * pad[0], stride[1] * (W - 1) + Kw - 2 * pad[1]
    y = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    _check(lib().mscnn_deconv_depthwise_fwd_f32(_dev(x), _dev(w), _dev(bias), _dev(y), N, Cc, H, W, Kh, Kw,
                                                pad[0], pad[1], stride[0], stride[1], _stream()))
    return y


def softmax(x, axis=1):
    outer = 1
    for d in x.shape[:axis]:
        outer *= d
    inner = 1
    for d in x.shape[axis + 1:]:
        inner *= d
    y = torch.empty_like(x)
    _check(lib().mscnn_softmax_fwd_f32(_dev(x), _dev(y), outer, x.shape[axis], inner, _stream()))
    return y


def roipool(feat, rois, pooled_h, pooled_w, spatial_scale, pad_ratio=0.0, out=None, c_total=None, c_offset=0):
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    c_total = c_total or Cc
    if out is None:
        out = torch.empty((R, c_total, pooled_h, pooled_w), dtype=torch.float32, device=feat.device)
    _check(lib().mscnn_roipool_fwd_f32(_dev(feat), _dev(rois), _dev(out), R, N, Cc, H, W, pooled_h, pooled_w,
                                       spatial_scale, pad_ratio, c_total, c_offset, _stream()))
    return out


def roipool_maps(feat, stream=None):
    """The four maps the fused ROI-pooling stage reads (mscnn_roipool_maps_build_f32), built on `stream` (default: the current one)."""
    N, Cc, H, W = feat.shape
    maps = torch.empty(lib().mscnn_roipool_maps_bytes(N, Cc, H, W) // 4, dtype=torch.float32, device=feat.device)
    _check(lib().mscnn_roipool_maps_build_f32(_dev(feat), _dev(maps), N, Cc, H, W, C.c_void_p(stream.cuda_stream) if stream is not None else _stream()))
    return maps


def roipool_pair(feat, rois, pooled_h, pooled_w, spatial_scale, pad_a, pad_b):
    """Two ROI poolings (context paddings pad_a / pad_b) into channels [0, C) and [C, 2C) of one output, one launch."""
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = torch.empty((R, 2 * Cc, pooled_h, pooled_w), dtype=torch.float32, device=feat.device)
    _check(lib().mscnn_roipool_pair_fwd_f32(_dev(feat), _dev(rois), _dev(out), R, N, Cc, H, W, pooled_h, pooled_w, spatial_scale,
                                            pad_a, 0, pad_b, Cc, 2 * Cc, _stream()))
    return out


def roialign(feat, rois, pooled_h, pooled_w, spatial_scale, pad_ratio=0.0):
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = torch.empty((R, Cc, pooled_h + 1, pooled_w + 1), dtype=torch.float32, device=feat.device)
    _check(lib().mscnn_roialign_fwd_f32(_dev(feat), _dev(rois), _dev(out), R, N, Cc, H, W, pooled_h, pooled_w, spatial_scale,
                                        pad_ratio, _stream()))
    return out


def eltwise(xs, op="SUM", coeffs=None):
    n = len(xs)
    for t in xs:
        _dev(t)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
    cf = (C.c_float * n)(*coeffs) if coeffs is not None and len(coeffs) else None
    y = torch.empty_like(xs[0])
    _check(lib().mscnn_eltwise_fwd_f32(ptrs, n, cf, _dev(y), y.numel(), {"PROD": 0, "SUM": 1, "MAX": 2}[op], _stream()))
    return y


def make_boxoutput_desc(head_shapes, num, channels, field_w, field_h, downsample, fg_thr=-5.0, iou_thr=0.65,
                        nms_type="IOU", field_whr=2.0, field_xyr=2.0, max_nms_num=2000, max_post_nms_num=0,
                        min_size=15.0, bbox_mean=None, bbox_std=None):
    d = BoxOutputDesc()
    d.num_heads = len(head_shapes); d.num = num; d.channels = channels
    for j, (h, w) in enumerate(head_shapes):
        d.head_h[j] = h; d.head_w[j] = w
        d.field_w[j] = field_w[j]; d.field_h[j] = field_h[j]; d.downsample_rate[j] = downsample[j]
    d.fg_thr = fg_thr; d.iou_thr = iou_thr; d.nms_mode = NMS_MODES[nms_type]
    d.field_whr = field_whr; d.field_xyr = field_xyr
    d.max_nms_num = max_nms_num; d.max_post_nms_num = max_post_nms_num; d.min_size = min_size
    if bbox_mean is not None and bbox_std is not None and len(bbox_mean) and len(bbox_std):
        d.do_bbox_norm = 1
        for k in range(4):
            d.bbox_mean[k] = bbox_mean[k]; d.bbox_std[k] = bbox_std[k]
    return d


class BoxOutput:
    """Device-resident BoxOutput layer state: descriptor + workspace + output buffers."""

    def __init__(self, desc, device="cuda"):
        self.desc = desc
        wb = lib().mscnn_boxoutput_workspace_bytes(C.byref(desc))
        if wb == 0:
            raise MscnnError(lib().mscnn_last_error().decode())
        self.cap = lib().mscnn_boxoutput_max_rows(C.byref(desc))
        self.ws = torch.empty(wb, dtype=torch.uint8, device=device)
        self.rois = torch.empty((self.cap, 5), dtype=torch.float32, device=device)
        self.props = torch.empty((self.cap, 6), dtype=torch.float32, device=device)
        self.aids = torch.empty(self.cap, dtype=torch.int32, device=device)
        self.count = torch.zeros(2, dtype=torch.int32, device=device)

    def forward_async(self, heads):
        n = self.desc.num_heads
        ptrs = (C.c_void_p * n)(*[h.data_ptr() for h in heads])
        for h in heads:
            _dev(h)
        _check(lib().mscnn_boxoutput_fwd_f32(C.byref(self.desc), ptrs, _dev(self.rois), _dev(self.props), _dev(self.aids),
                                             self.cap, _dev(self.count), _dev(self.ws), self.ws.numel(), _stream()))

    def forward(self, heads):
        self.forward_async(heads)
        R, nreal = self.count.tolist()      # the one small D2H of the layer (R drives the next Reshape)
        return self.rois[:R], self.props[:R], self.aids[:R], nreal


def detections_cascade(boxes, cls_prob, props, cls_id, det_thr=0.0, ratios=(1.0, 1.0), org_hw=(375, 1242), nms_overlap=0.5):
    """Final stage of the cascade drivers (run_cascademscnn.m:84-117) for one cascade stage's blobs."""
    R = props.shape[0]
    d = DetectionsDesc()
    d.ncls = cls_prob.shape[1]; d.cls_id = cls_id
    d.ratio_h, d.ratio_w = ratios
    d.org_h, d.org_w = org_hw
    d.nms_overlap = nms_overlap
    dev = props.device
    dets = torch.zeros((max(R, 1), 5), dtype=torch.float64, device=dev)
    ids = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    wb = lib().mscnn_detections_workspace_bytes(R)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    _check(lib().mscnn_detections_cascade_fwd(C.byref(d), C.c_float(det_thr), _dev(boxes), _dev(cls_prob), _dev(props), R,
                                              _dev(dets), _dev(ids), _dev(count), _dev(ws), C.c_size_t(wb), _stream()))
    D = int(count.item())
    return dets[:D], ids[:D]


def preprocess(img_rgb_u8, out_h, out_w, mean_bgr=(104.0, 117.0, 123.0), out=None):
    """run_mscnn_detection.m:64-69 on the device: uint8 HWC RGB image (cuda tensor) -> the net's (1, 3, out_h, out_w) input."""
    oh, ow, ch = img_rgb_u8.shape
    if ch != 3 or img_rgb_u8.dtype != torch.uint8:
        raise MscnnError("preprocess: expected a uint8 [H, W, 3] image")
    if out is None:
        out = torch.empty((1, 3, out_h, out_w), dtype=torch.float32, device=img_rgb_u8.device)
    L = lib()
    L.mscnn_preprocess_workspace_bytes.restype = C.c_size_t
    wb = L.mscnn_preprocess_workspace_bytes(oh, ow, out_h, out_w)
    ws = torch.empty(wb, dtype=torch.uint8, device=img_rgb_u8.device)
    m = (C.c_float * 3)(*mean_bgr)
    _check(L.mscnn_preprocess_u8_f32(_dev(img_rgb_u8), oh, ow, _dev(out), out_h, out_w, m, _dev(ws), C.c_size_t(wb), _stream()))
    torch.cuda.current_stream().synchronize()     # ws dies with this frame
    return out


def nms_greedy(boxes_xywh, thr, mode="IOU"):
    n = boxes_xywh.shape[0]
    keep = torch.zeros(max(n, 1), dtype=torch.uint8, device=boxes_xywh.device)
    wb = lib().mscnn_nms_workspace_bytes(n)
    ws = torch.empty(wb, dtype=torch.uint8, device=boxes_xywh.device)
    _check(lib().mscnn_nms_greedy_f32(_dev(boxes_xywh), n, thr, NMS_MODES[mode], _dev(keep), _dev(ws), wb, _stream()))
    return keep[:n].bool()


def decode_bbox(bbox, prior, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    R = bbox.shape[0]
    out = torch.empty((R, 5), dtype=torch.float32, device=bbox.device)
    m = (C.c_float * 4)(*mean); s = (C.c_float * 4)(*std)
    _check(lib().mscnn_decodebbox_fwd_f32(_dev(bbox), _dev(prior), _dev(out), R, bbox.shape[1], m, s, _stream()))
    return out


def detections(bbox_pred, cls_pred, props, cls_id, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2),
               proposal_thr=-10.0, ratios=(1.0, 1.0), org_hw=(375, 1242), nms_overlap=0.5):
    R = props.shape[0]
    d = DetectionsDesc()
    d.ncls = cls_pred.shape[1]; d.cls_id = cls_id
    for k in range(4):
        d.bbox_mean[k] = bbox_mean[k]; d.bbox_std[k] = bbox_std[k]
    d.proposal_thr = proposal_thr
    d.ratio_h, d.ratio_w = ratios
    d.org_h, d.org_w = org_hw
    d.nms_overlap = nms_overlap
    dev = props.device
    dets = torch.zeros((max(R, 1), 5), dtype=torch.float64, device=dev)
    ids = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    wb = lib().mscnn_detections_workspace_bytes(R)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    _check(lib().mscnn_detections_fwd(C.byref(d), _dev(bbox_pred), _dev(cls_pred), _dev(props), R, _dev(dets), _dev(ids),
                                      _dev(count), _dev(ws), wb, _stream()))
    D = int(count.item())
    return dets[:D], ids[:D]


# ---- health of the plane-GEMM kernel's stream-K hand-off (mscnn_hip.h) ----
def wgemm_handoff_event():
    """Tag of the last launch on the current device whose finisher gave up on a contributor (0: none); synchronise first."""
    return int(lib().mscnn_wgemm_handoff_event())


def wgemm_force_whole_tiles(on=True):
    lib().mscnn_wgemm_force_whole_tiles(int(bool(on)))


def wgemm_whole_tiles_forced():
    return bool(lib().mscnn_wgemm_whole_tiles_forced())


def debug_wgemm_handoff_fault(drop_publish=False, spin_limit=0):
    """Fault injection for the tests: contributors never publish their partial sums; finishers give up after spin_limit polls."""
    lib().mscnn_debug_wgemm_handoff_fault(int(bool(drop_publish)), int(spin_limit))


def debug_inner_product_rows(rows=0):
    """Rows of x per workgroup of the small-N InnerProduct kernel: 2, 4 or 8; 0 = the default (4 for N <= 5, else 2) -- tests and A/B runs."""
    lib().mscnn_debug_inner_product_rows(int(rows))
