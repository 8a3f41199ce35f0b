"""Host-side mirror of the reference extractor interface over the C ABI.

`SPExtractor` keeps the call shape of the reference class
(/root/reference/orb_slam2/include/orb_slam/cv/sp_extractor.h:49-88):
construct once with the number of features, call it with a CV_8UC1 image and an
(ignored) mask, get keypoints + a K x 256 float32 descriptor matrix, then read
the side outputs the tracker copies right after the call
(/root/reference/orb_slam2/src/type/frame.cpp:296-314): getCov2Inv(),
dense_dust_, heat_, occ_grid_.  Everything is computed by libspfe.so
(hand-written HIP for gfx950); there is NO CPU path: importing works anywhere,
constructing an extractor without the library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libspfe.so")

SPFE_FLAG_HEAT = 1
SPFE_FLAG_ASYNC_COV = 2
SPFE_FLAG_DESC_BF16 = 4   # records / results carry bf16 descriptors (RNE of the f32 ones)
SPFE_FLAG_LAZY_HEAT_INV = 8   # with SPFE_FLAG_HEAT: host calls bring back `heat` only; fetch_heat_inv() on demand
SPFE_PRECISION_F32 = 0
SPFE_PRECISION_BF16 = 1
NUM_PARAMS = 1300865
ABI_VERSION = 5           # SPFE_ABI_VERSION of include/spfe.h these ctypes structures mirror
_ERRORS = {-1: "SPFE_EINVAL", -2: "SPFE_EEMPTY", -3: "SPFE_EHIP", -4: "SPFE_EWEIGHTS"}

# every symbol include/spfe.h declares (tests check that the library exports all)
ABI_SYMBOLS = [
    "spfe_create", "spfe_destroy", "spfe_extract", "spfe_extract_batch", "spfe_postprocess",
    "spfe_get_record_layout", "spfe_record_bytes", "spfe_extract_batch_device", "spfe_last_ticket",
    "spfe_wait_records",
    "spfe_view_record", "spfe_debug_read", "spfe_stage_times", "spfe_stage_reset",
    "spfe_stage_name",
    "spfe_math_probe", "spfe_last_error", "spfe_version", "spfe_abi_version", "spfe_check_abi",
    "spfe_match", "spfe_match_records_device", "spfe_match_out_bytes",
    "spfe_match_patches", "spfe_match_patches_record_device",
    "spfe_set_staging", "spfe_extract_staged", "spfe_extract_batch_staged", "spfe_stage_batch_device",
    "spfe_comm_unique_id", "spfe_comm_init", "spfe_comm_destroy", "spfe_allgather_records", "spfe_comm_wait",
    "spfe_comm_stream", "spfe_comm_count", "spfe_submit_batch", "spfe_collect_batch",
    "spfe_align_dust", "spfe_align_dust_record_device", "spfe_align_dust_batch_device", "spfe_match_knn2",
    "spfe_track_dust_record_device", "spfe_fetch_heat_inv",
    "spfe_extract_begin", "spfe_extract_maps", "spfe_extract_rows", "spfe_extract_finish", "spfe_set_map_buffers",
]


class SpfeError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("num_features", C.c_int),
                ("max_batch", C.c_int), ("device", C.c_int), ("precision", C.c_int),
                ("flags", C.c_uint), ("weights", C.c_void_p), ("weights_path", C.c_char_p)]


class _Result(C.Structure):
    _fields_ = [("K", C.c_int), ("n_candidates", C.c_int), ("status", C.c_int),
                ("reserved", C.c_int), ("kp_xy", C.c_void_p),
                ("kp_response", C.c_void_p), ("desc", C.c_void_p), ("cov2", C.c_void_p),
                ("cov2_inv", C.c_void_p), ("occ_grid", C.c_void_p), ("dense_dust", C.c_void_p),
                ("semi_dust", C.c_void_p), ("heat", C.c_void_p), ("heat_inv", C.c_void_p),
                ("desc_bf16", C.c_void_p)]


class RecordLayout(C.Structure):
    _fields_ = [("bytes", C.c_size_t), ("kmax", C.c_int), ("off_hdr", C.c_size_t),
                ("off_xy", C.c_size_t), ("off_resp", C.c_size_t), ("off_cov", C.c_size_t),
                ("off_cinv", C.c_size_t), ("off_desc", C.c_size_t), ("off_occ", C.c_size_t),
                ("off_dd", C.c_size_t), ("off_sd", C.c_size_t), ("desc_elem_bytes", C.c_int)]


class _DustParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("max_iterations", C.c_int), ("huber_delta", C.c_double), ("inlier_chi2", C.c_double)]


DUST_MAX_POINTS = 512
DUST_OFF_UV = 128
DUST_OFF_INLIER = 128 + DUST_MAX_POINTS * 8
DUST_OUT_BYTES = 128 + DUST_MAX_POINTS * 9


class _Staging(C.Structure):
    _fields_ = [("src_height", C.c_int), ("src_width", C.c_int), ("channels", C.c_int), ("rgb", C.c_int),
                ("map_x", C.c_void_p), ("map_y", C.c_void_p)]


_lib = None


def _bind_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels ship their own libamdhip64.so.7 (+ HSA
    runtime) under torch/lib; libspfe.so needs the same SONAME.  Whichever copy the dynamic loader
    sees first serves both, and a torch imported AFTER libspfe had pulled in /opt/rocm's copy finds
    no devices.  So when a torch wheel with a bundled runtime is installed (it is the plumbing the
    multi-GPU driver uses), load that copy first; torch itself is not imported here."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen libspfe.so (built by __graft_entry__.build()).  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpfeError("libspfe.so not built at %s — run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (there is no CPU fallback)" % LIB_PATH)
    _bind_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.spfe_create.restype = C.c_int
    L.spfe_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    L.spfe_destroy.restype = None
    L.spfe_destroy.argtypes = [C.c_void_p]
    L.spfe_extract.restype = C.c_int
    L.spfe_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Result)]
    L.spfe_extract_batch.restype = C.c_int
    L.spfe_extract_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                     C.POINTER(_Result)]
    L.spfe_postprocess.restype = C.c_int
    L.spfe_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Result)]
    L.spfe_extract_begin.restype = C.c_int
    L.spfe_extract_begin.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    L.spfe_extract_maps.restype = C.c_int
    L.spfe_extract_maps.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float))]
    L.spfe_set_map_buffers.restype = C.c_int
    L.spfe_set_map_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.spfe_extract_rows.restype = C.c_int
    L.spfe_extract_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_float))]
    L.spfe_extract_finish.restype = C.c_int
    L.spfe_extract_finish.argtypes = [C.c_void_p, C.POINTER(_Result)]
    L.spfe_get_record_layout.restype = C.c_int
    L.spfe_get_record_layout.argtypes = [C.c_void_p, C.POINTER(RecordLayout)]
    L.spfe_record_bytes.restype = C.c_size_t
    L.spfe_record_bytes.argtypes = [C.c_void_p]
    L.spfe_extract_batch_device.restype = C.c_int
    L.spfe_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.spfe_last_ticket.restype = C.c_long
    L.spfe_last_ticket.argtypes = [C.c_void_p]
    L.spfe_wait_records.restype = C.c_int
    L.spfe_wait_records.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    L.spfe_view_record.restype = C.c_int
    L.spfe_view_record.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Result)]
    L.spfe_debug_read.restype = C.c_long
    L.spfe_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]
    L.spfe_stage_times.restype = C.c_int
    L.spfe_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    L.spfe_stage_reset.restype = C.c_int
    L.spfe_stage_reset.argtypes = [C.c_void_p]
    L.spfe_stage_name.restype = C.c_char_p
    L.spfe_stage_name.argtypes = [C.c_int]
    L.spfe_math_probe.restype = C.c_int
    L.spfe_math_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.spfe_set_staging.restype = C.c_int
    L.spfe_set_staging.argtypes = [C.c_void_p, C.POINTER(_Staging)]
    L.spfe_extract_staged.restype = C.c_int
    L.spfe_extract_staged.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Result)]
    L.spfe_extract_batch_staged.restype = C.c_int
    L.spfe_extract_batch_staged.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                            C.POINTER(_Result)]
    L.spfe_stage_batch_device.restype = C.c_int
    L.spfe_stage_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.spfe_match_patches.restype = C.c_int
    L.spfe_match_patches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_float, C.c_void_p]
    L.spfe_match_patches_record_device.restype = C.c_int
    L.spfe_match_patches_record_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                   C.c_float, C.c_void_p, C.c_void_p]
    L.spfe_match.restype = C.c_int
    L.spfe_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p]
    L.spfe_match_knn2.restype = C.c_int
    L.spfe_match_knn2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.spfe_match_records_device.restype = C.c_int
    L.spfe_match_records_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p]
    L.spfe_match_out_bytes.restype = C.c_size_t
    L.spfe_match_out_bytes.argtypes = [C.c_void_p]
    L.spfe_align_dust.restype = C.c_int
    L.spfe_align_dust.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(_DustParams),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.spfe_align_dust_record_device.restype = C.c_int
    L.spfe_align_dust_record_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.POINTER(_DustParams), C.c_void_p, C.c_void_p]
    L.spfe_track_dust_record_device.restype = C.c_int
    L.spfe_track_dust_record_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.POINTER(_DustParams), C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.spfe_align_dust_batch_device.restype = C.c_int
    L.spfe_align_dust_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(_DustParams), C.c_void_p, C.c_void_p]
    L.spfe_submit_batch.restype = C.c_int
    L.spfe_submit_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_long)]
    L.spfe_collect_batch.restype = C.c_int
    L.spfe_collect_batch.argtypes = [C.c_void_p, C.c_long, C.POINTER(_Result)]
    L.spfe_comm_unique_id.restype = C.c_int
    L.spfe_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
    L.spfe_comm_init.restype = C.c_int
    L.spfe_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.spfe_comm_destroy.restype = C.c_int
    L.spfe_comm_destroy.argtypes = [C.c_void_p]
    L.spfe_allgather_records.restype = C.c_int
    L.spfe_allgather_records.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int]
    L.spfe_comm_wait.restype = C.c_int
    L.spfe_comm_wait.argtypes = [C.c_void_p, C.c_void_p]
    L.spfe_comm_stream.restype = C.c_void_p
    L.spfe_comm_stream.argtypes = [C.c_void_p]
    L.spfe_comm_count.restype = C.c_int
    L.spfe_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.spfe_last_error.restype = C.c_char_p
    L.spfe_version.restype = C.c_char_p
    # the structures above mirror include/spfe.h by hand: a library built from another header revision is refused here,
    # not discovered as overrun arrays later
    try:
        L.spfe_check_abi.restype = C.c_int
        L.spfe_check_abi.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    except AttributeError:
        raise SpfeError("libspfe.so at %s predates spfe_check_abi (ABI < %d): rebuild it" % (LIB_PATH, ABI_VERSION))
    if L.spfe_check_abi(ABI_VERSION, C.sizeof(_Config), C.sizeof(_Result), C.sizeof(RecordLayout)) != 0:
        raise SpfeError(L.spfe_last_error().decode())
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        msg = load_library().spfe_last_error().decode()
        if rc == -2:
            # the reference throws std::runtime_error("input image is empty")
            # (sp_extractor.cpp:364-365)
            raise RuntimeError("input image is empty")
        raise SpfeError("%s: %s" % (_ERRORS.get(rc, rc), msg))


KEYPOINT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32),
                           ("angle", np.float32), ("response", np.float32),
                           ("octave", np.int32), ("class_id", np.int32)])


def _as_np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class FrameResult:
    """Deep copies of everything one extractor call produced for one frame."""

    def __init__(self, r, H, W, with_heat):
        hc, wc = H // 8, W // 8
        K = r.K
        self.K = K
        self.n_candidates = r.n_candidates
        self.status = r.status
        xy = _as_np(r.kp_xy, (K, 2), np.float32)
        resp = _as_np(r.kp_response, (K,), np.float32)
        kps = np.zeros(K, KEYPOINT_DTYPE)
        # cv::KeyPoint(x, y, 1.0f): angle -1, octave 0, class_id -1 (sp_extractor.cpp:231-232)
        kps["x"], kps["y"] = xy[:, 0], xy[:, 1]
        kps["size"], kps["angle"], kps["octave"], kps["class_id"] = 1.0, -1.0, 0, -1
        kps["response"] = resp  # :271
        self.keypoints = kps
        self.kp_xy = xy
        self.response = resp
        if r.desc:
            self.descriptors = _as_np(r.desc, (K, 256), np.float32)
            self.descriptors_bf16 = None
        else:   # SPFE_FLAG_DESC_BF16: the record carries bf16 rows; `descriptors` is their exact widening to f32
            self.descriptors_bf16 = _as_np(r.desc_bf16, (K, 256), np.uint16)
            self.descriptors = (self.descriptors_bf16.astype(np.uint32) << 16).view(np.float32)
        self.cov2 = _as_np(r.cov2, (K, 2), np.float32)
        self.cov2_inv = _as_np(r.cov2_inv, (K, 2), np.float32)
        self.occ_grid = _as_np(r.occ_grid, (hc, wc), np.int16)
        self.dense_dust = _as_np(r.dense_dust, (hc, wc), np.float32)
        self.semi_dust = _as_np(r.semi_dust, (hc, wc), np.float32)
        self.heat = _as_np(r.heat, (H, W), np.float32) if with_heat and r.heat else None
        self.heat_inv = _as_np(r.heat_inv, (H, W), np.float32) if with_heat and r.heat_inv else None


class SPExtractor:
    """MI355X SuperPoint extractor with the reference's call signature.

    Reference constructor: SPExtractor(int nfeatures) reading camera::height,
    camera::width and common::model_path from globals (sp_extractor.cpp:342-359);
    here they are explicit arguments.
    """

    def __init__(self, nfeatures, height, width, weights, max_batch=1, device=0, with_heat=True,
                 async_cov=False, precision="f32", desc_bf16=False, lazy_heat_inv=False):
        self._h = C.c_void_p()
        self._lib = load_library()
        self.nfeatures, self.height, self.width = int(nfeatures), int(height), int(width)
        self.max_batch, self.with_heat = int(max_batch), bool(with_heat)
        cfg = _Config()
        cfg.height, cfg.width, cfg.num_features = self.height, self.width, self.nfeatures
        if precision not in ("f32", "bf16"):
            raise SpfeError("precision must be 'f32' or 'bf16'")
        self.precision = precision
        cfg.max_batch, cfg.device = self.max_batch, int(device)
        cfg.precision = SPFE_PRECISION_BF16 if precision == "bf16" else SPFE_PRECISION_F32
        cfg.flags = (SPFE_FLAG_HEAT if with_heat else 0) | (SPFE_FLAG_ASYNC_COV if async_cov else 0) | \
            (SPFE_FLAG_DESC_BF16 if desc_bf16 else 0) | (SPFE_FLAG_LAZY_HEAT_INV if lazy_heat_inv and with_heat else 0)
        self.desc_bf16 = bool(desc_bf16)
        self.async_cov = bool(async_cov)
        keep = None
        if isinstance(weights, (str, bytes, os.PathLike)):
            cfg.weights, cfg.weights_path = None, os.fsencode(weights)
        else:
            keep = np.ascontiguousarray(weights, np.float32)
            if keep.size != NUM_PARAMS:
                raise SpfeError("weight blob has %d params, expected %d" % (keep.size, NUM_PARAMS))
            cfg.weights, cfg.weights_path = keep.ctypes.data, None
        _check(self._lib.spfe_create(C.byref(cfg), C.byref(self._h)))
        del keep
        self.layout = RecordLayout()
        _check(self._lib.spfe_get_record_layout(self._h, C.byref(self.layout)))
        # BaseExtractor(n, 1.0, 1, 1, 1): one pyramid level (base_extractor.h:12-47)
        self.nlevels, self.scaleFactor = 1, 1.0
        self.semi_dust_ = self.dense_dust_ = self.heat_ = self.heat_inv_ = self.occ_grid_ = None
        self.mask_ = None  # never written by the reference either
        self._cov2 = self._cov2_inv = None
        self.last = None

    # -- BaseExtractor getters (base_extractor.h:58-72) --
    def GetLevels(self):
        return 1

    def GetScaleFactor(self):
        return 1.0

    def GetScaleFactors(self):
        return [1.0]

    def GetInverseScaleFactors(self):
        return [1.0]

    def GetScaleSigmaSquares(self):
        return [1.0]

    def GetInverseScaleSigmaSquares(self):
        return [1.0]

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.spfe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_image(self, image):
        if image is None or getattr(image, "size", 0) == 0:
            raise RuntimeError("input image is empty")  # sp_extractor.cpp:364-365
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise SpfeError("image must be CV_8UC1 (2-D uint8)")  # assert at :368
        if img.shape != (self.height, self.width):
            raise SpfeError("image is %s, extractor was built for %s" %
                            (img.shape, (self.height, self.width)))
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        return img

    def _publish(self, fr):
        self.last = fr
        self.semi_dust_, self.dense_dust_ = fr.semi_dust, fr.dense_dust
        self.heat_, self.heat_inv_, self.occ_grid_ = fr.heat, fr.heat_inv, fr.occ_grid
        self._cov2, self._cov2_inv = fr.cov2, fr.cov2_inv

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors) — mask is ignored (:361-363)."""
        img = self._check_image(image)
        r = _Result()
        _check(self._lib.spfe_extract(self._h, img.ctypes.data, img.strides[0], C.byref(r)))
        fr = FrameResult(r, self.height, self.width, self.with_heat)
        self._publish(fr)
        return fr.keypoints, fr.descriptors

    def extract_batch(self, images):
        """n independent frames in one call; returns a list of FrameResult."""
        imgs = [self._check_image(im) for im in images]
        n = len(imgs)
        if n == 0:
            raise RuntimeError("input image is empty")
        strides = {im.strides[0] for im in imgs}
        if len(strides) != 1:
            imgs = [np.ascontiguousarray(im) for im in imgs]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        res = (_Result * n)()
        _check(self._lib.spfe_extract_batch(self._h, ptrs, imgs[0].strides[0], n, res))
        out = [FrameResult(res[i], self.height, self.width, self.with_heat) for i in range(n)]
        self._publish(out[-1])
        return out

    # -- the synchronous call in three parts (spfe.h: spfe_extract_begin / _maps / _finish) --
    def extract_begin(self, images):
        """Enqueue what extract_batch(images) runs and return at once; extract_finish() delivers the results."""
        imgs = [np.ascontiguousarray(self._check_image(im)) for im in images]
        if not imgs:
            raise RuntimeError("input image is empty")
        ptrs = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        _check(self._lib.spfe_extract_begin(self._h, ptrs, imgs[0].strides[0], len(imgs)))
        self._open_n = len(imgs)

    def extract_maps(self):
        """Block until the open call's H x W maps are in host memory: (heat [n,H,W], heat_inv [n,H,W] or None) as views of
        the library's buffers — or (None, None) when the maps travel with the record in this call."""
        ph, pi = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        _check(self._lib.spfe_extract_maps(self._h, C.byref(ph), None))      # heat arrives first ...
        _check(self._lib.spfe_extract_maps(self._h, None, C.byref(pi)))      # ... heat_inv behind it
        n = getattr(self, "_open_n", 0)
        view = lambda p: np.ctypeslib.as_array(p, shape=(n, self.height, self.width)) if p else None
        return view(ph), view(pi)

    def extract_rows(self, frame=0):
        """Block until the open call's descriptor rows of `frame` are in host memory: [K, 256] f32 view of the library's
        buffer, or None when the rows travel with the record in this call."""
        k, p = C.c_int(0), C.POINTER(C.c_float)()
        _check(self._lib.spfe_extract_rows(self._h, int(frame), C.byref(k), C.byref(p)))
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(max(k.value, 1), 256))[:k.value]

    def extract_finish(self):
        """The rest of the call begun by extract_begin(); returns what extract_batch would have."""
        n = max(getattr(self, "_open_n", 0), 1)
        res = (_Result * n)()
        self._open_n = 0
        _check(self._lib.spfe_extract_finish(self._h, res))
        out = [FrameResult(res[i], self.height, self.width, self.with_heat) for i in range(n)]
        self._publish(out[-1])
        return out

    def set_map_buffers(self, heat=None, heat_inv=None):
        """spfe_set_map_buffers: the synchronous calls' H x W maps straight into these float32 arrays ([max_batch, H, W],
        C-contiguous; the extractor keeps them alive and the library page-locks them while set); None = the library's buffer."""
        for a in (heat, heat_inv):
            if a is not None and (a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] or
                                  a.size != self.max_batch * self.height * self.width):
                raise SpfeError("map buffers must be C-contiguous float32 [max_batch, H, W]")
        _check(self._lib.spfe_set_map_buffers(self._h, heat.ctypes.data if heat is not None else None,
                                              heat_inv.ctypes.data if heat_inv is not None else None))
        self._map_buffers = (heat, heat_inv)

    # -- direct "dust" alignment (optimizer_dust.cpp:170-294) --
    @staticmethod
    def _dust_params(fx, fy, cx, cy, max_iterations, huber_delta, inlier_chi2):
        return _DustParams(float(fx), float(fy), float(cx), float(cy), int(max_iterations), float(huber_delta),
                           float(inlier_chi2))

    def align_dust(self, dense_dust, points_xyz, Tcw, fx, fy, cx, cy, max_iterations=40, huber_delta=0.9,
                   inlier_chi2=0.9):
        """Optimizer::PoseOptimizationDust(pFrame, mps, is_visible): -> dict(Tcw [4,4] f32, inlier bool[n],
        uv f32 [n,2] (dust_proj_u/v), n_inlier, iterations)."""
        dust = np.ascontiguousarray(dense_dust, np.float32)
        if dust.shape != (self.height // 8, self.width // 8):
            raise SpfeError("dense_dust must be [H/8, W/8]")
        pts = np.ascontiguousarray(points_xyz, np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        n = len(pts)
        Tout = np.zeros(16, np.float32)
        inl = np.zeros(max(n, 1), np.uint8)
        uv = np.zeros((max(n, 1), 2), np.float32)
        ni, it = C.c_int(0), C.c_int(0)
        prm = self._dust_params(fx, fy, cx, cy, max_iterations, huber_delta, inlier_chi2)
        _check(self._lib.spfe_align_dust(self._h, dust.ctypes.data, pts.ctypes.data, n, T.ctypes.data, C.byref(prm),
                                         Tout.ctypes.data, inl.ctypes.data, uv.ctypes.data, C.byref(ni), C.byref(it)))
        return dict(Tcw=Tout.reshape(4, 4), inlier=inl[:n].astype(bool), uv=uv[:n], n_inlier=ni.value,
                    iterations=it.value)

    def align_dust_record_device(self, d_record, d_points_xyz, n, d_Tcw, d_out, fx, fy, cx, cy, max_iterations=40,
                                 huber_delta=0.9, inlier_chi2=0.9, stream=None):
        prm = self._dust_params(fx, fy, cx, cy, max_iterations, huber_delta, inlier_chi2)
        _check(self._lib.spfe_align_dust_record_device(self._h, C.c_void_p(d_record), C.c_void_p(d_points_xyz), int(n),
                                                       C.c_void_p(d_Tcw), C.byref(prm), C.c_void_p(d_out),
                                                       C.c_void_p(stream or 0)))

    def track_dust_record_device(self, d_record, d_points_xyz, d_mp_desc, n, d_Tcw, d_dust_out, d_kp_idx, fx, fy, cx, cy,
                                 min_inliers=0, max_dist=0.75, max_iterations=40, huber_delta=0.9, inlier_chi2=0.9, stream=None):
        """Tracking::trackFrameDustKFLocal's chain behind the extraction (tracker_dust.cpp:92-172) on a resident record:
        PoseOptimizationDust, then the patch-wise association of the in_view points at their projections
        (spfe_track_dust_record_device); nothing leaves HBM in between."""
        prm = self._dust_params(fx, fy, cx, cy, max_iterations, huber_delta, inlier_chi2)
        _check(self._lib.spfe_track_dust_record_device(self._h, C.c_void_p(d_record), C.c_void_p(d_points_xyz),
                                                       C.c_void_p(d_mp_desc), int(n), C.c_void_p(d_Tcw), C.byref(prm),
                                                       int(min_inliers), float(max_dist), C.c_void_p(d_dust_out),
                                                       C.c_void_p(d_kp_idx), C.c_void_p(stream or 0)))

    def align_dust_batch_device(self, d_records, n_frames, d_points_xyz, d_n_points, d_Tcw, d_out, fx, fy, cx, cy,
                                max_iterations=40, huber_delta=0.9, inlier_chi2=0.9, stream=None):
        """n_frames independent solves in one launch (spfe_align_dust_batch_device): frame f uses record f of `d_records`,
        the points at d_points_xyz + f * DUST_MAX_POINTS * 3 floats (d_n_points[f] of them), pose d_Tcw + 16 f, and writes
        d_out + f * DUST_OUT_BYTES."""
        prm = self._dust_params(fx, fy, cx, cy, max_iterations, huber_delta, inlier_chi2)
        _check(self._lib.spfe_align_dust_batch_device(self._h, C.c_void_p(d_records), int(n_frames), C.c_void_p(d_points_xyz),
                                                      C.c_void_p(d_n_points), C.c_void_p(d_Tcw), C.byref(prm),
                                                      C.c_void_p(d_out), C.c_void_p(stream or 0)))

    @staticmethod
    def decode_dust_out(host_block, n):
        b = np.ascontiguousarray(host_block, np.uint8)
        cnt = b[64:72].view(np.int32)
        return dict(Tcw=b[:64].view(np.float32).reshape(4, 4).copy(), n_inlier=int(cnt[0]), iterations=int(cnt[1]),
                    uv=b[DUST_OFF_UV:DUST_OFF_UV + n * 8].view(np.float32).reshape(n, 2).copy(),
                    inlier=b[DUST_OFF_INLIER:DUST_OFF_INLIER + n].astype(bool))

    # -- pipelined host path: up to 3 batches in flight --
    def submit_batch(self, images):
        """Copy `images` (list of uint8 [H, W]) into pinned staging, enqueue H2D + the whole path + D2H of the
        records, return a ticket immediately (spfe_submit_batch)."""
        imgs = [self._check_image(im) for im in images]
        n = len(imgs)
        if n == 0:
            raise RuntimeError("input image is empty")
        if len({im.strides[0] for im in imgs}) != 1:
            imgs = [np.ascontiguousarray(im) for im in imgs]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        t = C.c_long(-1)
        _check(self._lib.spfe_submit_batch(self._h, ptrs, imgs[0].strides[0], n, C.byref(t)))
        self._pipe_n = getattr(self, "_pipe_n", {})
        self._pipe_n[t.value] = n
        return t.value

    def collect_batch(self, ticket, copy=True):
        """Block until batch `ticket` is back in host memory; list of FrameResult (deep copies).  copy=False
        returns the raw spfe_result array instead (pointers into the library's pinned buffers, valid until
        three further submits) — what a C++ caller gets."""
        n = self._pipe_n.pop(ticket)
        res = (_Result * n)()
        _check(self._lib.spfe_collect_batch(self._h, int(ticket), res))
        if not copy:
            return res
        out = [FrameResult(res[i], self.height, self.width, self.with_heat) for i in range(n)]
        self._publish(out[-1])
        return out

    def postprocess(self, semi, coarse):
        """Tail + selection + descriptors + covariance from host semi/coarse maps
        ([n,hc,wc,65], [n,hc,wc,256], or a single frame without the n axis)."""
        semi = np.ascontiguousarray(semi, np.float32)
        coarse = np.ascontiguousarray(coarse, np.float32)
        hc, wc = self.height // 8, self.width // 8
        semi = semi.reshape(-1, hc, wc, 65)
        coarse = coarse.reshape(-1, hc, wc, 256)
        n = semi.shape[0]
        res = (_Result * n)()
        _check(self._lib.spfe_postprocess(self._h, semi.ctypes.data, coarse.ctypes.data, n, res))
        out = [FrameResult(res[i], self.height, self.width, self.with_heat) for i in range(n)]
        self._publish(out[-1])
        return out

    # -- side outputs (sp_extractor.h:61-73) --
    def getCov(self):
        return self._cov2

    def getCov2Inv(self):
        return self._cov2_inv

    def getHeatMap(self):
        return self.heat_

    def getMask(self):
        return self.mask_

    # -- device-resident path --
    def record_bytes(self):
        return int(self._lib.spfe_record_bytes(self._h))

    def extract_batch_device(self, d_images, n, d_records=None, stream=None):
        """Enqueue n frames already in device memory (raw device pointers as ints)."""
        if not d_images:
            raise RuntimeError("input image is empty")
        _check(self._lib.spfe_extract_batch_device(self._h, C.c_void_p(d_images), int(n),
                                                   C.c_void_p(d_records or 0),
                                                   C.c_void_p(stream or 0)))
        return int(self._lib.spfe_last_ticket(self._h))

    def last_ticket(self):
        """Ticket of the most recent extract_batch_device call on this handle (spfe_last_ticket)."""
        return int(self._lib.spfe_last_ticket(self._h))

    def wait_records(self, ticket, stream=None):
        """Order `stream` after the covariance stage of call `ticket` (async_cov mode)."""
        _check(self._lib.spfe_wait_records(self._h, int(ticket), C.c_void_p(stream or 0)))

    # -- multi-GPU: RCCL all-gather of the records, inside the library (no torch) --
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (rank 0 creates it and ships it to every rank)."""
        buf = (C.c_ubyte * 128)()
        _check(load_library().spfe_comm_unique_id(buf, 128))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        _check(self._lib.spfe_comm_init(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def comm_destroy(self):
        _check(self._lib.spfe_comm_destroy(self._h))

    def comm_stream(self):
        """hipStream_t (int) of the library's communication stream, 0 before comm_init."""
        return int(self._lib.spfe_comm_stream(self._h) or 0)

    def comm_count(self):
        """Number of ranks in the communicator as RCCL reports it (ncclCommCount)."""
        n = C.c_int(0)
        _check(self._lib.spfe_comm_count(self._h, C.byref(n)))
        return n.value

    def allgather_records(self, ticket, d_local, d_all, frames_per_rank):
        """ncclAllGather of this rank's `frames_per_rank` records (device pointers as ints) on the library's
        communication stream, ordered after the records of call `ticket`."""
        _check(self._lib.spfe_allgather_records(self._h, int(ticket), C.c_void_p(d_local), C.c_void_p(d_all),
                                                int(frames_per_rank)))

    def comm_wait(self, stream=None):
        """Order `stream` after the last all-gather."""
        _check(self._lib.spfe_comm_wait(self._h, C.c_void_p(stream or 0)))

    def view_record(self, host_record):
        """Decode ONE record (bytes-like / uint8 array copied from the device)."""
        rec = np.ascontiguousarray(np.frombuffer(host_record, np.uint8)
                                   if not isinstance(host_record, np.ndarray) else host_record)
        r = _Result()
        _check(self._lib.spfe_view_record(self._h, rec.ctypes.data, C.byref(r)))
        return FrameResult(r, self.height, self.width, False)

    # -- input staging --
    def set_staging(self, src_height, src_width, channels=3, rgb=False, map_x=None, map_y=None):
        """Configure the raw-frame front end: cv::remap(m1, m2, INTER_LINEAR) (data_loader.cc:519-521),
        crop (system.cpp:160-161), cvtColor to gray (mono_tracker.cpp:18-28).  Maps: f32
        [src_height, src_width] (cv::initUndistortRectifyMap, CV_32FC1) or None."""
        st = _Staging(src_height, src_width, channels, 1 if rgb else 0, None, None)
        keep = []
        if map_x is not None or map_y is not None:
            mx = np.ascontiguousarray(map_x, np.float32)
            my = np.ascontiguousarray(map_y, np.float32)
            if mx.shape != (src_height, src_width) or my.shape != (src_height, src_width):
                raise SpfeError("maps must be [src_height, src_width]")
            keep = [mx, my]
            st.map_x, st.map_y = mx.ctypes.data, my.ctypes.data
        _check(self._lib.spfe_set_staging(self._h, C.byref(st)))
        self._staging = (src_height, src_width, channels)
        del keep

    def _check_raw(self, src):
        if src is None or getattr(src, "size", 0) == 0:
            raise RuntimeError("input image is empty")  # sp_extractor.cpp:364-365
        hs, ws, cn = self._staging
        src = np.asarray(src)
        want = (hs, ws) if cn == 1 else (hs, ws, cn)
        if src.dtype != np.uint8 or src.shape != want:
            raise SpfeError("raw frame must be uint8 %s, got %s %s" % (want, src.dtype, src.shape))
        return np.ascontiguousarray(src)

    def extract_staged(self, src):
        """Raw camera frame -> Frame (remap + crop + gray on the GPU, then the extraction path)."""
        return self.extract_batch_staged([src])[0]

    def extract_batch_staged(self, srcs):
        srcs = [self._check_raw(s) for s in srcs]
        n = len(srcs)
        hs, ws, cn = self._staging
        ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in srcs])
        res = (_Result * n)()
        _check(self._lib.spfe_extract_batch_staged(self._h, ptrs, ws * cn, n, res))
        frames = [FrameResult(res[i], self.height, self.width, self.with_heat) for i in range(n)]
        self._publish(frames[-1])
        return frames

    def stage_batch_device(self, d_src, n, d_gray, stream=None):
        _check(self._lib.spfe_stage_batch_device(self._h, d_src, n, d_gray, stream))

    # -- descriptor matching --
    def match(self, query, train, cross_check=True):
        """cv::BFMatcher(NORM_L2, crossCheck).match(query) with `train` added, as
        SPMatcher::SearchByBruteForce calls it (sp_matcher.cpp:1642-1674).
        query [nq,256], train [nt,256] f32 -> (train_idx int32 [nq], -1 = no match; distance f32 [nq])."""
        q = np.ascontiguousarray(query, np.float32).reshape(-1, 256)
        t = np.ascontiguousarray(train, np.float32).reshape(-1, 256)
        idx = np.empty(len(q), np.int32)
        dist = np.empty(len(q), np.float32)
        _check(self._lib.spfe_match(self._h, q.ctypes.data if len(q) else None, len(q),
                                    t.ctypes.data if len(t) else None, len(t), 1 if cross_check else 0,
                                    idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def match_knn2(self, query, train):
        """knnMatch(query, matches, 2) against `train`, exact: -> (train_idx [nq, 2] int32, distance [nq, 2] f32)."""
        q = np.ascontiguousarray(query, np.float32).reshape(-1, 256)
        t = np.ascontiguousarray(train, np.float32).reshape(-1, 256)
        idx = np.full((max(len(q), 1), 2), -1, np.int32)
        dist = np.zeros((max(len(q), 1), 2), np.float32)
        _check(self._lib.spfe_match_knn2(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data,
                                         dist.ctypes.data))
        return idx[:len(q)], dist[:len(q)]

    def match_patches(self, mp_desc, mp_uv, occ_grid, kp_desc, max_dist=0.75):
        """Patch-wise association of projected map points (tracker_dust.cpp:113-172): map point i at
        dust-map position mp_uv[i] (cells) takes the nearest keypoint of its 2 x 2 cells below max_dist,
        earlier map points first.  -> int32 [n_points] keypoint index or -1."""
        m = np.ascontiguousarray(mp_desc, np.float32).reshape(-1, 256)
        uv = np.ascontiguousarray(mp_uv, np.float32).reshape(-1, 2)
        occ = np.ascontiguousarray(occ_grid, np.int16)
        kd = np.ascontiguousarray(kp_desc, np.float32).reshape(-1, 256)
        if occ.shape != (self.height // 8, self.width // 8):
            raise SpfeError("occ_grid must be [height/8, width/8]")
        out = np.empty(len(m), np.int32)
        _check(self._lib.spfe_match_patches(self._h, m.ctypes.data if len(m) else None, uv.ctypes.data if len(m) else None,
                                            len(m), occ.ctypes.data, kd.ctypes.data if len(kd) else None, len(kd),
                                            max_dist, out.ctypes.data if len(m) else np.empty(1, np.int32).ctypes.data))
        return out

    def match_patches_record_device(self, d_mp_desc, d_mp_uv, n_points, d_record, d_kp_idx, max_dist=0.75,
                                    stream=None):
        _check(self._lib.spfe_match_patches_record_device(self._h, d_mp_desc, d_mp_uv, n_points, d_record, max_dist,
                                                          d_kp_idx, stream))

    def match_out_bytes(self):
        return int(self._lib.spfe_match_out_bytes(self._h))

    def match_records_device(self, d_query_records, d_train_records, n_pairs, d_out, cross_check=True,
                             stream=None):
        """Device pointers (ints) to n_pairs query / train records and to n_pairs * match_out_bytes()
        bytes of output; enqueues on `stream` (int hipStream_t, None = the handle's stream)."""
        _check(self._lib.spfe_match_records_device(self._h, d_query_records, d_train_records, n_pairs,
                                                   1 if cross_check else 0, d_out, stream))

    def decode_match_out(self, host_block, n_query=None):
        """uint8[match_out_bytes()] copied from the device -> (train_idx, distance)."""
        b = np.ascontiguousarray(host_block, np.uint8)
        kmax = self.match_out_bytes() // 8
        idx = b[:kmax * 4].view(np.int32)
        dist = b[kmax * 4:kmax * 8].view(np.float32)
        n = kmax if n_query is None else n_query
        return idx[:n].copy(), dist[:n].copy()

    def fetch_heat_inv(self, frame=0):
        """heat_inv (sp_extractor.cpp:468) of a frame of the last synchronous host call, copied back on demand
        (spfe_fetch_heat_inv; the companion of lazy_heat_inv=True)."""
        p = C.c_void_p()
        self._lib.spfe_fetch_heat_inv.restype = C.c_int
        self._lib.spfe_fetch_heat_inv.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        _check(self._lib.spfe_fetch_heat_inv(self._h, int(frame), C.byref(p)))
        return _as_np(p.value, (self.height, self.width), np.float32)

    def debug_read(self, name, frame=0):
        shapes = {"semi": (self.height // 8, self.width // 8, 65),
                  "coarse": (self.height // 8, self.width // 8, 256),
                  "head": (self.height // 8, self.width // 8, 512),
                  "feat": (self.height // 8, self.width // 8, 128),
                  "heat_log": (self.height, self.width), "heat_inv": (self.height, self.width),
                  "heat": (self.height, self.width),
                  "cell_score": (self.height // 8, self.width // 8)}
        div = [1, 2, 2, 4, 4, 8, 8, 8]
        ch = [64, 64, 64, 64, 128, 128, 128, 128]
        for i in range(8):
            shapes["act%d" % i] = (self.height // div[i], self.width // div[i], ch[i])
        shapes["coarse_sparse"] = shapes["coarse"]   # the descriptor map as the last call left it (only the rows it read)
        if name.startswith("cov_"):   # covariance scratch (int32): counters [4], nxt / workers / npop [kmax]
            out = np.empty(4 if name == "cov_counters" else self.nfeatures + 1, np.int32)
        elif name in ("db_total", "da_gathered", "conv1b_tile_rows", "conv1b_split_rows", "select_huge", "two_chains", "twin_failed"):   # gathered descriptor head: number of listed cells of the last call /
            out = np.empty(1, np.int32)             # whether convDa ran on those cells only
        elif name == "db_list":       # ... and the list (global cell indices b * C + cell), max_batch * min(4 kmax, C) entries
            C_ = (self.height // 8) * (self.width // 8)
            out = np.empty(self.max_batch * min(4 * (self.nfeatures + 1), C_), np.int32)
        else:
            out = (np.empty((self.height, self.width), np.uint8) if name == "image"   # the staged gray frame
                   else np.empty(shapes[name], np.float32))
        n = self._lib.spfe_debug_read(self._h, name.encode(), frame, out.ctypes.data, out.nbytes)
        if n < 0:
            _check(int(n))
        return out

    def stage_reset(self):
        _check(self._lib.spfe_stage_reset(self._h))

    def stage_times(self):
        buf = (C.c_float * 32)()
        n = self._lib.spfe_stage_times(self._h, buf, 32)
        if n < 0:
            _check(n)
        return {self._lib.spfe_stage_name(i).decode(): buf[i] for i in range(n)}


def math_probe(x):
    """Device spfe_expf(x), spfe_logf(|x|) for a float32 array (test hook)."""
    x = np.ascontiguousarray(x, np.float32)
    e = np.empty_like(x)
    l = np.empty_like(x)
    _check(load_library().spfe_math_probe(x.ctypes.data, e.ctypes.data, l.ctypes.data, x.size))
    return e, l
