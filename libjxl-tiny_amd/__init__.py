"""libjxl-tiny_amd -- MI355X-native JPEG XL "tiny" encoder hot path.

Thin ctypes binding of the two native libraries (see include/jxl_tiny_amd.h):

* ``csrc/libjxltiny_hip.so``  hand-written HIP kernels for gfx950 + device C ABI
* ``host/libjxltiny_host.so`` C++ drop-in EncodeFile/EncodeFrame + bitstream back-end

The package directory name contains a hyphen; import it with
``importlib.util.spec_from_file_location("libjxl_tiny_amd", ".../libjxl-tiny_amd/__init__.py")``
(``__graft_entry__.load_package()`` does that).

There is no CPU implementation in this package: every compute entry point needs
a HIP device and raises ``JxlTinyError`` otherwise.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
HIP_LIB = PKG_DIR / "csrc" / "libjxltiny_hip.so"
HOST_LIB = PKG_DIR / "host" / "libjxltiny_host.so"
CJXL_TINY = PKG_DIR / "host" / "cjxl_tiny"

fp = C.POINTER(C.c_float)
FLAG_FORCE_DCT8, FLAG_DEBUG_DUMP, FLAG_PROFILE = 1, 2, 4

# Symbols include/jxl_tiny_amd.h declares, per library (checked by the tests); *_TESTING: the ones of
# include/jxl_tiny_amd_testing.h (test-suite and profiling tools only).
HIP_SYMBOLS = ["jxlt_context_create", "jxlt_context_destroy", "jxlt_last_error", "jxlt_context_device",
               "jxlt_device_count", "jxlt_bind_thread_near_device",
               "jxlt_image_upload", "jxlt_image_set_device", "jxlt_image_upload_pfm", "jxlt_image_set_device_pfm", "jxlt_image_attach_host", "jxlt_image_attach_host_pfm", "jxlt_image_size", "jxlt_pinned_alloc",
               "jxlt_pinned_free", "jxlt_pinned_register", "jxlt_pinned_unregister", "jxlt_encode_enqueue", "jxlt_set_strategy_distance", "jxlt_context_set_wait_mode",
               "jxlt_fetch_histograms", "jxlt_fetch_dc_histogram", "jxlt_histograms_ready",
               "jxlt_pack_begin", "jxlt_pack_sizes", "jxlt_pack_deliver", "jxlt_release_cached_memory",
               "jxlt_output_buffer",
               "jxlt_synchronize", "jxlt_fetch_result", "jxlt_encode_stats"]
HIP_SYMBOLS_TESTING = ["jxlt_debug_fetch", "jxlt_fetch_side_info", "jxlt_pack_sections", "jxlt_kernel_times",
                       "jxlt_debug_denormal_probe"]
HOST_SYMBOLS = ["jxlt_compute_distance_params", "jxlt_assemble_frame",
                "jxlt_encode_file_planar", "jxlt_encode_file_planar_devices", "jxlt_encode_pfm_file", "jxlt_emulate_reference_static_constants", "jxlt_emulate_reference_single_symbol_codes", "jxlt_encode_resident", "jxlt_encode_resident_view", "jxlt_write_file_header", "jxlt_last_frame_timeline",
                "jxlt_free", "jxlt_batch_encoder_create", "jxlt_batch_encoder_create_multi",
                "jxlt_batch_encoder_destroy", "jxlt_batch_encoder_run",
                "jxlt_shard_rect", "jxlt_multi_encoder_create", "jxlt_multi_encoder_destroy",
                "jxlt_multi_encoder_last_error", "jxlt_multi_encoder_encode", "jxlt_multi_encoder_encode_pfm",
                "jxlt_multi_encoder_set_device_slab", "jxlt_multi_encoder_encode_resident",
                "jxlt_shard_group_open", "jxlt_shard_group_close", "jxlt_shard_group_last_error",
                "jxlt_shard_encode", "jxlt_shard_pipeline_open", "jxlt_shard_pipeline_close",
                "jxlt_shard_pipeline_last_error", "jxlt_shard_pipeline_submit_device", "jxlt_shard_pipeline_wait"]
HOST_SYMBOLS_TESTING = ["jxlt_debug_dc_records", "jxlt_assemble_frame_groups", "jxlt_build_code_tables", "jxlt_finish_frame",
                        "jxlt_shard_group_last_timeline",
                        "jxlt_shard_encode_ops", "jxlt_shard_pipeline_open_ops",
                        "jxlt_shard_pipeline_submit_ops"]


class JxlTinyError(RuntimeError):
    pass


class DistanceParams(C.Structure):
    _fields_ = [("distance", C.c_float), ("global_scale", C.c_int32), ("quant_dc", C.c_int32),
                ("scale", C.c_float), ("inv_scale", C.c_float), ("scale_dc", C.c_float),
                ("x_qm_scale", C.c_uint32), ("epf_iters", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("distance", C.c_float), ("scale", C.c_float), ("inv_scale", C.c_float),
                ("scale_dc", C.c_float), ("x_qm_scale", C.c_uint32), ("flags", C.c_uint32)]


class FrameResult(C.Structure):
    _fields_ = [("xsize", C.c_size_t), ("ysize", C.c_size_t),
                ("xsize_blocks", C.c_size_t), ("ysize_blocks", C.c_size_t),
                ("xsize_tiles", C.c_size_t), ("ysize_tiles", C.c_size_t),
                ("num_groups", C.c_size_t),
                ("quant_dc", C.POINTER(C.c_int16) * 3),
                ("raw_quant_field", C.POINTER(C.c_uint8)),
                ("ac_strategy", C.POINTER(C.c_uint8)),
                ("ytox_map", C.POINTER(C.c_int8)), ("ytob_map", C.POINTER(C.c_int8)),
                ("tokens", C.POINTER(C.c_uint8)),
                ("group_token_offset", C.POINTER(C.c_uint64))]


class PackedSections(C.Structure):
    _fields_ = [("bytes", C.POINTER(C.c_uint8)), ("section_offset", C.POINTER(C.c_uint64)),
                ("section_bits", C.POINTER(C.c_uint32)), ("num_sections", C.c_size_t)]


class EncodeStats(C.Structure):
    _fields_ = [("tiles", C.c_uint32), ("tiles_redone_exact_roots", C.c_uint32),
                ("encodes_with_redone_tiles", C.c_uint32), ("copy_calls", C.c_uint32),
                ("longest_copy_call_us", C.c_float)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("milliseconds", C.c_float)]


def build(verbose=False):
    """Compiles both native libraries and cjxl_tiny in-tree (hipcc, gfx950)."""
    out = subprocess.run(["make", "-C", str(PKG_DIR)], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise JxlTinyError("native build failed")


_hip = None
_host = None


def hip_lib():
    global _hip
    if _hip is None:
        if not HIP_LIB.exists():
            raise JxlTinyError("%s is missing: run __graft_entry__.build()" % HIP_LIB)
        try:
            # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not
            # coexist, so when torch is installed let it load its runtime first and share it.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(str(HIP_LIB))
        L.jxlt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.jxlt_context_destroy.argtypes = [C.c_void_p]
        L.jxlt_context_destroy.restype = None
        L.jxlt_release_cached_memory.argtypes = [C.c_int]
        L.jxlt_release_cached_memory.restype = C.c_size_t
        L.jxlt_last_error.argtypes = [C.c_void_p]
        L.jxlt_last_error.restype = C.c_char_p
        L.jxlt_image_upload.argtypes = [C.c_void_p, C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t]
        L.jxlt_image_set_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t,
                                            C.c_size_t]
        L.jxlt_encode_enqueue.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.jxlt_synchronize.argtypes = [C.c_void_p]
        L.jxlt_fetch_result.argtypes = [C.c_void_p, C.POINTER(FrameResult)]
        L.jxlt_fetch_histograms.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)),
                                            C.POINTER(C.POINTER(C.c_uint32))]
        L.jxlt_pack_sections.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(PackedSections)]
        L.jxlt_kernel_times.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_int]
        L.jxlt_encode_stats.argtypes = [C.c_void_p, C.POINTER(EncodeStats)]
        L.jxlt_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not HOST_LIB.exists():
            raise JxlTinyError("%s is missing: run __graft_entry__.build()" % HOST_LIB)
        L = C.CDLL(str(HOST_LIB))
        L.jxlt_compute_distance_params.argtypes = [C.c_float, C.POINTER(DistanceParams)]
        L.jxlt_compute_distance_params.restype = None
        L.jxlt_assemble_frame.argtypes = [C.POINTER(FrameResult), C.POINTER(DistanceParams), C.c_int,
                                          C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_assemble_frame_groups.argtypes = [C.POINTER(FrameResult), C.POINTER(C.POINTER(C.c_uint8)),
                                                 C.POINTER(C.c_size_t), C.POINTER(DistanceParams), C.c_int,
                                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_encode_file_planar.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                              C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_encode_file_planar_devices.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                                      C.POINTER(C.c_int), C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                                      C.POINTER(C.c_size_t)]
        L.jxlt_encode_resident.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                           C.POINTER(C.c_size_t)]
        L.jxlt_encode_resident_view.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                                C.POINTER(C.c_size_t)]
        L.jxlt_write_file_header.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)),
                                             C.POINTER(C.c_size_t)]
        L.jxlt_build_code_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.jxlt_finish_frame.argtypes = [C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p,
                                        C.POINTER(PackedSections), C.POINTER(PackedSections),
                                        C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_debug_dc_records.argtypes = [C.POINTER(FrameResult), C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)),
                                            C.POINTER(C.c_size_t)]
        L.jxlt_free.argtypes = [C.c_void_p]
        L.jxlt_free.restype = None
        _host = L
    return _host


def distance_params(distance):
    """DistanceParams of the reference (enc_frame.cc:115-156), after EncodeFile's clamp."""
    if not distance > 0:
        raise JxlTinyError("distance must be > 0")
    if distance <= 0.03:
        distance = 0.03
    p = DistanceParams()
    host_lib().jxlt_compute_distance_params(C.c_float(distance), C.byref(p))
    return p


def _take_bytes(ptr, n):
    data = C.string_at(ptr, n.value)
    host_lib().jxlt_free(ptr)
    return data


class NativeView:
    """A codestream in a buffer owned by the native library (not freed from Python)."""

    def __init__(self, ptr, size):
        self.ptr, self.size = ptr, size

    def __len__(self):
        return self.size

    def tobytes(self):
        return C.string_at(self.ptr, self.size)


def file_header(xsize, ysize):
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    if host_lib().jxlt_write_file_header(xsize, ysize, C.byref(out), C.byref(n)) != 0:
        raise JxlTinyError("invalid image size")
    return _take_bytes(out, n)


class HotPathOutput:
    """numpy copy of one pass of the device hot path."""

    def __init__(self, fr, debug=None):
        yb, xb, yt, xt, ng = fr.ysize_blocks, fr.xsize_blocks, fr.ysize_tiles, fr.xsize_tiles, fr.num_groups

        def arr(ptr, shape, dtype):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()

        self.xsize, self.ysize = fr.xsize, fr.ysize
        self.quant_dc = np.stack([arr(fr.quant_dc[c], (yb, xb), np.int16) for c in range(3)])
        self.raw_quant = arr(fr.raw_quant_field, (yb, xb), np.uint8)
        self.strategy = arr(fr.ac_strategy, (yb, xb), np.uint8)
        self.ytox = arr(fr.ytox_map, (yt, xt), np.int8)
        self.ytob = arr(fr.ytob_map, (yt, xt), np.int8)
        offs = [fr.group_token_offset[g] for g in range(ng + 1)]
        blob = C.string_at(fr.tokens, offs[ng]) if offs[ng] else b""
        self.group_tokens = [blob[offs[g]:offs[g + 1]] for g in range(ng)]
        self.xyb = self.qf = self.mask = self.ent8 = None
        if debug:
            self.xyb, self.qf, self.mask, self.ent8 = debug

    def all_tokens(self):
        return b"".join(self.group_tokens)


class Encoder:
    """One device context (one per GPU / host thread)."""

    def __init__(self, device=0):
        self._L = hip_lib()
        self._ctx = C.c_void_p()
        rc = self._L.jxlt_context_create(device, C.byref(self._ctx))
        if rc != 0:
            raise JxlTinyError("jxlt_context_create failed (%d): %s" %
                               (rc, self._L.jxlt_last_error(None).decode()))
        self.device = device
        self._keepalive = None

    def close(self):
        if self._ctx:
            self._L.jxlt_context_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise JxlTinyError("%s failed (%d): %s" % (what, rc, self._L.jxlt_last_error(self._ctx).decode()))
        return rc

    def set_wait_mode(self, mode):
        """0: the calls wait for the device spinning / sleeping through long waits (lowest latency of one frame);
        1: throughput mode, what the lanes of a batch use -- short sleeps, and the frame's small launches shared where
        they can be (jxlt_context_set_wait_mode)."""
        self._L.jxlt_context_set_wait_mode.argtypes = [C.c_void_p, C.c_int]
        self._L.jxlt_context_set_wait_mode.restype = C.c_int
        self._check(self._L.jxlt_context_set_wait_mode(self._ctx, int(mode)), "jxlt_context_set_wait_mode")

    def upload(self, planes):
        """planes: float32 [3, h, w] C-contiguous host array."""
        assert planes.dtype == np.float32 and planes.ndim == 3 and planes.shape[0] == 3
        planes = np.ascontiguousarray(planes)
        _, h, w = planes.shape
        ptrs = (fp * 3)(*[planes[c].ctypes.data_as(fp) for c in range(3)])
        self._check(self._L.jxlt_image_upload(self._ctx, ptrs, w * 4, w, h), "jxlt_image_upload")

    def attach_host(self, planes):
        """planes: float32 [3, h, w] in page-locked memory (pinned_empty): stays there until the next encode,
        whose kernels run under the row-wise upload (jxlt_image_attach_host)."""
        assert planes.dtype == np.float32 and planes.ndim == 3 and planes.shape[0] == 3 and planes.strides[2] == 4
        _, h, w = planes.shape
        ptrs = (fp * 3)(*[planes[c].ctypes.data_as(fp) for c in range(3)])
        self._L.jxlt_image_attach_host.argtypes = [C.c_void_p, C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t]
        self._check(self._L.jxlt_image_attach_host(self._ctx, ptrs, planes.strides[1], w, h), "jxlt_image_attach_host")
        self._keepalive = planes

    def attach_host_pfm(self, payload, w, h, big_endian=False):
        """payload: numpy array over the raw PFM sample payload in page-locked memory (jxlt_image_attach_host_pfm)."""
        self._L.jxlt_image_attach_host_pfm.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        self._check(self._L.jxlt_image_attach_host_pfm(self._ctx, payload.ctypes.data, w, h, 1 if big_endian else 0),
                    "jxlt_image_attach_host_pfm")
        self._keepalive = payload

    def set_device_image(self, ptrs, pitch_bytes, w, h, keepalive=None):
        """ptrs: three device addresses (ints), e.g. torch tensor data_ptr()."""
        arr = (C.c_void_p * 3)(*ptrs)
        self._check(self._L.jxlt_image_set_device(self._ctx, arr, pitch_bytes, w, h), "jxlt_image_set_device")
        self._keepalive = keepalive

    def set_strategy_distance(self, first_call_distance):
        """Reproduce a later EncodeFile call of a reference process whose first call used this
        distance (jxlt_set_strategy_distance); 0 switches it off."""
        self._L.jxlt_set_strategy_distance.argtypes = [C.c_void_p, C.c_float]
        self._check(self._L.jxlt_set_strategy_distance(self._ctx, C.c_float(first_call_distance)),
                    "jxlt_set_strategy_distance")

    def enqueue(self, distance, flags=0):
        dp = distance_params(distance)
        p = Params(dp.distance, dp.scale, dp.inv_scale, dp.scale_dc, dp.x_qm_scale, flags)
        self._check(self._L.jxlt_encode_enqueue(self._ctx, C.byref(p)), "jxlt_encode_enqueue")
        self._last = (dp, flags)
        return dp

    def synchronize(self):
        self._check(self._L.jxlt_synchronize(self._ctx), "jxlt_synchronize")

    def fetch_raw(self):
        fr = FrameResult()
        self._check(self._L.jxlt_fetch_result(self._ctx, C.byref(fr)), "jxlt_fetch_result")
        return fr

    def fetch_histograms(self):
        """(ac, dc) symbol histograms [64, 64] uint32 of the last enqueue (jxlt_fetch_histograms)."""
        a, d = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        self._check(self._L.jxlt_fetch_histograms(self._ctx, C.byref(a), C.byref(d)), "jxlt_fetch_histograms")
        cp = lambda p: np.ctypeslib.as_array(p, shape=(4096,)).reshape(64, 64).copy()
        return cp(a), cp(d)

    def pack_sections(self, kind, table):
        """Device-side packing of the DC-group (kind 0) or AC-group (kind 1) sections with the code
        table [4096] uint32; returns (bytes uint8[], offsets uint64[n+1], bits uint32[n]) copies."""
        table = np.ascontiguousarray(table, np.uint32).reshape(-1)
        ps = PackedSections()
        self._check(self._L.jxlt_pack_sections(self._ctx, kind, table.ctypes.data, C.byref(ps)), "jxlt_pack_sections")
        n = ps.num_sections
        off = np.ctypeslib.as_array(ps.section_offset, shape=(n + 1,)).copy()
        bits = np.ctypeslib.as_array(ps.section_bits, shape=(n,)).copy()
        total = int(off[n])
        data = np.ctypeslib.as_array(ps.bytes, shape=(max(total, 1),))[:total].copy()
        return data, off, bits

    def stats(self):
        """Statistics of the last encode (jxlt_encode_stats): tiles, tiles redone with computed square roots,
        encodes of this context in which any tile was."""
        st = EncodeStats()
        self._check(self._L.jxlt_encode_stats(self._ctx, C.byref(st)), "jxlt_encode_stats")
        return {"tiles": int(st.tiles), "tiles_redone_exact_roots": int(st.tiles_redone_exact_roots),
                "encodes_with_redone_tiles": int(st.encodes_with_redone_tiles), "copy_calls": int(st.copy_calls),
                "longest_copy_call_us": float(st.longest_copy_call_us)}

    def kernel_times(self):
        arr = (KernelTime * 8)()
        n = self._check(self._L.jxlt_kernel_times(self._ctx, arr, 8), "jxlt_kernel_times")
        return {arr[i].name.decode(): arr[i].milliseconds for i in range(n)}

    def hot_path(self, planes, distance, force_dct8=False, debug=False):
        """Upload + one pass of the hot path + fetch; returns HotPathOutput."""
        self.upload(planes)
        flags = (FLAG_FORCE_DCT8 if force_dct8 else 0) | (FLAG_DEBUG_DUMP if debug else 0)
        self.enqueue(distance, flags)
        fr = self.fetch_raw()
        dbg = None
        if debug:
            yb, xb = fr.ysize_blocks, fr.xsize_blocks

            def get(what, shape):
                a = np.empty(shape, np.float32)
                self._check(self._L.jxlt_debug_fetch(self._ctx, what, a.ctypes.data, a.nbytes), "jxlt_debug_fetch")
                return a

            xyb = np.stack([get(c, (yb * 8, xb * 8)) for c in range(3)])
            dbg = (xyb, get(3, (yb, xb)), get(4, (yb, xb)), get(5, (yb // 2 + 1, xb // 2 + 1, 8)))
        return HotPathOutput(fr, dbg)

    def assemble(self, fr, dp, num_threads=0):
        """Frame bitstream (header + TOC + sections) from a fetched FrameResult."""
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = host_lib().jxlt_assemble_frame(C.byref(fr), C.byref(dp), num_threads, C.byref(out), C.byref(n))
        if rc != 0:
            raise JxlTinyError("jxlt_assemble_frame failed (%d)" % rc)
        return _take_bytes(out, n)

    def encode_resident(self, distance, num_threads=0, copy=True):
        """Full codestream of the image currently set/uploaded on the device (production path:
        device pipeline + device section packing + host assembly).
        copy=True  -> Python bytes (jxlt_encode_resident);
        copy=False -> NativeView on the context's page-locked output buffer, valid until the next
                      encode on this Encoder (jxlt_encode_resident_view: no host-side copy of the
                      packed sections at all)."""
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        fn = host_lib().jxlt_encode_resident if copy else host_lib().jxlt_encode_resident_view
        rc = fn(self._ctx, C.c_float(distance), num_threads, C.byref(out), C.byref(n))
        if rc != 0:
            raise JxlTinyError("jxlt_encode_resident failed (%d): %s" %
                               (rc, self._L.jxlt_last_error(self._ctx).decode()))
        return _take_bytes(out, n) if copy else NativeView(out, n.value)

    def encode_resident_raw_tokens(self, distance, num_threads=0, flags=0):
        """Same result through the raw-token route (tokens copied to the host and packed there)."""
        dp = self.enqueue(distance, flags)
        fr = self.fetch_raw()
        return file_header(fr.xsize, fr.ysize) + self.assemble(fr, dp, num_threads)


def release_cached_memory(device=-1):
    """Device memory that destroyed contexts left for their successors goes back to the HIP runtime
    (jxlt_release_cached_memory); returns the number of bytes released."""
    return int(hip_lib().jxlt_release_cached_memory(device))


def build_code_tables(ac_hist, dc_hist):
    """Prefix-code tables [4096] uint32 ((depth << 16) | bits) from (summed) histograms."""
    ac_hist = np.ascontiguousarray(ac_hist, np.uint32)
    dc_hist = np.ascontiguousarray(dc_hist, np.uint32)
    ac_t, dc_t = np.zeros(4096, np.uint32), np.zeros(4096, np.uint32)
    rc = host_lib().jxlt_build_code_tables(ac_hist.ctypes.data, dc_hist.ctypes.data, ac_t.ctypes.data,
                                           dc_t.ctypes.data)
    if rc != 0:
        raise JxlTinyError("jxlt_build_code_tables failed (%d)" % rc)
    return ac_t, dc_t


def finish_frame(xsize, ysize, distance, ac_hist, dc_hist, dc_sections, ac_sections):
    """Codestream of the whole frame from summed histograms and all packed sections in raster
    order; *_sections = (bytes uint8[], offsets uint64[n+1], bits uint32[n])."""
    keep = []

    def mk(sec):
        data, off, bits = (np.ascontiguousarray(sec[0], np.uint8), np.ascontiguousarray(sec[1], np.uint64),
                           np.ascontiguousarray(sec[2], np.uint32))
        if data.size == 0:
            data = np.zeros(1, np.uint8)
        keep.extend([data, off, bits])
        ps = PackedSections()
        ps.bytes = data.ctypes.data_as(C.POINTER(C.c_uint8))
        ps.section_offset = off.ctypes.data_as(C.POINTER(C.c_uint64))
        ps.section_bits = bits.ctypes.data_as(C.POINTER(C.c_uint32))
        ps.num_sections = len(bits)
        return ps

    ac_hist = np.ascontiguousarray(ac_hist, np.uint32)
    dc_hist = np.ascontiguousarray(dc_hist, np.uint32)
    dcs, acs = mk(dc_sections), mk(ac_sections)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = host_lib().jxlt_finish_frame(xsize, ysize, C.c_float(distance), ac_hist.ctypes.data, dc_hist.ctypes.data,
                                      C.byref(dcs), C.byref(acs), C.byref(out), C.byref(n))
    if rc != 0:
        raise JxlTinyError("jxlt_finish_frame failed (%d)" % rc)
    return _take_bytes(out, n)


def emulate_reference_static_constants(on):
    """jxl::EmulateReferenceStaticConstants: latch the process's first distance for the transform
    search's multipliers like the reference library (enc_ac_strategy.cc:178-185)."""
    host_lib().jxlt_emulate_reference_static_constants(1 if on else 0)


def emulate_reference_single_symbol_codes(on):
    """jxl::EmulateReferenceSingleSymbolCodes: write one bit per token of a single-symbol prefix code like
    the reference does (undecodable output in those cases) instead of the conformant zero bits."""
    host_lib().jxlt_emulate_reference_single_symbol_codes(1 if on else 0)


def encode_pfm_file(path, distance, device=0):
    """cjxl_tiny in one call: PFM file -> .jxl bytes, PFM payload ingested by the device kernels."""
    L = host_lib()
    L.jxlt_encode_pfm_file.argtypes = [C.c_char_p, C.c_float, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                       C.POINTER(C.c_size_t)]
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = L.jxlt_encode_pfm_file(str(path).encode(), C.c_float(distance), device, C.byref(out), C.byref(n))
    if rc != 0:
        raise JxlTinyError("jxlt_encode_pfm_file failed (%d)" % rc)
    return _take_bytes(out, n)


def encode_file(planes, distance, device=0):
    """Drop-in EncodeFile through the C++ host library: float32 [3,h,w] -> .jxl bytes."""
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    _, h, w = planes.shape
    ptrs = (fp * 3)(*[planes[c].ctypes.data_as(fp) for c in range(3)])
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = host_lib().jxlt_encode_file_planar(ptrs, w * 4, w, h, C.c_float(distance), device, C.byref(out),
                                            C.byref(n))
    if rc != 0:
        raise JxlTinyError("jxlt_encode_file_planar failed (%d)" % rc)
    return _take_bytes(out, n)


class FrameTimeline(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("dc_histogram_ms", "ac_histogram_ms", "codes_ms", "sizes_ms", "done_ms")]


def last_frame_timeline():
    """jxlt_last_frame_timeline: host-side stage times (ms from the call's start) of this thread's last frame."""
    tl = FrameTimeline()
    L = host_lib()
    L.jxlt_last_frame_timeline.argtypes = [C.POINTER(FrameTimeline)]
    if L.jxlt_last_frame_timeline(C.byref(tl)) != 0:
        return None
    return {k: getattr(tl, k) for k, _ in FrameTimeline._fields_}


def encode_file_devices(planes, distance, devices):
    """jxlt_encode_file_planar_devices: the drop-in's EncodeFile over the calling thread's device list."""
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    _, h, w = planes.shape
    ptrs = (fp * 3)(*[planes[c].ctypes.data_as(fp) for c in range(3)])
    devs = (C.c_int * len(devices))(*devices)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = host_lib().jxlt_encode_file_planar_devices(ptrs, w * 4, w, h, C.c_float(distance), devs, len(devices),
                                                    C.byref(out), C.byref(n))
    if rc != 0:
        raise JxlTinyError("jxlt_encode_file_planar_devices failed (%d)" % rc)
    return _take_bytes(out, n)


class BatchFrame(C.Structure):
    _fields_ = [("planes", fp * 3), ("pitch_bytes", C.c_size_t), ("pfm_payload", C.c_void_p),
                ("pfm_big_endian", C.c_int), ("xsize", C.c_size_t), ("ysize", C.c_size_t),
                ("in_device_memory", C.c_int), ("device_ordinal", C.c_int)]


def pinned_empty(shape, dtype=np.float32):
    """numpy array in page-locked host memory (jxlt_pinned_alloc); keep the returned owner alive."""
    L = hip_lib()
    L.jxlt_pinned_alloc.argtypes = [C.c_size_t]
    L.jxlt_pinned_alloc.restype = C.c_void_p
    L.jxlt_pinned_free.argtypes = [C.c_void_p]
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = L.jxlt_pinned_alloc(n)
    if not p:
        raise JxlTinyError("jxlt_pinned_alloc failed (no usable HIP device)")
    buf = (C.c_uint8 * n).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)

    class _Owner:
        def __del__(self, p=p, free=L.jxlt_pinned_free):
            free(p)
    return arr, _Owner()


class BatchEncoder:
    """jxlt_batch_encoder_*: a batch of independent frames on one GPU through several device contexts
    (uploads, kernels and downloads of different frames overlap).  BASELINE config #5."""

    def __init__(self, device=0, lanes=3, devices=None):
        """devices: list of GPU ordinals for a multi-device encoder (`lanes` contexts on each, one frame queue:
        jxlt_batch_encoder_create_multi); default: `lanes` contexts on `device`."""
        self._L = host_lib()
        L = self._L
        L.jxlt_batch_encoder_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.jxlt_batch_encoder_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.jxlt_batch_encoder_destroy.argtypes = [C.c_void_p]
        L.jxlt_batch_encoder_destroy.restype = None
        L.jxlt_batch_encoder_run.argtypes = [C.c_void_p, C.POINTER(BatchFrame), C.c_size_t, C.c_float,
                                             C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        self._enc = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = L.jxlt_batch_encoder_create_multi(arr, len(devices), lanes, C.byref(self._enc))
        else:
            rc = L.jxlt_batch_encoder_create(device, lanes, C.byref(self._enc))
        if rc != 0:
            raise JxlTinyError("jxlt_batch_encoder_create failed (%d): %s" %
                               (rc, hip_lib().jxlt_last_error(None).decode()))

    def close(self):
        if self._enc:
            self._L.jxlt_batch_encoder_destroy(self._enc)
            self._enc = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self, frames):
        """frames: float32 [3,h,w] arrays (planar), or (payload_bytes_or_array, w, h, big_endian) tuples
        (raw PFM payloads).  Returns (ctypes array, keepalive)."""
        descs = (BatchFrame * len(frames))()
        keep = []
        for d, f in zip(descs, frames):
            if hasattr(f, "data_ptr"):  # a torch tensor [3, h, w] float32 on the encoder's GPU: read in place
                assert f.dim() == 3 and f.shape[0] == 3 and f.stride(2) == 1
                keep.append(f)
                for c in range(3):
                    d.planes[c] = C.cast(C.c_void_p(f[c].data_ptr()), fp)
                d.pitch_bytes = f.stride(1) * 4
                d.xsize, d.ysize = f.shape[2], f.shape[1]
                d.in_device_memory = 1
                d.device_ordinal = f.device.index or 0
                continue
            if isinstance(f, tuple):
                payload, w, h, big = f
                a = np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload
                keep.append(a)
                d.pfm_payload = a.ctypes.data
                d.pfm_big_endian = 1 if big else 0
                d.xsize, d.ysize = w, h
            else:
                assert f.dtype == np.float32 and f.ndim == 3 and f.shape[0] == 3
                assert f.strides[2] == 4 and f.strides[1] % 4 == 0
                keep.append(f)
                for c in range(3):
                    d.planes[c] = f[c].ctypes.data_as(fp)
                d.pitch_bytes = f.strides[1]
                d.xsize, d.ysize = f.shape[2], f.shape[1]
        return descs, keep

    def run_described(self, descs, n, distance, take=True):
        outs = (C.POINTER(C.c_uint8) * n)()
        sizes = (C.c_size_t * n)()
        rc = self._L.jxlt_batch_encoder_run(self._enc, descs, n, C.c_float(distance), outs, sizes)
        if rc != 0:
            for i in range(n):
                if outs[i]:
                    self._L.jxlt_free(outs[i])
            raise JxlTinyError("jxlt_batch_encoder_run failed (%d)" % rc)
        if take:
            res = [C.string_at(outs[i], sizes[i]) for i in range(n)]
            for i in range(n):
                self._L.jxlt_free(outs[i])
            return res
        total = sum(sizes[i] for i in range(n))
        for i in range(n):
            self._L.jxlt_free(outs[i])
        return total

    def encode(self, frames, distance):
        descs, keep = self.describe(frames)
        out = self.run_described(descs, len(frames), distance)
        del keep
        return out


# --------------------------------------------------------------------------- one frame over several GPUs
def bind_thread_near_device(device=0):
    """jxlt_bind_thread_near_device: the calling thread (and the threads it creates from now on) on the CPUs next to
    the GPU.  Returns the previous affinity mask (for os.sched_setaffinity) or None when nothing was changed."""
    import os
    L = hip_lib()
    L.jxlt_bind_thread_near_device.argtypes = [C.c_int]
    try:
        before = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return None
    return before if L.jxlt_bind_thread_near_device(device) == 0 else None


def shard_rect(xsize, ysize, world, rank):
    """(x0, y0, x1, y1): the pixels of participant `rank` of `world` -- a rectangle of whole DC groups, empty when
    there are fewer DC groups than participants (jxlt_shard_rect)."""
    L = host_lib()
    sz = C.POINTER(C.c_size_t)
    L.jxlt_shard_rect.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, sz, sz, sz, sz]
    v = [C.c_size_t() for _ in range(4)]
    if L.jxlt_shard_rect(xsize, ysize, world, rank, *[C.byref(t) for t in v]) != 0:
        raise JxlTinyError("jxlt_shard_rect: invalid arguments")
    return tuple(int(t.value) for t in v)


class MultiEncoder:
    """jxlt_multi_encoder_*: ONE frame over several GPUs of this process (one device context and host thread
    per entry of `devices`; an ordinal may repeat).  BASELINE config #4."""

    def __init__(self, devices):
        self._L = host_lib()
        L = self._L
        L.jxlt_multi_encoder_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.jxlt_multi_encoder_destroy.argtypes = [C.c_void_p]
        L.jxlt_multi_encoder_destroy.restype = None
        L.jxlt_multi_encoder_last_error.argtypes = [C.c_void_p]
        L.jxlt_multi_encoder_last_error.restype = C.c_char_p
        L.jxlt_multi_encoder_encode.argtypes = [C.c_void_p, C.POINTER(fp), C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_multi_encoder_encode_pfm.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_float,
                                                    C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_multi_encoder_set_device_slab.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_size_t,
                                                         C.c_size_t, C.c_size_t]
        L.jxlt_multi_encoder_encode_resident.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_float,
                                                         C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        self.devices = list(devices)
        self._enc = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = L.jxlt_multi_encoder_create(arr, len(self.devices), C.byref(self._enc))
        if rc != 0:
            raise JxlTinyError("jxlt_multi_encoder_create failed (%d): %s" % (rc, hip_lib().jxlt_last_error(None).decode()))
        self._keep = {}

    def close(self):
        if self._enc:
            self._L.jxlt_multi_encoder_destroy(self._enc)
            self._enc = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise JxlTinyError("%s failed (%d): %s" % (what, rc, self._L.jxlt_multi_encoder_last_error(self._enc).decode()))

    def encode(self, planes, distance):
        """planes: float32 [3, h, w] host array (page-locked memory from pinned_empty uploads at PCIe speed).
        Returns a NativeView on the encoder's output buffer (valid until its next encode)."""
        assert planes.dtype == np.float32 and planes.ndim == 3 and planes.shape[0] == 3 and planes.strides[2] == 4
        _, h, w = planes.shape
        ptrs = (fp * 3)(*[planes[c].ctypes.data_as(fp) for c in range(3)])
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        self._check(self._L.jxlt_multi_encoder_encode(self._enc, ptrs, planes.strides[1], w, h, C.c_float(distance),
                                                      C.byref(out), C.byref(n)), "jxlt_multi_encoder_encode")
        return NativeView(out, n.value)

    def encode_pfm(self, payload, w, h, big_endian, distance):
        a = np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        self._check(self._L.jxlt_multi_encoder_encode_pfm(self._enc, a.ctypes.data, w, h, 1 if big_endian else 0,
                                                          C.c_float(distance), C.byref(out), C.byref(n)),
                    "jxlt_multi_encoder_encode_pfm")
        return NativeView(out, n.value)

    def set_device_slab(self, slab, ptrs, pitch_bytes, w, rows, keepalive=None):
        arr = (C.c_void_p * 3)(*ptrs)
        self._check(self._L.jxlt_multi_encoder_set_device_slab(self._enc, slab, arr, pitch_bytes, w, rows),
                    "jxlt_multi_encoder_set_device_slab")
        self._keep[slab] = keepalive

    def encode_resident(self, w, h, distance):
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        self._check(self._L.jxlt_multi_encoder_encode_resident(self._enc, w, h, C.c_float(distance), C.byref(out),
                                                               C.byref(n)), "jxlt_multi_encoder_encode_resident")
        return NativeView(out, n.value)


class SectionRun(C.Structure):
    """jxlt_section_run: sections [first_section, first_section + num_sections) of a context's frame go, back to
    back, to dst_offset bytes into the destination."""
    _fields_ = [("first_section", C.c_uint32), ("num_sections", C.c_uint32), ("dst_offset", C.c_uint64)]


_SLAB_FN = {
    "enqueue": C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Params)),
    "dc_histogram": C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.POINTER(C.c_uint32))),
    "begin_dc_pack": C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32)),
    "ac_histogram": C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.POINTER(C.c_uint32))),
    "measure": C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(PackedSections), C.POINTER(PackedSections)),
    "write": C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(SectionRun), C.c_size_t,
                         C.POINTER(SectionRun), C.c_size_t),
    "finish": C.CFUNCTYPE(C.c_int, C.c_void_p),
}


class SlabOps(C.Structure):
    """jxlt_slab_ops: the slab operations the shard protocol runs on (a device context in production;
    the CPU tests bind their own)."""
    _fields_ = [("self", C.c_void_p)] + [(k, v) for k, v in _SLAB_FN.items()]


class ShardGroup:
    """jxlt_shard_group_*: the ranks of a one-process-per-GPU job assemble ONE frame in a POSIX shared-memory
    segment.  Rank 0 must have returned from the constructor before the other ranks construct theirs."""

    def __init__(self, name, rank, world, output_capacity, max_sections):
        self._L = host_lib()
        L = self._L
        L.jxlt_shard_group_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
        L.jxlt_shard_group_close.argtypes = [C.c_void_p]
        L.jxlt_shard_group_close.restype = None
        L.jxlt_shard_group_last_error.argtypes = [C.c_void_p]
        L.jxlt_shard_group_last_error.restype = C.c_char_p
        L.jxlt_shard_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float,
                                        C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.jxlt_shard_encode_ops.argtypes = [C.c_void_p, C.POINTER(SlabOps), C.c_size_t, C.c_size_t, C.c_float,
                                            C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        self.rank, self.world = rank, world
        self._g = C.c_void_p()
        rc = L.jxlt_shard_group_open(name.encode(), rank, world, output_capacity, max_sections, C.byref(self._g))
        if rc != 0:
            raise JxlTinyError("jxlt_shard_group_open(%s, rank %d) failed (%d)" % (name, rank, rc))

    def close(self):
        if self._g:
            self._L.jxlt_shard_group_close(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _result(self, rc, out, n, what):
        if rc != 0:
            raise JxlTinyError("%s failed (%d): %s" % (what, rc, self._L.jxlt_shard_group_last_error(self._g).decode()))
        return NativeView(out, n.value) if out else None

    STAGES = ("enqueued", "dc_histogram", "ac_histogram", "code_tables", "own_sizes", "layout", "hand_over_issued", "all_placed")

    def last_timeline(self):
        """Host-side stage times of this rank's last frame, ms from the call's start (testing header)."""
        self._L.jxlt_shard_group_last_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        ms = (C.c_double * 8)()
        if self._L.jxlt_shard_group_last_timeline(self._g, ms) != 0:
            return None
        return dict(zip(self.STAGES, [float(v) for v in ms]))

    def encode(self, enc, w, h, distance):
        """Collective: `enc` (an Encoder) holds this rank's slab; returns a NativeView of the whole codestream
        on rank 0, None elsewhere."""
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._L.jxlt_shard_encode(self._g, enc._ctx, w, h, C.c_float(distance), C.byref(out), C.byref(n))
        return self._result(rc, out, n, "jxlt_shard_encode")

    def encode_ops(self, ops, w, h, distance):
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._L.jxlt_shard_encode_ops(self._g, C.byref(ops), w, h, C.c_float(distance), C.byref(out), C.byref(n))
        return self._result(rc, out, n, "jxlt_shard_encode_ops")


class ShardPipeline:
    """jxlt_shard_pipeline_*: frames in flight over a one-process-per-GPU group -- `depth` lanes (shard group +
    device context + host thread each), frame k on lane k % depth.  Rank 0 must have returned from the constructor
    before the other ranks construct theirs.  lane_ops (tests): one SlabOps per lane instead of device contexts."""

    def __init__(self, name, rank, world, device, depth, output_capacity, max_sections, lane_ops=None):
        self._L = host_lib()
        L = self._L
        L.jxlt_shard_pipeline_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                               C.POINTER(C.c_void_p)]
        L.jxlt_shard_pipeline_open_ops.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(SlabOps), C.c_int, C.c_size_t,
                                                   C.c_size_t, C.POINTER(C.c_void_p)]
        L.jxlt_shard_pipeline_close.argtypes = [C.c_void_p]
        L.jxlt_shard_pipeline_close.restype = None
        L.jxlt_shard_pipeline_last_error.argtypes = [C.c_void_p]
        L.jxlt_shard_pipeline_last_error.restype = C.c_char_p
        L.jxlt_shard_pipeline_submit_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t,
                                                        C.c_size_t, C.c_size_t, C.c_float, C.POINTER(C.c_uint64)]
        L.jxlt_shard_pipeline_submit_ops.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.POINTER(C.c_uint64)]
        L.jxlt_shard_pipeline_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_uint8)),
                                               C.POINTER(C.c_size_t)]
        self.rank, self.world, self.depth = rank, world, depth
        self._p = C.c_void_p()
        self._keep = None
        if lane_ops is not None:
            arr = (SlabOps * depth)(*lane_ops)
            self._keep = (arr, lane_ops)
            rc = L.jxlt_shard_pipeline_open_ops(name.encode(), rank, world, arr, depth, output_capacity, max_sections,
                                                C.byref(self._p))
        else:
            rc = L.jxlt_shard_pipeline_open(name.encode(), rank, world, device, depth, output_capacity, max_sections,
                                            C.byref(self._p))
        if rc != 0:
            raise JxlTinyError("jxlt_shard_pipeline_open(%s, rank %d) failed (%d)" % (name, rank, rc))

    def close(self):
        if self._p:
            self._L.jxlt_shard_pipeline_close(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit_device(self, plane_ptrs, pitch_bytes, w, h, slab_rows, distance):
        """This rank's slab (three device pointers; slab_rows 0: none) of the next frame; returns its ticket."""
        ptrs = (C.c_void_p * 3)(*[C.c_void_p(int(p)) for p in plane_ptrs]) if slab_rows else None
        t = C.c_uint64()
        rc = self._L.jxlt_shard_pipeline_submit_device(self._p, ptrs, pitch_bytes, w, h, slab_rows, C.c_float(distance),
                                                       C.byref(t))
        if rc != 0:
            raise JxlTinyError("jxlt_shard_pipeline_submit_device failed (%d)" % rc)
        return int(t.value)

    def submit_ops(self, w, h, distance):
        t = C.c_uint64()
        rc = self._L.jxlt_shard_pipeline_submit_ops(self._p, w, h, C.c_float(distance), C.byref(t))
        if rc != 0:
            raise JxlTinyError("jxlt_shard_pipeline_submit_ops failed (%d)" % rc)
        return int(t.value)

    def wait(self, ticket):
        """NativeView of frame `ticket`'s codestream on rank 0 (valid until `depth` more frames have been
        submitted), None elsewhere."""
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._L.jxlt_shard_pipeline_wait(self._p, ticket, C.byref(out), C.byref(n))
        if rc != 0:
            raise JxlTinyError("jxlt_shard_pipeline_wait failed (%d): %s" %
                               (rc, self._L.jxlt_shard_pipeline_last_error(self._p).decode()))
        return NativeView(out, n.value) if out else None


def denormal_self_check(enc):
    """The bits of 3.0f x 2^-147 as the device computes it (testing header: jxlt_debug_denormal_probe); 12 when FP32
    denormals are kept."""
    L = hip_lib()
    L.jxlt_debug_denormal_probe.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.jxlt_debug_denormal_probe.restype = C.c_int
    bits = C.c_uint32(0)
    rc = L.jxlt_debug_denormal_probe(enc._ctx, C.byref(bits))
    if rc != 0:
        raise JxlTinyError("jxlt_debug_denormal_probe failed: %d" % rc)
    return int(bits.value)
