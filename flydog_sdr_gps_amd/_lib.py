"""ctypes binding of libkiwigpu.so (include/kiwigpu.h).  Fails loudly."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABI_VERSION = 4


class KiwiGpuError(RuntimeError):
    def __init__(self, status, where, text):
        super().__init__("%s failed: %s (status %d)" % (where, text, status))
        self.status = status


def library_path():
    """The in-tree build; KIWIGPU_LIBRARY names another build of the same ABI (experiments)."""
    return os.environ.get("KIWIGPU_LIBRARY") or os.path.join(_HERE, "libkiwigpu.so")


class AcqResultC(C.Structure):
    _fields_ = [("snr", C.c_float), ("dop", C.c_int32), ("idx", C.c_int32), ("valid", C.c_int32)]


class AcqCellC(C.Structure):
    _fields_ = [("snr", C.c_float), ("max_pwr", C.c_float), ("tot_pwr", C.c_float),
                ("idx", C.c_int32)]


class WfChanCfgC(C.Structure):
    _fields_ = [("zoom", C.c_int32), ("window_func", C.c_int32), ("interp", C.c_int32),
                ("cic_comp", C.c_int32), ("overlapped", C.c_int32), ("fft_used", C.c_int32),
                ("plot_width", C.c_int32), ("plot_width_clamped", C.c_int32),
                ("fft_offset", C.c_float)]


result_dtype = np.dtype([("snr", "<f4"), ("dop", "<i4"), ("idx", "<i4"), ("valid", "<i4")])
cell_dtype = np.dtype([("snr", "<f4"), ("max_pwr", "<f4"), ("tot_pwr", "<f4"), ("idx", "<i4")])

# name -> (restype, argtypes): every symbol include/kiwigpu.h declares
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "kg_strerror": (C.c_char_p, [_i]),
    "kg_last_error": (C.c_char_p, []),
    "kg_abi_version": (_i, []),
    "kg_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "kg_ctx_create_on_stream": (_i, [_i, _vp, C.POINTER(_vp)]),
    "kg_ctx_destroy": (None, [_vp]),
    "kg_ctx_sync": (_i, [_vp]),
    "kg_ctx_poll": (_i, [_vp]),
    "kg_ctx_stream": (_vp, [_vp]),
    "kg_ctx_device_name": (_i, [_vp, C.c_char_p, _sz]),
    "kg_ctx_num_cus": (_i, [_vp]),
    "kg_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "kg_dev_free": (_i, [_vp, _vp]),
    "kg_dev_upload": (_i, [_vp, _vp, _vp, _sz]),
    "kg_dev_download": (_i, [_vp, _vp, _vp, _sz]),
    "kg_dev_mem_info": (_i, [_vp, _vp, _vp]),
    "kg_timer_start": (_i, [_vp]),
    "kg_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "kg_ctx_mark": (_i, [_vp, _i]),
    "kg_acq_create": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_vp)]),
    "kg_acq_create_shape": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "kg_acq_nsamples": (_i, [_vp]),
    "kg_acq_fft_len": (_i, [_vp]),
    "kg_acq_destroy": (None, [_vp]),
    "kg_acq_set_code": (_i, [_vp, _i, _vp, _i, _i, _i]),
    "kg_acq_set_code_fft": (_i, [_vp, _i, _vp, _i]),
    "kg_acq_get_code_fft": (_i, [_vp, _i, _vp]),
    "kg_acq_sample_bits": (_i, [_vp, _i, _vp]),
    "kg_acq_sample_bits_dev": (_i, [_vp, _i, _vp]),
    "kg_acq_sample_iq16": (_i, [_vp, _i, _vp]),
    "kg_acq_sample_iq16_dev": (_i, [_vp, _i, _vp]),
    "kg_acq_sample_iq16_batch": (_i, [_vp, _i, _i, _vp, _sz]),
    "kg_acq_sample_iq16_batch_dev": (_i, [_vp, _i, _i, _vp, _sz]),
    "kg_acq_set_data_fft": (_i, [_vp, _i, _vp]),
    "kg_acq_get_data_fft": (_i, [_vp, _i, _vp]),
    "kg_acq_get_data_td": (_i, [_vp, _i, _vp]),
    "kg_acq_correlate_async": (_i, [_vp, _i, _vp, _i]),
    "kg_acq_correlate_blocks_async": (_i, [_vp, _i, _i, _vp, _i]),
    "kg_acq_fetch": (_i, [_vp, _vp, _vp]),
    "kg_acq_correlate": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "kg_acq_results_dev": (_vp, [_vp]),
    "kg_wf_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "kg_wf_destroy": (None, [_vp]),
    "kg_wf_set_tables": (_i, [_vp, _vp, _vp]),
    "kg_wf_set_channel": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "kg_wf_frames_dev": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kg_wf_frames_at_dev": (_i, [_vp, _i, _vp, _vp, C.c_uint64, _vp, _vp]),
    "kg_wf_frames": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kg_wf_debug_frame": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "kg_ddc_create": (_i, [_vp, _i, _sz, C.POINTER(_vp)]),
    "kg_ddc_destroy": (None, [_vp]),
    "kg_ddc_set_wf": (_i, [_vp, _i, C.c_uint64, _i]),
    "kg_ddc_reset_wf": (_i, [_vp, _i]),
    "kg_ddc_set_phase": (_i, [_vp, _i, C.c_uint64]),
    "kg_ddc_nco_table": (_i, [_vp, _vp]),
    "kg_ddc_wf_capture_dev": (_i, [_vp, _vp, _sz, _vp, _i, _vp, _sz, _sz, _vp]),
    "kg_ddc_wf_set_deferred": (_i, [_vp, _i]),
    "kg_ddc_wf_join": (_i, [_vp, _vp]),
    "kg_ddc_wf_tail_after": (_i, [_vp, _vp]),
    "kg_ddc_wf_outputs": (C.c_long, [_vp, _i, _sz]),
    "kg_ddc_wf_push_dev": (_i, [_vp, _vp, _sz, _vp, _i, _vp, _sz, _vp]),
    "kg_rxddc_create": (_i, [_vp, _i, _sz, C.POINTER(_vp)]),
    "kg_rxddc_create_mode": (_i, [_vp, _i, _sz, _i, C.POINTER(_vp)]),
    "kg_rxddc_decim": (_i, [_vp]),
    "kg_rxddc_destroy": (None, [_vp]),
    "kg_rxddc_set_freq": (_i, [_vp, _i, C.c_uint64]),
    "kg_rxddc_reset": (_i, [_vp, _i]),
    "kg_rxddc_outputs": (C.c_long, [_vp, _i, _sz]),
    "kg_rxddc_push_dev": (_i, [_vp, _vp, _sz, _vp, _i, _vp, _sz, _vp]),
    "kg_dpump_unpack_dev": (_i, [_vp, _vp, _i, _i, _vp, C.c_float, C.c_float, C.c_float, _i, _vp, _sz]),
    "kg_dpump_unpack_rows_dev": (_i, [_vp, _vp, _sz, _i, _i, _vp, C.c_float, C.c_float, C.c_float, _i, _vp, _sz]),
    "kg_fir_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "kg_fir_destroy": (None, [_vp]),
    "kg_fir_setup": (_i, [_vp, _i, C.c_float, C.c_float, C.c_float, C.c_float, _i, _i, _i]),
    "kg_fir_set_coef": (_i, [_vp, _i, _vp]),
    "kg_fir_get_coef": (_i, [_vp, _i, _vp]),
    "kg_fir_reset": (_i, [_vp, _i]),
    "kg_fir_pos": (_i, [_vp, _i]),
    "kg_fir_process": (_i, [_vp, _i, _vp, _i, _vp]),
    "kg_fir_process_dev": (_i, [_vp, _vp, _i, _vp, _sz, _i, _vp, _sz, _vp]),
    "kg_post_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "kg_post_destroy": (None, [_vp]),
    "kg_post_set_agc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, C.c_float]),
    "kg_post_agc_delay": (_i, [_vp, _i]),
    "kg_post_set_smeter": (_i, [_vp, _i, C.c_float]),
    "kg_post_set_mode": (_i, [_vp, _i, _i]),
    "kg_post_reset": (_i, [_vp, _i]),
    "kg_post_process_dev": (_i, [_vp, _vp, _i, _vp, _sz, _i, _vp, _vp, _vp, _sz]),
    "kg_post_smeter": (_i, [_vp, _vp, _i, _vp, _vp]),
    "kg_post_cfir_init_lp": (_i, [_vp, _i, _i, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "kg_post_cfir_init_const": (_i, [_vp, _i, _i, _i, _vp, C.c_float]),
    "kg_post_cfir_get_taps": (_i, [_vp, _i, _i, _vp]),
    "kg_post_cfir_process_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _i, _vp, _sz]),
    "kg_post_squelch_perform_dev": (_i, [_vp, _vp, _i, _vp, _sz, _i, _vp, _sz]),
    "kg_post_set_am_passband": (_i, [_vp, _i, C.c_double, C.c_double, C.c_double]),
    "kg_post_set_deemp": (_i, [_vp, _i, _i, _i]),
    "kg_post_squelch_setup": (_i, [_vp, _i, C.c_float]),
    "kg_post_squelch_set": (_i, [_vp, _i, _i, _i]),
    "kg_post_squelch_reset": (_i, [_vp, _i]),
    "kg_post_squelch_state": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "kg_adpcm_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "kg_adpcm_destroy": (None, [_vp]),
    "kg_adpcm_set_state": (_i, [_vp, _i, _i, _i]),
    "kg_adpcm_get_state": (_i, [_vp, _i, _vp, _vp]),
    "kg_adpcm_encode_dev": (_i, [_vp, _vp, _i, _vp, _sz, _i, _vp, _sz]),
    "kg_snd_payload_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp, _sz]),
    "kg_snd_header": (None, [C.c_uint8, C.c_uint32, C.c_float, _vp]),
    "kg_wf_packets_dev": (_i, [_vp, _vp, _sz, _i, _vp, _vp, _sz, _vp]),
    "kg_fir_process_taps_dev": (_i, [_vp, _vp, _i, _vp, _sz, _i, _vp, _sz, _vp, _vp, _vp, _sz]),
    "kg_fir_refilter_dev": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp, _sz]),
    "kg_fir_set_coef_plain": (_i, [_vp, _i, _vp]),
    "kg_snd_gps_begin": (None, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "kg_snd_gps_stamp": (None, [_vp, _i, _i, _i, _i, _i, C.c_double, C.c_double, C.c_uint64, _vp]),
    "kg_acq_chan_start": (None, [_i, _i, _i, C.c_double, _vp]),
    "kg_aper_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "kg_aper_destroy": (None, [_vp]),
    "kg_aper_update_dev": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _i]),
    "kg_aper_report": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "kg_aper_get": (_i, [_vp, _i, _vp]),
    "kg_ddc_wf_step_dev": (_i, [_vp, _vp, _sz, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "kg_rxbank_create": (_i, [_i, _i, _sz, _i, C.POINTER(_vp)]),
    "kg_rxbank_destroy": (None, [_vp]),
    "kg_rxbank_ctx": (_vp, [_vp]),
    "kg_rxbank_ddc": (_vp, [_vp]),
    "kg_rxbank_wf": (_vp, [_vp]),
    "kg_rxbank_rxddc": (_vp, [_vp]),
    "kg_rxbank_fir": (_vp, [_vp]),
    "kg_rxbank_post": (_vp, [_vp]),
    "kg_rxbank_adpcm": (_vp, [_vp]),
    "kg_rxbank_set_wf": (_i, [_vp, _i, C.c_uint64, _i, _i]),
    "kg_rxbank_set_wf_pkt": (_i, [_vp, _i, C.c_uint32, C.c_uint32, _i]),
    "kg_rxbank_set_unpack": (_i, [_vp, C.c_float, C.c_float, C.c_float, _i]),
    "kg_rxbank_step": (_i, [_vp, _vp, _vp, _vp]),
    "kg_rxbank_adc_done": (_i, [_vp, _vp, _i]),
    "kg_snd_iq_payload_dev": (_i, [_vp, _vp, _i, _vp, _sz, _i, _i, _vp, _sz]),
    "kg_rxbank_set_little_endian": (_i, [_vp, _i, _i]),
    "kg_post_get_mode": (_i, [_vp, _i]),
    "kg_math_dev": (_i, [_vp, _i, C.c_float, _vp, C.c_uint32, C.c_size_t, _vp]),
    "kg_rxbank_poll": (_i, [_vp]),
    "kg_rxbank_ready": (_i, [_vp]),
    "kg_rxbank_join": (_i, [_vp, _i]),
    "kg_rxbank_leave": (_i, [_vp, _i]),
    "kg_rxbank_is_active": (_i, [_vp, _i]),
    "kg_rxbank_audio_map": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "kg_fir_process_each_dev": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _vp, _sz, _vp]),
    "kg_rxbank_sync": (_i, [_vp]),
    "kg_rxbank_frame_map": (_i, [_vp, _vp, _vp, _vp]),
    "kg_rxbank_buffers": (_i, [_vp, _vp]),
    "kg_rxbank_host_profile": (_i, [_vp, C.c_char_p, _sz]),
    "kg_acq_debug_fft_stamps": (_i, [_vp, _i, _vp, _i]),
    "kg_acq_debug_corr_stamps": (_i, [_vp, _i, _vp, _i, _vp, _i]),
}


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one
    process cannot both own the GPU: whichever initialises second reports "No HIP GPUs are
    available" (measured on the MI355X box: libkiwigpu.so first, then torch.zeros(device="cuda")
    fails; torch first, both work, because libkiwigpu.so's NEEDED libamdhip64 then resolves to the
    copy torch has mapped).  So when torch is installed it is imported BEFORE libkiwigpu.so is
    opened.  A host without torch uses the system runtime; KIWIGPU_NO_TORCH_PRELOAD=1 skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("KIWIGPU_NO_TORCH_PRELOAD") == "1":
        return
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception as e:      # noqa: BLE001 -- a broken torch install must not take the C-ABI users down
            import warnings
            warnings.warn("kiwigpu: torch is installed but failed to import (%s: %s); libkiwigpu.so will use "
                          "the system HIP runtime" % (type(e).__name__, e))


def load_library():
    """dlopen libkiwigpu.so and bind every declared symbol.  No fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise KiwiGpuError(-1, "load_library",
                           "%s is not built (run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C flydog_sdr_gps_amd/csrc`); there is no "
                           "CPU fallback" % path)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.kg_abi_version() != ABI_VERSION:
        raise KiwiGpuError(-2, "load_library", "ABI version mismatch")
    _LIB = lib
    return lib


def check(status, where):
    if status < 0:
        lib = load_library()
        text = lib.kg_last_error().decode() or lib.kg_strerror(status).decode()
        raise KiwiGpuError(status, where, text)
    return status


def borrow(cls, ctx, handle, **attrs):
    """An object of one of the wrapper classes (Ddc, Waterfall, RxDdc, FastFir, Post, Adpcm) over a handle another object
    owns -- the per-seam objects of a receiver bank (kg_rxbank_ddc() ...): the configuration methods work, close() leaves the handle
    alone.  The host-buffer conveniences (push / process / encode ...) refuse: a bank's objects address the rows of every
    caller-visible buffer by RECEIVER number (its buffers hold one row per receiver), not by position in the call's list."""
    self = cls.__new__(cls)
    self.ctx, self.lib = ctx, ctx.lib
    self.h = C.c_void_p(handle) if isinstance(handle, int) else handle
    self._borrowed = True
    for k, v in attrs.items():
        setattr(self, k, v)
    return self


def own_rows(obj, what):
    """Guard of the host-buffer conveniences: see borrow()."""
    if getattr(obj, "_borrowed", False):
        raise KiwiGpuError(-5, what, "the object belongs to a receiver bank, whose rows are addressed by receiver number: step the bank "
                           "and read its buffers (RxBank.fetch)")


def ptr(a):
    """void* of a numpy array, or pass through an int device pointer."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One per GPU per process (kg_ctx).  stream: a hipStream_t handle (int), e.g.
    torch.cuda.current_stream().cuda_stream -- 0 is HIP's legacy default stream and is used
    as such -- or None for a library-owned non-blocking stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        if stream is None:
            check(self.lib.kg_ctx_create(int(device), None, C.byref(h)), "kg_ctx_create")
        else:
            check(self.lib.kg_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h)),
                  "kg_ctx_create_on_stream")
        self.h = h
        self.device = device

    @classmethod
    def borrow(cls, handle, device=0):
        """A Context over a kg_ctx another object owns (a receiver bank's): close() leaves it alone."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.h = C.c_void_p(handle) if isinstance(handle, int) else handle
        self.device = device
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.kg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.kg_ctx_sync(self.h), "kg_ctx_sync")

    def poll(self):
        return check(self.lib.kg_ctx_poll(self.h), "kg_ctx_poll") == 1

    @property
    def stream(self):
        return self.lib.kg_ctx_stream(self.h)

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        check(self.lib.kg_ctx_device_name(self.h, buf, 256), "kg_ctx_device_name")
        return buf.value.decode()

    @property
    def num_cus(self):
        return check(self.lib.kg_ctx_num_cus(self.h), "kg_ctx_num_cus")

    # device memory without any other HIP runtime in the process
    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.kg_dev_alloc(self.h, int(nbytes), C.byref(p)), "kg_dev_alloc")
        return p.value

    def free(self, dptr):
        check(self.lib.kg_dev_free(self.h, C.c_void_p(dptr)), "kg_dev_free")

    def upload(self, dptr, host):
        host = np.ascontiguousarray(host)
        check(self.lib.kg_dev_upload(self.h, C.c_void_p(dptr), ptr(host), host.nbytes), "kg_dev_upload")

    def download(self, dptr, host):
        assert host.flags["C_CONTIGUOUS"]
        check(self.lib.kg_dev_download(self.h, ptr(host), C.c_void_p(dptr), host.nbytes), "kg_dev_download")

    def mem_info(self):
        """-> (free bytes, total bytes) of the device"""
        f, t = C.c_size_t(), C.c_size_t()
        check(self.lib.kg_dev_mem_info(self.h, C.byref(f), C.byref(t)), "kg_dev_mem_info")
        return f.value, t.value

    def mark(self, tag):
        """profiling aid: an empty kernel of `tag` workgroups on the stream (kg_ctx_mark)"""
        check(self.lib.kg_ctx_mark(self.h, int(tag)), "kg_ctx_mark")

    def timer_start(self):
        check(self.lib.kg_timer_start(self.h), "kg_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.kg_timer_stop(self.h, C.byref(ms)), "kg_timer_stop")
        return ms.value
