"""ctypes binding of libmegreader_hip.so (the C ABI declared in include/megreader_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C megreader_amd/csrc``).  There is no
CPU fallback: if the shared object is missing or a kernel reports an error, the caller gets an exception.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEGREADER_HIP_LIB") or os.path.join(_HERE, "csrc", "libmegreader_hip.so")   # tools/ablate_*.py
# point MEGREADER_HIP_LIB at the separate -DMR_ABLATION build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "megreader_hip.h")

MR_F32 = 0
MR_BF16 = 1
ABI_VERSION = 3     # include/megreader_hip.h: MR_ABI_VERSION

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_double
_CODES = {"p": _P, "i": _I, "l": _L, "f": _F, "d": _D, "s": _P}

# name -> argument codes (p pointer, i int, l long long, f float, s hipStream_t)
SIGNATURES = {
    "mr_gemm_nt": "iplpiplpiiiis",
    "mr_gemm_tn": "iplplpiiiiips",
    "mr_gemm_tn2": "iplplpiiiiipps",
    "mr_conv2d_fwd": "ipppp" + "i" * 18 + "s",
    "mr_conv2d_fwd_pool": "ippppp" + "i" * 25 + "s",
    "mr_conv2d_fwd_stats": "ippppp" + "i" * 16 + "s",
    "mr_bn_stats": "ipplis",
    "mr_conv2d_dgrad": "ippp" + "i" * 17 + "s",
    "mr_conv2d_dgrad_add": "ipppp" + "i" * 17 + "s",
    "mr_conv2d_dgrad_bnb": "ippppppppp" + "p" + "i" * 17 + "s",
    "mr_conv2d_wgrad": "ipppp" + "i" * 17 + "s",
    "mr_conv2d_wgrad_tab": "ipppp" + "i" * 17 + "pis",
    "mr_nchw_to_nhwc": "ippiiiiis",
    "mr_nhwc_to_nchw": "ippiiiiis",
    "mr_cast": "ipipls",
    "mr_relu_bwd": "ipppls".replace(" ", ""),
    "mr_add": "ippplis",
    "mr_colsum": "ippiilis",
    "mr_permute_021": "ippiiis",
    "mr_prep_conv_weight": "ipllllppiiiiiis",
    "mr_prep_matrix": "ipipipiiiis",
    "mr_prep_bias": "pppiis",
    "mr_prep_batch": "ipilps",
    "mr_opt_tick": "ps",
    "mr_accumulate_multi": "ippps",
    "mr_zero_multi": "ipps",
    "mr_adaptive_avgpool_multi_fwd": "ippppiiiiis",
    "mr_adaptive_avgpool_multi_bwd": "ipppipiiiis",
    "mr_adam_step": "pppplps",
    "mr_sgd_step": "ppplps",
    "mr_bn_fwd_train": "ipppppppppp" + "iliffps",
    "mr_bn_fwd_eval": "ippppppppp" + "ilifs",
    "mr_bn_bwd": "ipppppppppp" + "pilis",
    "mr_stem_pack": "pllllpis",
    "mr_stem_fwd": "ippllllppppiiiis",
    "mr_stem_bwd": "ipppppllllpiiiis",
    "mr_maxpool_fwd": "ippp" + "i" * 12 + "s",
    "mr_maxpool_bwd": "ipppp" + "i" * 12 + "s",
    "mr_lstm_fwd": "ipppppiiipls",
    "mr_lstm_bwd": "ipppppiiipls",
    "mr_ctc_fwd": "ipipippiiiiiiipppppps",
    "mr_ctc_bwd": "ipppppippipiiiiiipis",
    "mr_softmax_nc1t": "ipipiiis",
    "mr_adaptive_avgpool_fwd": "ippiiiiiis",
    "mr_adaptive_avgpool_bwd": "ippiiiiiis",
    "mr_bilinear_fwd": "ipp" + "i" * 9 + "s",
    "mr_bilinear_bwd": "ipp" + "i" * 8 + "s",
    "mr_nearest_up_fwd": "ippp" + "i" * 7 + "s",
    "mr_nearest_up_bwd": "ipp" + "i" * 7 + "s",
    "mr_copy_channels": "ipiipiilis",
    "mr_deconv2x2_d2s": "ipippiiiis",
    "mr_deconv2x2_s2d": "ippiiiiis",
    "mr_scale_channels": "ipppilis",
    "mr_ctc2d_head_fwd": "ipipipppiiiifs",
    "mr_ctc2d_head_bwd": "ippppipiiiiifs",
    "mr_dcn2_im2col": "ipplplp" + "i" * 11 + "s",
    "mr_dcn2_coord_grad": "ippplplpp" + "i" * 11 + "s",
    "mr_dcn2_col2im": "ipplplp" + "i" * 11 + "s",
    "mr_dcn2_fwd": "ippp" + "plpl" + "pp" + "i" * 12 + "s",
    "mr_dcn2_bwd": "ippp" + "plpl" + "pppppp" + "i" * 12 + "s",
    "mr_dcn2_bwd2": "ippp" + "plpl" + "pppi" + "pppp" + "i" * 12 + "s",
    "mr_dcn2_bwd3": "ippp" + "plpl" + "pppi" + "pppp" + "p" + "i" * 12 + "s",
    "mr_dcn_unpack": "ipippiiiis",
    "mr_dcn_pack_grad": "ippppiiiiis",
    "mr_db_components": "pfpppiiii" + "s",
    "mr_db_box_scores": "pppiiii" + "s",
    "mr_deform_psroi_fwd": "ppppp" + "iiiiiii" + "f" + "iiiii" + "f" + "s",
    "mr_deform_psroi_bwd": "ppppppp" + "iiiiiii" + "f" + "iiiii" + "f" + "s",
    "mr_attn_step_fwd": "ipppppp" + "iiii" + "s",
    "mr_attn_step_bwd": "ippppppppppp" + "iiii" + "s",
    "mr_gru_gates_fwd": "ippppppiis",
    "mr_gru_gates_bwd": "ipppppppiis",
    "mr_embed_rows_fwd": "ipppiiiis",
    "mr_embed_rows_bwd": "ipppiiiis",
    "mr_attn_fwd2": "iplpppppiiiis",
    "mr_attn_bwd2": "ipplplppppplppiiiis",
    "mr_attn_denc": "ipppiiiis",
    "mr_gru_fwd2": "iplppplpppiis",
    "mr_gru_bwd2": "ippppplppplpiis",
    "mr_rows_scatter_add": "ipplpiiis",
    "mr_scatter_strided": "ippiiiiiiiis",
    "mr_gemm_gru_fwd": "iplplplpplpppiiis",
    "mr_gemm_gru_bwd": "iplplpppplppplpiiis",
    "mr_decode_persist_fwd": "ppp" + "l" + "p" + "l" + "pppp" + "i" + "p" * 9 + "l" + "iiii" + "s",
    "mr_decode_persist_bwd": "pp" + "l" + "p" * 9 + "l" + "p" * 7 + "l" + "iiii" + "s",
    "mr_out_nll_fwd": "iplplpplpppppp" + "iiiis",
    "mr_nll_step_fwd": "ipiplppppiiiis",
    "mr_nll_step_feed_fwd": "ipiplpppppp" + "iiis",
    "mr_nll_step_bwd": "ippplppiiis",
    "mr_ctc_greedy_decode": "iplll" + "iiiii" + "pps",
    "mr_ctc2d_greedy_decode": "pllllplll" + "iiiiii" + "pps",
    "mr_seq_measure": "pipiiiippppps",
    "mr_resize_normalize": "ppiiidddps",
    "mr_encode_labels": "ppiippiipps",
    "mr_ctc2d_fwd": "ippppiiiiiipps",
    "mr_ctc2d_bwd": "ippppppppp" + "iiiiiis",
    "mr_tn_flush": "s",
    "mr_tn_flush_beside": "s",
    "mr_db_loss_fwd": "pppppppppp" + "ilffffs",
    "mr_db_head_tail_fwd": "ippppplfs",
    "mr_db_head_tail_bwd": "i" + "pppppppp" + "lfs",
    "mr_db_loss_bwd": "pppppppppppp" + "ilffs",
}

_lib = None

TUNING_FIELDS = ("nt_variant", "nt_deep", "nt_big", "nt_p8", "nt_force_bm", "nt_force_bn", "gemm_skinny", "tn_big", "tn_buf",
                 "tn_taps", "tn_taps_group", "tn_group", "tn_fin", "tn_taps_fin", "tn_taps_w8", "tn_model", "tn_splits",
                 "bn_fused", "lstm_persist", "lstm_fwd_bn", "lstm_bwd_bn", "dcn_fused", "dcn_v1_bwd", "bn_onepass", "skinny_depth", "nt_big_min_k", "tn_taps_min_p", "tn_defer", "pool_fixed", "ctc_linear", "nt_wide8", "nt_ksplit", "nt_m32", "nt_m32_opt", "dcn_gcol", "dcn_col_fwd", "decode_persist")


class Tuning(ctypes.Structure):
    """struct mr_tuning (include/megreader_hip.h): the library's only process-wide switches."""
    _fields_ = [(name, ctypes.c_int) for name in TUNING_FIELDS] + [("reserved", ctypes.c_int * 3)]


def get_tuning():
    """Current process-wide tuning state as a dict."""
    t = Tuning()
    if load().mr_tuning_get(ctypes.byref(t)) != 0:
        raise RuntimeError("mr_tuning_get failed: %s" % load().mr_last_error().decode())
    return {name: getattr(t, name) for name in TUNING_FIELDS}


def set_tuning(**fields):
    """Replace the named fields of the process-wide tuning state (mr_tuning_set); returns their previous values.  A/B and test
    hook: every field selects between kernels that compute the same result (include/megreader_hip.h)."""
    lib = load()
    t = Tuning()
    if lib.mr_tuning_get(ctypes.byref(t)) != 0:
        raise RuntimeError("mr_tuning_get failed: %s" % lib.mr_last_error().decode())
    old = {}
    for name, value in fields.items():
        if name not in TUNING_FIELDS:
            raise KeyError("mr_tuning has no field %r" % name)
        old[name] = getattr(t, name)
        setattr(t, name, int(value))
    if lib.mr_tuning_set(ctypes.byref(t)) != 0:
        raise RuntimeError("mr_tuning_set failed: %s" % lib.mr_last_error().decode())
    return old


def _install_setter_shims(lib):
    """Rounds 1-3 exported one `mr_set_<field>(value) -> previous value` function per switch; tests and tools still spell their
    A/B toggles that way.  The C ABI now has ONE struct (mr_tuning_get / mr_tuning_set): these are Python callables of the old
    names on the loaded library object, nothing more."""
    def one(field, clamp=None):
        def setter(value):
            value = int(value)
            if clamp is not None and not clamp(value):
                return get_tuning()[field]          # the old setters ignored out-of-range values
            return set_tuning(**{field: value})[field]
        return setter
    lib.mr_set_nt_variant = one("nt_variant", lambda v: v in (1, 2))
    lib.mr_set_nt_deep = one("nt_deep", lambda v: 0 <= v <= 2)
    lib.mr_set_nt_big = one("nt_big", lambda v: -1 <= v <= 13)
    lib.mr_set_nt_p8 = one("nt_p8")
    lib.mr_set_gemm_skinny = lambda v: one("gemm_skinny")(1 if v else 0)
    lib.mr_set_tn_big = one("tn_big", lambda v: -1 <= v <= 2)
    lib.mr_set_tn_buf = one("tn_buf", lambda v: v in (0, 1))
    lib.mr_set_tn_taps = one("tn_taps", lambda v: v in (0, 1))
    lib.mr_set_tn_taps_group = one("tn_taps_group", lambda v: v >= 0)
    lib.mr_set_tn_group = one("tn_group", lambda v: v >= 0)
    lib.mr_set_tn_fin = one("tn_fin", lambda v: v in (0, 2))
    lib.mr_set_tn_taps_fin = one("tn_taps_fin", lambda v: 0 <= v <= 2)
    lib.mr_set_tn_taps_w8 = one("tn_taps_w8", lambda v: v in (0, 1))
    lib.mr_set_tn_model = lambda v: one("tn_model")(1 if v else 0)
    lib.mr_set_tn_splits = lambda v: one("tn_splits")(max(0, int(v)))
    lib.mr_set_bn_fused = one("bn_fused", lambda v: v in (0, 1))
    lib.mr_set_dcn_fused = lambda v: one("dcn_fused")(1 if v else 0)
    lib.mr_set_dcn_v1_bwd = lambda v: one("dcn_v1_bwd")(1 if v else 0)

    def set_lstm_persist(on):
        set_tuning(lstm_persist=int(on) if int(on) in (0, 1, 2) else 1)
        return 0
    lib.mr_set_lstm_persist = set_lstm_persist

    def set_lstm_variant(fwd_bn, bwd_bn):
        f = {}
        if fwd_bn >= 0:
            f["lstm_fwd_bn"] = fwd_bn
        if bwd_bn >= 0:
            f["lstm_bwd_bn"] = bwd_bn
        set_tuning(**f)
        return 0
    lib.mr_set_lstm_variant = set_lstm_variant

    def force_nt_tile(bm, bn):
        set_tuning(nt_force_bm=bm, nt_force_bn=bn if bm else 0)
        return 0
    lib.mr_force_nt_tile = force_nt_tile


def header_symbols():
    """Names of all `int mr_*(...)` entry points declared in include/megreader_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\bint\s+(mr_\w+)\s*\(", text)))


def load():
    """Load the shared library (once) and attach argument types.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmegreader_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C megreader_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.mr_last_error.restype = ctypes.c_char_p
    lib.mr_last_error.argtypes = []
    lib.mr_abi_version.restype = ctypes.c_int
    lib.mr_abi_version.argtypes = []
    if lib.mr_abi_version() != ABI_VERSION:
        raise RuntimeError("%s implements C ABI version %d, this binding expects %d -- rebuild it (`make -C megreader_amd/csrc`)"
                           % (LIB_PATH, lib.mr_abi_version(), ABI_VERSION))
    lib.mr_init.restype = ctypes.c_int
    lib.mr_init.argtypes = []
    lib.mr_nt_tile_code.restype = ctypes.c_int
    lib.mr_nt_tile_code.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.mr_sizeof_prep_job.restype = ctypes.c_int
    lib.mr_sizeof_prep_job.argtypes = []
    lib.mr_nt_kernel_code.restype = ctypes.c_int
    lib.mr_nt_kernel_code.argtypes = [ctypes.c_int] * 5
    lib.mr_bn_scratch_doubles.restype = ctypes.c_longlong
    lib.mr_bn_scratch_doubles.argtypes = [ctypes.c_int]
    lib.mr_dcn2_ws_bytes.restype = ctypes.c_longlong
    lib.mr_dcn2_ws_bytes.argtypes = [ctypes.c_int] * 11
    lib.mr_dcn2_dx_direct.restype = ctypes.c_int
    lib.mr_dcn2_dx_direct.argtypes = [ctypes.c_int] * 8
    lib.mr_db_loss_ws_bytes.restype = ctypes.c_longlong
    lib.mr_db_loss_ws_bytes.argtypes = []
    lib.mr_dcn2_fused.restype = ctypes.c_int
    lib.mr_dcn2_fused.argtypes = [ctypes.c_int] * 7
    if hasattr(lib, "mr_set_tn_abl"):      # only libmegreader_hip_abl.so (tools build, include/megreader_hip_ablation.h)
        lib.mr_set_tn_abl.restype = ctypes.c_int
        lib.mr_set_tn_abl.argtypes = [ctypes.c_int]
        lib.mr_set_tn_taps_abl.restype = ctypes.c_int
        lib.mr_set_tn_taps_abl.argtypes = [ctypes.c_int]
    lib.mr_set_tn_taps_workspace.restype = ctypes.c_int
    lib.mr_set_tn_taps_workspace.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.mr_tn_taps_would_run.restype = ctypes.c_int
    lib.mr_tn_taps_would_run.argtypes = [ctypes.c_int] * 17
    lib.mr_lstm_debug_buffer.restype = ctypes.c_int
    lib.mr_lstm_debug_buffer.argtypes = [ctypes.c_void_p]
    lib.mr_lstm_ws_bytes.restype = ctypes.c_longlong
    lib.mr_lstm_ws_bytes.argtypes = [ctypes.c_int] * 4
    lib.mr_decode_persist_ok.restype = ctypes.c_int
    lib.mr_decode_persist_ok.argtypes = [ctypes.c_int] * 5
    lib.mr_decode_persist_ws_bytes.restype = ctypes.c_longlong
    lib.mr_decode_persist_ws_bytes.argtypes = [ctypes.c_int]
    lib.mr_decode_persist_bwd_ok.restype = ctypes.c_int
    lib.mr_decode_persist_bwd_ok.argtypes = [ctypes.c_int] * 5
    lib.mr_decode_persist_bwd_ws_bytes.restype = ctypes.c_longlong
    lib.mr_decode_persist_bwd_ws_bytes.argtypes = [ctypes.c_int]
    lib.mr_sizeof_img_desc.restype = ctypes.c_int
    lib.mr_sizeof_img_desc.argtypes = []
    lib.mr_tn_defer.restype = ctypes.c_int
    lib.mr_tn_defer.argtypes = [ctypes.c_int]
    lib.mr_tn_pending.restype = ctypes.c_int
    lib.mr_tn_pending.argtypes = []
    lib.mr_tn_discard.restype = ctypes.c_int
    lib.mr_tn_discard.argtypes = []
    lib.mr_dcn2_col_saved.restype = ctypes.c_int
    lib.mr_dcn2_col_saved.argtypes = [ctypes.c_int] * 7
    lib.mr_conv2d_fwd_pool_ok.restype = ctypes.c_int
    lib.mr_conv2d_fwd_pool_ok.argtypes = [ctypes.c_int] * 23
    lib.mr_phase_timer.restype = ctypes.c_int
    lib.mr_phase_timer.argtypes = [ctypes.c_int]
    lib.mr_phase_read.restype = ctypes.c_int
    lib.mr_phase_read.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.mr_stem_bwd_workspace.restype = ctypes.c_longlong
    lib.mr_stem_bwd_workspace.argtypes = [ctypes.c_int]
    for name, codes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [_CODES[c] for c in codes]
    for name in ("mr_tuning_get", "mr_tuning_set", "mr_tuning_defaults"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [ctypes.POINTER(Tuning)]
    _install_setter_shims(lib)
    _lib = lib
    if torch.cuda.is_available():
        if lib.mr_init() != 0:
            raise RuntimeError("mr_init failed: %s" % lib.mr_last_error().decode())
    elif lib.mr_init() != 0:
        # no GPU (build container, host-only queries such as mr_nt_kernel_code): mr_init still applies MEGREADER_TUNING before
        # it touches the device, so a malformed variable is reported here too; the device part failing is expected
        err = lib.mr_last_error().decode()
        if "MEGREADER_TUNING" in err:
            raise RuntimeError("mr_init failed: %s" % err)
    return lib


HOST_ONLY = ("mr_abi_version", "mr_nt_tile_code", "mr_init", "mr_tuning_get", "mr_tuning_set", "mr_tuning_defaults",
             "mr_stem_bwd_workspace", "mr_lstm_ws_bytes", "mr_lstm_debug_buffer", "mr_dcn2_ws_bytes", "mr_bn_scratch_doubles",
             "mr_sizeof_img_desc", "mr_nt_kernel_code", "mr_tn_taps_would_run", "mr_set_tn_taps_workspace",
             "mr_sizeof_prep_job", "mr_tn_defer", "mr_tn_pending", "mr_tn_discard", "mr_phase_timer", "mr_phase_read", "mr_conv2d_fwd_pool_ok", "mr_dcn2_col_saved", "mr_dcn2_dx_direct", "mr_dcn2_fused", "mr_db_loss_ws_bytes", "mr_decode_persist_ok", "mr_decode_persist_ws_bytes", "mr_decode_persist_bwd_ok",
             "mr_decode_persist_bwd_ws_bytes")  # entry points that take no stream and launch nothing


def dtype_code(dtype):
    if dtype == torch.float32:
        return MR_F32
    if dtype == torch.bfloat16:
        return MR_BF16
    raise TypeError("megreader_amd kernels support float32 and bfloat16, got %s" % dtype)


def vec_of(dtype):
    return 4 if dtype == torch.float32 else 8


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


class KernelTimer(object):
    """Optional HIP-event bracketing of selected C-ABI calls on the stream they are launched on (bench.py uses it
    to measure the dominant kernel's launch durations live inside the timed region)."""

    def __init__(self, names, track_deferred=False):
        self.names = set(names)
        self.records = []  # (name, args, start_event, end_event)
        # track_deferred: calls that the library only RECORDS (mr_tn_defer: deferred weight-gradient problems) are collected and
        # attributed to the mr_tn_flush that launches them as one grouped kernel -- ("mr_tn_flush", [(name, args), ...], e0, e1)
        self.track_deferred = track_deferred
        self.deferred = []

    def results(self):
        """[(name, args, milliseconds)] -- call after torch.cuda.synchronize()."""
        return [(n, a, s.elapsed_time(e)) for n, a, s, e in self.records]


TIMER = None  # set to a KernelTimer to enable

_TN_WS = {}
_TN_WS_CALLS = frozenset(("mr_gemm_tn", "mr_gemm_tn2", "mr_conv2d_wgrad", "mr_conv2d_wgrad_tab",
                          # (the split reduction of the NT kernels, mr_tuning.nt_ksplit, uses the same workspace)
                          "mr_conv2d_fwd", "mr_conv2d_fwd_stats", "mr_conv2d_dgrad", "mr_conv2d_dgrad_add", "mr_conv2d_dgrad_bnb",
                          "mr_gemm_nt"))


def ensure_tn_workspace(device=None):
    """Register (once per device) the workspace of the weight-gradient kernels' in-launch split reduction
    (mr_set_tn_taps_workspace): 16 KB of tickets + one 147456-byte slab per workgroup of a full launch (2 per CU).
    Launches that use it must be stream-ordered with each other: MEGREADER_FAN / MEGREADER_OVERLAP (weight-gradient
    GEMMs forked onto side streams) switch the reduction back to plain atomics."""
    if device is None:
        idx = torch.cuda.current_device()
    else:
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _TN_WS.get(idx)
    if ws is None:
        lib = load()
        cus = torch.cuda.get_device_properties(idx).multi_processor_count
        nbytes = 16384 + (2 * cus + 64) * 147456
        ws = _TN_WS[idx] = torch.zeros(nbytes, dtype=torch.uint8, device=torch.device("cuda", idx))
        with torch.cuda.device(idx):   # the library keeps one workspace per device, keyed by the current device
            rc = lib.mr_set_tn_taps_workspace(ws.data_ptr(), nbytes)
        if rc != 0:
            raise RuntimeError("mr_set_tn_taps_workspace failed: %s" % lib.mr_last_error().decode())
        if os.environ.get("MEGREADER_FAN", "0") == "1" or os.environ.get("MEGREADER_OVERLAP", "0") == "1":
            lib.mr_set_tn_group(1)
            lib.mr_set_tn_taps_group(1)
            set_tuning(nt_ksplit=0)      # GEMMs of one layer on several streams: nobody may use the shared slabs / tickets
    return ws


def call(name, *args):
    """Invoke a C entry point on torch's current HIP stream; raise RuntimeError on a non-zero return code."""
    lib = load()
    if name in _TN_WS_CALLS and torch.cuda.current_device() not in _TN_WS:
        ensure_tn_workspace()
    timer = TIMER
    if timer is not None and name in timer.names:
        pend = lib.mr_tn_pending() if timer.track_deferred else 0
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args, stream_ptr())
        e1.record()
        if timer.track_deferred and lib.mr_tn_pending() > pend:
            timer.deferred.append((name, args))          # recorded, not launched: its time is the flush's
        else:
            timer.records.append((name, args, e0, e1))
    elif timer is not None and timer.track_deferred and name in ("mr_tn_flush", "mr_tn_flush_beside") and timer.deferred:
        group, timer.deferred = timer.deferred, []
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args, stream_ptr())
        e1.record()
        timer.records.append(("mr_tn_flush", group, e0, e1))
    else:
        rc = getattr(lib, name)(*args, stream_ptr())
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (name, rc, lib.mr_last_error().decode()))


def dcn_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, backward, device):
    """Caller-owned workspace of mr_dcn2_fwd / mr_dcn2_bwd (include/megreader_hip.h: mr_dcn2_ws_bytes): None when the
    fused forward needs none, else an uninitialised byte buffer (CSR of the scatter pattern / column matrix)."""
    n = load().mr_dcn2_ws_bytes(dtype_code(dtype), N, H, W, C, Co, kh, kw, Ho, Wo, 1 if backward else 0)
    if n <= 0:
        return None
    return torch.empty((n,), dtype=torch.uint8, device=device)


_DCN_WS = {}


def dcn_backward_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, device):
    """(workspace, flags) for mr_dcn2_bwd2.  Fused path: ONE persistent buffer per geometry, zeroed when it is created -- the CSR
    build returns its counters to zero by itself, so later calls pass flags bit 0 and no memset node runs in front of them (13
    DCN layers per detector step; layers of one geometry share the buffer, stream-ordered).  General path: a fresh buffer, 0."""
    if not load().mr_dcn2_fused(dtype_code(dtype), H, W, C, Co, kh, kw):
        return dcn_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, True, device), 0
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), dtype, N, H, W, C, Co, kh, kw, Ho, Wo)
    ws = _DCN_WS.get(key)
    if ws is None:
        n = load().mr_dcn2_ws_bytes(dtype_code(dtype), N, H, W, C, Co, kh, kw, Ho, Wo, 1)
        # (never evicted: captured hipGraphs hold raw pointers into these buffers -- ADVICE r5; one entry per layer geometry)
        ws = _DCN_WS[key] = torch.zeros((max(int(n), 16),), dtype=torch.uint8, device=dev)
    return ws, 1


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError(
                "megreader_amd ops run only on an AMD GPU (HIP); got a %s tensor. There is no CPU fallback." % t.device)
