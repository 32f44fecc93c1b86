"""ctypes binding of libcartpolepp_hip.so (the C ABI declared in include/cartpolepp_abi.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing this module
raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C cartpoleplusplus_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The release library reads no environment variable and is the only one loaded by default.  CARTPOLEPP_ABLATION=1 selects the
# ablation build of the SAME sources (lib/libcartpolepp_hip_ablation.so: the CPP_* kernel-selection switches compiled in; parity
# tests of the fallback kernels, bench.py's f32 control run); CARTPOLEPP_ABLATION=<name> an in-tree experiment build
# lib/libcartpolepp_hip_<name>.so (profiles/).  No path can be injected: the file must sit in this package's lib/ directory.
_variant = os.environ.get("CARTPOLEPP_ABLATION", "")
if _variant and not _variant.replace("_", "").isalnum():
    raise ImportError("cartpoleplusplus_amd: CARTPOLEPP_ABLATION=%r is not a build name" % _variant)
if _variant == "exact":
    raise ImportError("cartpoleplusplus_amd: CARTPOLEPP_ABLATION=exact names a build that no longer exists -- the exact-product "
                      "arithmetic is a mode of the release library (Context.set_precision('exact'), --exact-products, "
                      "bench.py --precision exact)")
LIB_PATH = os.path.join(_HERE, "lib", "libcartpolepp_hip%s.so" % (
    "" if _variant in ("", "0") else "_ablation" if _variant == "1" else "_" + _variant))

CPP_F32, CPP_F16, CPP_U8 = 0, 1, 2
CPP_ACTOR, CPP_CRITIC, CPP_HEAD = 0, 1, 2
CPP_OPT_SGD, CPP_OPT_MOMENTUM, CPP_OPT_ADAM = 0, 1, 2
CPP_PRECISION_FAST, CPP_PRECISION_EXACT = 0, 1      # cpp_ctx_set_precision


class NetSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pixel", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C", C.c_int32), ("state_elems", C.c_int32), ("action_dim", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden", C.c_int32 * 8), ("head_out", C.c_int32),
                ("head_act", C.c_int32), ("use_batch_norm", C.c_int32), ("use_dropout", C.c_int32),
                ("dropout_seed", C.c_uint32)]


class DdpgHyper(C.Structure):
    _fields_ = [("actor_learning_rate", C.c_float), ("critic_learning_rate", C.c_float),
                ("discount", C.c_float), ("gradient_clip", C.c_float),
                ("target_update_rate", C.c_float)]


class NafHyper(C.Structure):
    _fields_ = [("discount", C.c_float), ("gradient_clip", C.c_float), ("target_update_rate", C.c_float),
                ("optimiser", C.c_int32), ("learning_rate", C.c_float), ("momentum", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float)]


_P = C.c_void_p
_I, _L, _F = C.c_int, C.c_int64, C.c_float
_U64 = C.c_uint64
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Every symbol declared in include/cartpolepp_abi.h appears here;
# tests/test_abi_symbols.py checks the header and this table against the built library.
SIGNATURES = {
    "cpp_abi_version": (_I, []),
    "cpp_last_error": (C.c_char_p, []),
    "cpp_ctx_create": (_I, [_I, _P, _PP]),
    "cpp_ctx_destroy": (_I, [_P]),
    "cpp_sync": (_I, [_P]),
    "cpp_ctx_set_precision": (_I, [_P, _I]),
    "cpp_ctx_get_precision": (_I, [_P, C.POINTER(_I)]),
    "cpp_ctx_set_route_threshold": (_I, [_P, _F]),
    "cpp_ctx_get_route": (_I, [_P, C.POINTER(_I), C.POINTER(_F)]),
    "cpp_timer_begin": (_I, [_P]),
    "cpp_timer_end": (_I, [_P, C.POINTER(_F)]),
    "cpp_prof_enable": (_I, [_P, _I]),
    "cpp_prof_reset": (_I, [_P]),
    "cpp_prof_num_kernels": (_I, []),
    "cpp_prof_kernel_name": (C.c_char_p, [_I]),
    "cpp_prof_read": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(_L)]),
    "cpp_net_create": (_I, [_P, C.POINTER(NetSpec), _I, _PP]),
    "cpp_net_destroy": (_I, [_P]),
    "cpp_net_num_params": (_L, [_P]),
    "cpp_net_num_vars": (_I, [_P]),
    "cpp_net_var_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_L)]),
    "cpp_net_set_params": (_I, [_P, _P, _L]),
    "cpp_net_get_params": (_I, [_P, _P, _L]),
    "cpp_net_get_grads": (_I, [_P, _P, _L]),
    "cpp_net_soft_update": (_I, [_P, _P, _F]),
    "cpp_net_forward": (_I, [_P, _P, _I, _I, _P, _P]),
    "cpp_net_forward_each": (_I, [_P, _P, _I, _I, _P, _P]),
    "cpp_net_get_pool": (_I, [_P, _I, _I, _P]),
    "cpp_batch_create": (_I, [_P, _I, _L, _I, _PP]),
    "cpp_batch_destroy": (_I, [_P]),
    "cpp_batch_upload": (_I, [_P, _I, _P, _P, _I, _P, _P, _P]),
    "cpp_batch_download": (_I, [_P, _P, _P, _P, _P, _P]),
    "cpp_batch_size": (_I, [_P]),
    "cpp_batch_state_dtype": (_I, [_P]),
    "cpp_replay_create": (_I, [_P, _I, _I, _L, _I, _PP]),
    "cpp_replay_create_ex": (_I, [_P, _I, _I, _L, _I, _I, _PP]),
    "cpp_replay_destroy": (_I, [_P]),
    "cpp_replay_write_states": (_I, [_P, _P, _I, _P, _I]),
    "cpp_replay_write_rows": (_I, [_P, _P, _I, _P, _P, _P, _P, _P]),
    "cpp_replay_read_rows": (_I, [_P, _P, _I, _P, _P, _P, _P, _P]),
    "cpp_replay_set_stats_channels": (_I, [_P, _I]),
    "cpp_replay_set_size": (_I, [_P, _I]),
    "cpp_replay_read_states": (_I, [_P, _P, _I, _P]),
    "cpp_replay_sample": (_I, [_P, _I, _P, _U64, _U64, _I, _P]),
    "cpp_replay_last_indexes": (_I, [_P, _I, _P]),
    "cpp_replay_fill_synthetic": (_I, [_P, _I, _U64]),
    "cpp_ddpg_create": (_I, [_P, _P, _P, _P, _P, C.POINTER(DdpgHyper), _PP]),
    "cpp_ddpg_destroy": (_I, [_P]),
    "cpp_ddpg_train_actor": (_I, [_P, _P]),
    "cpp_ddpg_train_critic": (_I, [_P, _P]),
    "cpp_ddpg_check_loss": (_I, [_P, _P, _P, _P, _P]),
    "cpp_ddpg_q_gradients_wrt_actions": (_I, [_P, _P, _P, _P, _P]),
    "cpp_ddpg_compute_gradients": (_I, [_P, _P]),
    "cpp_ddpg_grad_buffer": (_I, [_P, _PP, C.POINTER(_L)]),
    "cpp_ddpg_apply_gradients": (_I, [_P, _F]),
    "cpp_ddpg_update_targets": (_I, [_P]),
    "cpp_ddpg_train_step": (_I, [_P, _P, _I, _I, _P, _U64]),
    "cpp_ddpg_train_rows": (_I, [_P, _P, _I, _P]),
    "cpp_ddpg_sample_and_compute": (_I, [_P, _P, _I, _U64]),
    "cpp_ddpg_last_stats": (_I, [_P, _P]),
    "cpp_ddpg_last_values": (_I, [_P, _I, _P, _P, _P, _P]),
    "cpp_comm_unique_id": (_I, [_P, _I]),
    "cpp_comm_create": (_I, [_P, _P, _I, _I, _PP]),
    "cpp_comm_destroy": (_I, [_P]),
    "cpp_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "cpp_comm_allreduce": (_I, [_P, _P, _L, _I]),
    "cpp_comm_max_double": (_I, [_P, C.POINTER(C.c_double)]),
    "cpp_comm_max_doubles": (_I, [_P, C.POINTER(C.c_double), _I]),
    "cpp_comm_barrier": (_I, [_P]),
    "cpp_ddpg_allreduce_grads": (_I, [_P, _P]),
    "cpp_ddpg_average_params": (_I, [_P, _P]),
    "cpp_ddpg_dp_train_step": (_I, [_P, _P, _P, _I, _I, _U64, _I, _I]),
    "cpp_naf_sample_and_compute": (_I, [_P, _P, _I, _U64]),
    "cpp_naf_allreduce_grads": (_I, [_P, _P]),
    "cpp_naf_average_params": (_I, [_P, _P]),
    "cpp_naf_dp_train_step": (_I, [_P, _P, _P, _I, _I, _U64, _I]),
    "cpp_naf_clear_numeric_error": (_I, [_P]),
    "cpp_ddpg_dp_status": (_I, [_P, C.POINTER(_I), C.c_char_p, _I]),
    "cpp_naf_dp_status": (_I, [_P, C.POINTER(_I), C.c_char_p, _I]),
    "cpp_naf_create": (_I, [_P, _P, _P, _P, _P, _I, C.POINTER(NafHyper), _PP]),
    "cpp_naf_destroy": (_I, [_P]),
    "cpp_naf_action": (_I, [_P, _P, _I, _I, _P]),
    "cpp_naf_train": (_I, [_P, _P, C.POINTER(_F)]),
    "cpp_naf_debug_values": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "cpp_naf_compute_gradients": (_I, [_P, _P]),
    "cpp_naf_grad_buffer": (_I, [_P, _PP, C.POINTER(_L)]),
    "cpp_naf_apply_gradients": (_I, [_P, _F]),
    "cpp_naf_update_targets": (_I, [_P]),
    "cpp_naf_train_step": (_I, [_P, _P, _I, _I, _P, _U64]),
    "cpp_naf_train_rows": (_I, [_P, _P, _I, _P, C.POINTER(_F)]),
    "cpp_naf_train_rows_async": (_I, [_P, _P, _I, _P, C.POINTER(_U64)]),
    "cpp_naf_loss_wait": (_I, [_P, _U64, C.POINTER(_F)]),
    "cpp_naf_last_stats": (_I, [_P, _P]),
    "cpp_naf_opt_state_size": (_L, [_P]),
    "cpp_naf_get_opt_state": (_I, [_P, _P, _P, _L, C.POINTER(_U64)]),
    "cpp_naf_set_opt_state": (_I, [_P, _P, _P, _L, _U64]),
}


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "cartpoleplusplus_amd: %s not found. This package has no CPU fallback; build the HIP "
            "library first (python -c 'import __graft_entry__ as g; g.build()')." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if lib.cpp_abi_version() != 1:
        raise ImportError("cartpoleplusplus_amd: ABI version mismatch")
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise RuntimeError((lib.cpp_last_error() or b"").decode("utf-8", "replace"))


def ptr(arr):
    """host pointer of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)


def as_state_array(x):
    """States cross the ABI as f16 (what the replay memory stores) or f32 (what the env emits)."""
    x = np.asarray(x)
    if x.dtype == np.float16:
        return np.ascontiguousarray(x), CPP_F16
    return np.ascontiguousarray(x, dtype=np.float32), CPP_F32


class Context(object):
    """One GPU + one HIP stream: the stand-in for the reference's default tf.Session
    (ddpg_cartpole.py:416; `tf.get_default_session()` everywhere else)."""

    def __init__(self, device_id=0, stream=None):
        h = C.c_void_p()
        check(lib.cpp_ctx_create(int(device_id), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle, self.device_id = h, int(device_id)

    def sync(self):
        check(lib.cpp_sync(self.handle))

    def set_precision(self, mode):
        """'fast' (default: two f16 pieces / six bf16 products) or 'exact' (three / nine: every product exact) -- the arithmetic
        contract of the conv kernels on the f16 / bf16 matrix pipes (include/cartpolepp_abi.h, cpp_ctx_set_precision).  Before the
        agents' trainers exist."""
        modes = {"fast": CPP_PRECISION_FAST, "exact": CPP_PRECISION_EXACT, CPP_PRECISION_FAST: CPP_PRECISION_FAST,
                 CPP_PRECISION_EXACT: CPP_PRECISION_EXACT}
        if mode not in modes:
            raise ValueError("precision %r is neither 'fast' nor 'exact'" % (mode,))
        check(lib.cpp_ctx_set_precision(self.handle, modes[mode]))

    def set_route_threshold(self, threshold):
        """whitening scale above which the next training steps run conv1 on the f32-input kernels (nearly constant channels;
        include/cartpolepp_abi.h, cpp_ctx_set_route_threshold); 0 disables the switch."""
        check(lib.cpp_ctx_set_route_threshold(self.handle, float(threshold)))

    def route(self):
        """(conv1 currently on the f32-input kernels?, largest whitening scale of the last finished step)"""
        f, m = C.c_int(), C.c_float()
        check(lib.cpp_ctx_get_route(self.handle, C.byref(f), C.byref(m)))
        return bool(f.value), float(m.value)

    @property
    def precision(self):
        m = C.c_int()
        check(lib.cpp_ctx_get_precision(self.handle, C.byref(m)))
        return "exact" if m.value == CPP_PRECISION_EXACT else "fast"

    def timer_begin(self):
        check(lib.cpp_timer_begin(self.handle))

    def timer_end(self):
        ms = C.c_float()
        check(lib.cpp_timer_end(self.handle, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        check(lib.cpp_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        check(lib.cpp_prof_reset(self.handle))

    def prof_read(self):
        """{kernel name: (total_ms, launches)} for kernels launched while profiling was on."""
        out = {}
        for k in range(lib.cpp_prof_num_kernels()):
            ms, n = C.c_double(), C.c_int64()
            check(lib.cpp_prof_read(self.handle, k, C.byref(ms), C.byref(n)))
            if n.value:
                out[lib.cpp_prof_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def close(self):
        if self.handle:
            lib.cpp_ctx_destroy(self.handle)
            self.handle = None


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx
