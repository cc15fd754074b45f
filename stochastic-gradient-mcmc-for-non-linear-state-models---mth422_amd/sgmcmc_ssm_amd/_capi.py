"""ctypes binding of libpfgrad.so (include/pfgrad.h).

This is the only door from Python to the particle-filter kernels.  There is no CPU
fallback: if the library is missing or no MI355X is visible the calls raise."""
import ctypes as C
import gc
import os
import time

import numpy as np

from . import _build

# ---- enums (include/pfgrad.h) ---------------------------------------------------------
MODEL = {"svm": 0, "garch": 1, "lgssm": 2}
KERNEL = {"prior": 0, "optimal": 1}
SMOOTHER = {"nemeth": 0, "filter": 1, "paris": 2, "nemeth_systematic": 3, "poyiadjis_n2": 4,
            "poyiadjis_n": 5}       # launch-level id only (never in a descriptor): see include/pfgrad.h
STAT = {"score": 0, "suff": 1, "none": 2, "predictive": 3}
DTYPE = {"f64": 0, "f32": 1}
RNG = {"replay": 0, "device": 1, "philox": 1}     # "philox" = alias of "device" (Philox-keyed lanes)
FLAG_GARCH_STATIONARY_PRIOR = 1
FLAG_PARIS_NO_ACCEPT_REJECT = 2
FLAG_PARIS_RAW_STREAM = 4        # paris_stream = the window's whole np.random stream of doubles (pfgrad.h)
FLAG_PARIS_RAW_CARRY = 8         # ... whose first entry is the generator's pending cached Gaussian
MAX_STAT, MAX_THETA, OUT_DOUBLES, MAX_PRED, STAMP_WORDS = 4, 4, 8, 16, 16
STATE_DIM = {"svm": 1, "garch": 2, "lgssm": 1}
STAT_DIM = {"svm": 3, "garch": 4, "lgssm": 4}
THETA_DIM = {"svm": 3, "garch": 4, "lgssm": 4}

PFG_OK, PFG_ERR_INVALID, PFG_ERR_UNSUPPORTED, PFG_ERR_DEVICE, PFG_ERR_NOMEM, PFG_ERR_NUMERIC = 0, -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)


class Problem(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("kernel", C.c_int32), ("smoother", C.c_int32), ("stat", C.c_int32),
        ("dtype", C.c_int32), ("rng", C.c_int32),
        ("N", C.c_int32), ("T", C.c_int32), ("t1", C.c_int32), ("tL", C.c_int32),
        ("flags", C.c_uint32), ("reserved", C.c_int32),
        ("lambduh", C.c_double), ("prior_mean", C.c_double), ("prior_var", C.c_double),
        ("y", _dp), ("weights", _dp), ("theta", _dp),
        ("z0", _dp), ("u", _dp), ("z", _dp),
        ("seed", C.c_uint64), ("stream", C.c_uint64),
        ("init_x", _dp), ("init_logw", _dp), ("init_stats", _dp),
        ("Ntilde", C.c_int32), ("max_accept_reject", C.c_int32),
        ("paris_idx_u", _dp), ("paris_acc_u", _dp), ("paris_man_u", _dp),
        ("num_steps_ahead", C.c_int32), ("elementwise", C.c_int32),
        ("pred_z", _dp),
        ("paris_stream", _dp), ("paris_stream_len", C.c_int64),
        ("paris_manual_threshold", C.c_int32), ("reserved2", C.c_int32),
        ("step", C.c_uint64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("mean_stat", C.c_double * MAX_STAT), ("loglik", C.c_double),
        ("x_T", _dp), ("logw_T", _dp), ("stats_T", _dp),
        ("trace_x", _dp), ("trace_logw", _dp), ("trace_stats", _dp), ("trace_ll", _dp),
        ("status", C.c_int32), ("paris_carry_back", C.c_int32),
        ("trace_anc", C.POINTER(C.c_int32)),
        ("pred", C.c_double * MAX_PRED),
        ("rec_u", C.POINTER(C.c_uint32)), ("rec_z", _dp), ("rec_z0", _dp),
        ("rec_ud", _dp),
        ("ew_mean", _dp), ("ew_stats", _dp),
        ("paris_consumed", C.c_int64),
    ]


# numpy mirrors of pfg_problem / pfg_result for the vectorised marshalling of large device-generator batches
# (Context._run_batch_plain); layouts are asserted against the ctypes structures at import
PROBLEM_DTYPE = np.dtype([
    ("model", "i4"), ("kernel", "i4"), ("smoother", "i4"), ("stat", "i4"), ("dtype", "i4"), ("rng", "i4"),
    ("N", "i4"), ("T", "i4"), ("t1", "i4"), ("tL", "i4"), ("flags", "u4"), ("reserved", "i4"),
    ("lambduh", "f8"), ("prior_mean", "f8"), ("prior_var", "f8"),
    ("y", "u8"), ("weights", "u8"), ("theta", "u8"), ("z0", "u8"), ("u", "u8"), ("z", "u8"),
    ("seed", "u8"), ("stream", "u8"),
    ("init_x", "u8"), ("init_logw", "u8"), ("init_stats", "u8"),
    ("Ntilde", "i4"), ("max_accept_reject", "i4"),
    ("paris_idx_u", "u8"), ("paris_acc_u", "u8"), ("paris_man_u", "u8"),
    ("num_steps_ahead", "i4"), ("elementwise", "i4"), ("pred_z", "u8"),
    ("paris_stream", "u8"), ("paris_stream_len", "i8"), ("paris_manual_threshold", "i4"), ("reserved2", "i4"),
    ("step", "u8")], align=True)
RESULT_DTYPE = np.dtype([
    ("mean_stat", "f8", (MAX_STAT,)), ("loglik", "f8"),
    ("x_T", "u8"), ("logw_T", "u8"), ("stats_T", "u8"),
    ("trace_x", "u8"), ("trace_logw", "u8"), ("trace_stats", "u8"), ("trace_ll", "u8"),
    ("status", "i4"), ("paris_carry_back", "i4"), ("trace_anc", "u8"),
    ("pred", "f8", (MAX_PRED,)),
    ("rec_u", "u8"), ("rec_z", "u8"), ("rec_z0", "u8"), ("rec_ud", "u8"), ("ew_mean", "u8"), ("ew_stats", "u8"),
    ("paris_consumed", "i8")], align=True)
assert PROBLEM_DTYPE.itemsize == C.sizeof(Problem) and RESULT_DTYPE.itemsize == C.sizeof(Result)
assert all(PROBLEM_DTYPE.fields[n][1] == getattr(Problem, n).offset for n, _ in Problem._fields_)
assert all(RESULT_DTYPE.fields[n][1] == getattr(Result, n).offset for n, _ in Result._fields_)


class PriorHyper(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "df_Qinv", "scale_Qinv", "df_Rinv", "scale_Rinv",
        "mean_A", "var_col_A", "mean_C", "var_col_C",
        "scale_mu", "shape_mu", "alpha_phi", "beta_phi", "alpha_lambduh", "beta_lambduh")]


# numpy mirror of pfg_dev_problem (device descriptors are built on the host as a structured
# array and uploaded; all pointer fields are device addresses)
DEV_PROBLEM_DTYPE = np.dtype([
    ("y", "u8"), ("weights", "u8"), ("theta", "u8"),
    ("z0", "u8"), ("u", "u8"), ("z", "u8"),
    ("init_x", "u8"), ("init_logw", "u8"), ("init_stats", "u8"),
    ("out", "u8"),
    ("final_x", "u8"), ("final_logw", "u8"), ("final_stats", "u8"),
    ("trace_x", "u8"), ("trace_logw", "u8"), ("trace_stats", "u8"), ("trace_ll", "u8"),
    ("step_ctr", "u8"), ("scratch", "u8"),
    ("prior_mean", "f8"), ("prior_var", "f8"), ("lambduh", "f8"),
    ("seed", "u8"), ("stream", "u8"),
    ("T", "i4"), ("t1", "i4"), ("tL", "i4"), ("N", "i4"),
    ("smoother", "i4"), ("stat", "i4"), ("flags", "u4"), ("reserved", "i4"),
    ("paris_idx_u", "u8"), ("paris_acc_u", "u8"), ("paris_man_u", "u8"),
    ("Ntilde", "i4"), ("max_accept_reject", "i4"),
    ("trace_anc", "u8"),
    ("pred_z", "u8"), ("pred_out", "u8"), ("pred_scratch", "u8"),
    ("num_steps_ahead", "i4"), ("reserved3", "i4"),
    ("rec_u", "u8"), ("rec_z", "u8"), ("rec_z0", "u8"),
    ("rec_ud", "u8"), ("trace_paris_J", "u8"),
    ("paris_stream", "u8"), ("paris_stream_len", "i8"), ("paris_consumed", "u8"),
    ("paris_manual_threshold", "i4"), ("reserved4", "i4"),
    ("stamps", "u8"),
], align=True)

EXPORTS = ("pfg_version", "pfg_struct_size", "pfg_create", "pfg_destroy", "pfg_last_error", "pfg_run", "pfg_run_batch",
           "pfg_ctx_stream", "pfg_launch_device", "pfg_launch_device_smoother", "pfg_scratch_bytes", "pfg_variant_name", "pfg_synchronize",
           "pfg_sgld_update_device", "pfg_sghmc_update_device", "pfg_imq_ksd", "pfg_sample_windows_device",
           "pfg_last_variant", "pfg_legacy_streams", "pfg_host_register", "pfg_host_unregister",
           "pfg_launch_device_traced", "pfg_last_traced", "pfg_launch_device_grid", "pfg_launch_device_grid_phase",
           "pfg_launch_device_grid_smoother")

_lib = None


class PfgError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libpfgrad error {0}: {1}".format(code, message))
        self.code = code
        self.message = message


def library_path():
    """In-tree csrc/libpfgrad.so; PFGRAD_LIB=<path> substitutes another build (A/B timing)."""
    return os.environ.get("PFGRAD_LIB") or _build.LIB_PATH


def load_library():
    """dlopen the in-tree libpfgrad.so (never builds implicitly on a GPU box: build() does)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libpfgrad.so is not built ({0}). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python -m sgmcmc_ssm_amd._build`; there is no CPU fallback.".format(path))
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (ROCm 7.0 here)
    # while libpfgrad.so was linked against the system one (ROCm 7.2, same SONAME).  Whichever
    # is loaded first serves both; loading the system runtime first breaks torch.cuda.  So when
    # torch is installed, let it load its runtime before we dlopen (torch is only plumbing:
    # device memory, streams, torch.distributed -- no torch type crosses the C ABI).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.pfg_version.restype = C.c_int
    lib.pfg_struct_size.argtypes = [C.c_int]
    lib.pfg_struct_size.restype = C.c_int
    sizes = (C.sizeof(Problem), C.sizeof(Result), DEV_PROBLEM_DTYPE.itemsize, C.sizeof(PriorHyper))
    for which, mine in enumerate(sizes):
        if lib.pfg_struct_size(which) != mine:
            raise RuntimeError("ABI mismatch: struct {0} is {1} bytes in libpfgrad.so, {2} in the binding"
                               .format(which, lib.pfg_struct_size(which), mine))
    lib.pfg_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.pfg_create.restype = C.c_int
    lib.pfg_destroy.argtypes = [C.c_void_p]
    lib.pfg_destroy.restype = None
    lib.pfg_last_error.argtypes = [C.c_void_p]
    lib.pfg_last_error.restype = C.c_char_p
    lib.pfg_run.argtypes = [C.c_void_p, C.POINTER(Problem), C.POINTER(Result)]
    lib.pfg_run.restype = C.c_int
    lib.pfg_run_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(Problem), C.POINTER(Result)]
    lib.pfg_run_batch.restype = C.c_int
    lib.pfg_launch_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]
    lib.pfg_launch_device.restype = C.c_int
    lib.pfg_ctx_stream.argtypes = [C.c_void_p]
    lib.pfg_ctx_stream.restype = C.c_void_p
    lib.pfg_launch_device_smoother.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]
    lib.pfg_launch_device_smoother.restype = C.c_int
    lib.pfg_launch_device_traced.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]
    lib.pfg_launch_device_traced.restype = C.c_int
    lib.pfg_launch_device_grid.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]
    lib.pfg_launch_device_grid.restype = C.c_int
    lib.pfg_launch_device_grid_phase.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]
    lib.pfg_launch_device_grid_phase.restype = C.c_int
    lib.pfg_launch_device_grid_smoother.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.c_void_p, C.c_void_p]
    lib.pfg_launch_device_grid_smoother.restype = C.c_int
    lib.pfg_last_traced.argtypes = [C.c_void_p]
    lib.pfg_last_traced.restype = C.c_int
    lib.pfg_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pfg_scratch_bytes.restype = C.c_int64
    lib.pfg_variant_name.argtypes = [C.c_int] * 5
    lib.pfg_variant_name.restype = C.c_char_p
    lib.pfg_legacy_streams.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.pfg_legacy_streams.restype = C.c_int
    lib.pfg_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.pfg_host_register.restype = C.c_int
    lib.pfg_host_unregister.argtypes = [C.c_void_p]
    lib.pfg_host_unregister.restype = C.c_int
    lib.pfg_last_variant.argtypes = [C.c_void_p]
    lib.pfg_last_variant.restype = C.c_char_p
    lib.pfg_synchronize.argtypes = [C.c_void_p]
    lib.pfg_synchronize.restype = C.c_int
    lib.pfg_sgld_update_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.POINTER(PriorHyper), C.c_double, C.c_double, C.c_uint64,
                                           C.c_uint64, C.c_void_p, C.c_void_p]
    lib.pfg_sgld_update_device.restype = C.c_int
    lib.pfg_sghmc_update_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(PriorHyper), C.c_double, C.c_double, C.c_double,
                                            C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.pfg_sghmc_update_device.restype = C.c_int
    lib.pfg_imq_ksd.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, _dp]
    lib.pfg_imq_ksd.restype = C.c_int
    lib.pfg_sample_windows_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p,
                                              C.c_void_p]
    lib.pfg_sample_windows_device.restype = C.c_int
    _lib = lib
    return lib


def legacy_streams(random_state, N, T, z0, u, z, threads=0):
    """Fill z0 (N,), u (T,N), z (T,N) with what `random_state` (np.random or a RandomState) would return
    for  z0 = normal(size=N); for t: u[t] = random_sample(N); z[t] = normal(size=N)  -- natively, bit for
    bit -- and advance its state accordingly."""
    lib = load_library()
    rs = random_state            # np.random (module: get_state / set_state act on the global generator) or a RandomState
    name, key, pos, has_gauss, gauss = rs.get_state()
    if name != "MT19937":
        raise ValueError("legacy streams need an MT19937 RandomState")
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    pos_c, hg_c, g_c = C.c_int32(int(pos)), C.c_int32(int(has_gauss)), C.c_double(float(gauss))
    for a in (z0, u, z):
        if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
            raise ValueError("stream buffers must be C-contiguous float64")
    rc = lib.pfg_legacy_streams(key.ctypes.data_as(C.c_void_p), C.byref(pos_c), C.byref(hg_c), C.byref(g_c), int(N), int(T),
                                z0.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                                int(threads))
    if rc != 0:
        raise PfgError(rc, "pfg_legacy_streams failed")
    rs.set_state((name, key, pos_c.value, hg_c.value, g_c.value))


def host_register(a):
    """Page-lock the ndarray `a` (pfg_host_register): pfg_run_batch then stages it by DMA from where it lies.
    The caller keeps `a` alive until host_unregister(a).  Returns False when the runtime refuses (no GPU, limits)."""
    return load_library().pfg_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes) == 0


def host_unregister(a):
    return load_library().pfg_host_unregister(a.ctypes.data_as(C.c_void_p)) == 0


_OPTIONAL_ARRAYS = ("weights", "z0", "u", "z", "init_x", "init_logw", "init_stats",
                    "paris_idx_u", "paris_acc_u", "paris_man_u", "pred_z", "paris_stream")
_NO_ARRAYS = dict.fromkeys(("y", "theta") + _OPTIONAL_ARRAYS)


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class Context:
    """One pfg_ctx (one per process / GPU). Not thread-safe."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.pfg_create(C.byref(h), int(device_id))
        if rc != 0:
            msg = self.lib.pfg_last_error(None).decode()
            raise PfgError(rc, msg + " (the HIP particle-filter path needs an MI355X; there is no CPU fallback)")
        self.handle = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pfg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.pfg_last_error(self.handle).decode()
            if rc == PFG_ERR_NUMERIC:
                raise ValueError(msg)
            if rc == PFG_ERR_UNSUPPORTED:
                raise NotImplementedError(msg)
            if rc == PFG_ERR_INVALID:
                raise ValueError(msg)
            raise PfgError(rc, msg)

    # ---- host-buffer path ----------------------------------------------------------------
    def run_batch(self, problems, want_final=False, want_trace=False, want_draws=False, want_elementwise=False):
        """problems: list of dicts with keys
             model, kernel, smoother, stat, dtype, rng (strings), N, T (implied by y), t1, tL,
             lambduh, prior_mean, prior_var, flags, y, weights, theta, z0,u,z | seed,stream,
             init_x, init_logw, init_stats
           returns list of dicts(mean_stat, loglik[, x_t, log_weights, statistics, all_*])."""
        B = len(problems)
        if B == 0:
            return []
        if B >= 64 and not (want_final or want_trace or want_draws or want_elementwise):
            outs = self._run_batch_plain(problems)
            if outs is not None:
                return outs
        ps = (Problem * B)()
        rs = (Result * B)()
        gc_was_on = gc.isenabled() and B >= 256
        if gc_was_on:
            gc.disable()        # thousands of small containers below: generation-2 sweeps made this loop superlinear in B
        try:
            keep, outs = self._marshal(problems, ps, rs, want_final, want_trace, want_draws, want_elementwise)
        finally:
            if gc_was_on:
                gc.enable()     # also when a problem is refused (ValueError): the collector must not stay off
        t_call = time.perf_counter()
        rc = self.lib.pfg_run_batch(self.handle, B, ps, rs)
        self.last_call_seconds = time.perf_counter() - t_call      # the C call alone (pack, H2D, launch, D2H), without this marshalling
        self._check(rc)
        for b, o in enumerate(outs):
            # the device record is STAT_DIM[model] wide; sufficient statistics use 3 columns
            h = 3 if problems[b].get("stat", "score") != "score" else STAT_DIM[problems[b]["model"]]
            o["mean_stat"] = np.array(rs[b].mean_stat[:h])
            o["loglik"] = float(rs[b].loglik)
            if problems[b].get("stat", "score") == "predictive":
                o["predictive"] = np.array(rs[b].pred[:int(problems[b].get("num_steps_ahead", 0)) + 1])
            if problems[b].get("paris_stream", None) is not None:
                o["paris_consumed"] = int(rs[b].paris_consumed)
                o["paris_carry_back"] = int(rs[b].paris_carry_back)
            for name in ("statistics", "all_statistics"):
                if name in o:
                    o[name] = o[name][..., :h]
        del keep
        return outs

    def _marshal(self, problems, ps, rs, want_final, want_trace, want_draws, want_elementwise):
        """Fill the ctypes problem / result arrays from the problem dicts; returns (arrays kept alive, output dicts)."""
        keep = []           # keep numpy buffers alive for the duration of the call
        outs = []
        for b, q in enumerate(problems):
            model = q["model"]
            ns, h = STATE_DIM[model], STAT_DIM[model]
            y = _as_f64(q["y"]).reshape(-1)
            T, N = y.shape[0], int(q["N"])
            p = ps[b]
            p.model, p.kernel = MODEL[model], KERNEL[q["kernel"]]
            p.smoother, p.stat = SMOOTHER[q.get("smoother", "nemeth")], STAT[q.get("stat", "score")]
            p.dtype, p.rng = DTYPE[q.get("dtype", "f64")], RNG[q.get("rng", "replay")]
            p.N, p.T = N, T
            p.t1 = int(q.get("t1", 0))
            tL = q.get("tL", None)
            p.tL = T if tL is None else int(tL)
            p.flags = int(q.get("flags", 0))
            p.lambduh = float(q.get("lambduh", 1.0))
            p.prior_mean = float(q.get("prior_mean", 0.0))
            p.prior_var = float(q.get("prior_var", 1.0))
            theta = _as_f64(q["theta"]).reshape(-1)
            arrs = _NO_ARRAYS.copy()
            arrs["y"], arrs["theta"] = y, theta
            p.y, p.theta = _ptr(y), _ptr(theta)
            p.Ntilde = int(q.get("Ntilde", 2))
            p.max_accept_reject = int(q.get("max_accept_reject", 0))
            p.num_steps_ahead = int(q.get("num_steps_ahead", 0))
            for name in _OPTIONAL_ARRAYS:               # absent arrays stay NULL in the zero-initialised struct
                v = q.get(name, None)
                if v is not None:
                    a = arrs[name] = _as_f64(v).reshape(-1)
                    setattr(p, name, _ptr(a))
            if arrs["weights"] is not None and arrs["weights"].shape[0] < p.tL - p.t1:
                raise ValueError("weights shorter than tL - t1")
            if p.rng == RNG["replay"] and not (p.flags & FLAG_PARIS_RAW_STREAM):
                if arrs["u"] is None or arrs["z"] is None or arrs["u"].shape[0] != T * N or arrs["z"].shape[0] != T * N:
                    raise ValueError("replay streams u, z must have T*N entries")
                if arrs["init_x"] is None and (arrs["z0"] is None or arrs["z0"].shape[0] != N):
                    raise ValueError("replay stream z0 must have N entries")
            p.seed = int(q.get("seed", 0)) & 0xFFFFFFFFFFFFFFFF
            p.stream = int(q.get("stream", 0)) & 0xFFFFFFFFFFFFFFFF
            p.step = int(q.get("step", 0)) & 0xFFFFFFFFFFFFFFFF
            keep.append(arrs)
            o = {}
            r = rs[b]
            if arrs["paris_stream"] is not None:
                p.paris_stream_len = arrs["paris_stream"].shape[0]
                p.paris_manual_threshold = int(q.get("paris_manual_threshold", 0))
            elif p.smoother == SMOOTHER["paris"] and p.rng == RNG["replay"]:
                pool = T * p.Ntilde * p.max_accept_reject * N
                for name, need in (("paris_idx_u", pool), ("paris_acc_u", pool), ("paris_man_u", T * p.Ntilde * N)):
                    if need and (arrs[name] is None or arrs[name].shape[0] != need):
                        raise ValueError("{0} must have {1} entries".format(name, need))
            if p.stat == STAT["predictive"] and p.rng == RNG["replay"] and model != "lgssm":
                need = T * (p.num_steps_ahead + 1) * N
                if need and (arrs["pred_z"] is None or arrs["pred_z"].shape[0] != need):
                    raise ValueError("pred_z must have T*(num_steps_ahead+1)*N = {0} entries".format(need))
            is_filter = p.smoother == SMOOTHER["filter"]
            if want_final or want_trace:
                o["x_t"] = np.zeros((N, ns))
                o["log_weights"] = np.zeros(N)
                r.x_T, r.logw_T = _ptr(o["x_t"]), _ptr(o["log_weights"])
                if not is_filter:
                    o["statistics"] = np.zeros((N, h))
                    r.stats_T = _ptr(o["statistics"])
            if want_elementwise:
                # elementwise sufficient statistics of the window (pf_latent_var_distr): device pass
                L = min(p.tL, T) - p.t1
                p.elementwise = 1
                o["ew_mean"] = np.zeros(3 * max(L, 0))
                r.ew_mean = _ptr(o["ew_mean"])
                if want_final:
                    o["ew_stats"] = np.zeros((N, 3 * max(L, 0)))
                    r.ew_stats = _ptr(o["ew_stats"])
            if want_trace:
                o["all_x_t"] = np.zeros((T + 1, N, ns))
                o["all_log_weights"] = np.zeros((T + 1, N))
                o["all_loglikelihood_estimate"] = np.zeros(T + 1)
                o["all_ancestors"] = np.zeros((T, N), dtype=np.int32)
                r.trace_anc = o["all_ancestors"].ctypes.data_as(C.POINTER(C.c_int32))
                r.trace_x, r.trace_logw = _ptr(o["all_x_t"]), _ptr(o["all_log_weights"])
                r.trace_ll = _ptr(o["all_loglikelihood_estimate"])
                if not is_filter:
                    o["all_statistics"] = np.zeros((T + 1, N, h))
                    r.trace_stats = _ptr(o["all_statistics"])
                if want_draws:
                    # test instrumentation: the DEVICE generator's draws of this very launch
                    o["rec_u"] = np.zeros((T, N), dtype=np.uint32)
                    o["rec_z"] = np.zeros((T, N))
                    o["rec_z0"] = np.zeros(N)
                    r.rec_u = o["rec_u"].ctypes.data_as(C.POINTER(C.c_uint32))
                    r.rec_z, r.rec_z0 = _ptr(o["rec_z"]), _ptr(o["rec_z0"])
                    o["rec_ud"] = np.zeros((T, N))
                    r.rec_ud = _ptr(o["rec_ud"])
            outs.append(o)
        return keep, outs

    def _run_batch_plain(self, problems):
        """Large batches that only ask for (mean_stat, loglik) from the device generator: the descriptors are
        filled column by column into a structured array (a ctypes attribute store per field and problem cost
        19 us per problem, four times the kernel's share of a 12288-window launch).  None = not eligible."""
        B = len(problems)
        blocked = ("z0", "u", "z", "init_x", "init_logw", "init_stats", "paris_idx_u", "paris_acc_u", "paris_man_u", "pred_z", "paris_stream")
        for q in problems:
            if RNG[q.get("rng", "replay")] != RNG["device"] or q.get("stat", "score") == "predictive":
                return None
            for name in blocked:
                if q.get(name, None) is not None:
                    return None
        conv = {}                                   # id(array) -> (converted array, address, length): shared inputs convert once

        def prep(a):
            k = id(a)
            got = conv.get(k)
            if got is None:
                b = _as_f64(a).reshape(-1)
                got = conv[k] = (b, b.ctypes.data, b.shape[0], a)          # `a` kept alive: its id must stay unique
            return got
        ps = np.zeros(B, PROBLEM_DTYPE)
        ys = [prep(q["y"]) for q in problems]
        ths = [prep(q["theta"]) for q in problems]
        ws = [None if q.get("weights", None) is None else prep(q["weights"]) for q in problems]
        ps["y"] = [t[1] for t in ys]
        ps["T"] = T = np.array([t[2] for t in ys], dtype=np.int64)
        ps["theta"] = [t[1] for t in ths]
        ps["weights"] = [0 if t is None else t[1] for t in ws]
        ps["model"] = [MODEL[q["model"]] for q in problems]
        ps["kernel"] = [KERNEL[q["kernel"]] for q in problems]
        ps["smoother"] = [SMOOTHER[q.get("smoother", "nemeth")] for q in problems]
        ps["stat"] = [STAT[q.get("stat", "score")] for q in problems]
        ps["dtype"] = [DTYPE[q.get("dtype", "f64")] for q in problems]
        ps["rng"] = RNG["device"]
        ps["N"] = [q["N"] for q in problems]
        ps["t1"] = t1 = np.array([q.get("t1", 0) for q in problems], dtype=np.int64)
        tL = [q.get("tL", None) for q in problems]
        ps["tL"] = tL = np.array([t if v is None else v for v, t in zip(tL, T)], dtype=np.int64)
        ps["flags"] = [q.get("flags", 0) for q in problems]
        ps["lambduh"] = [q.get("lambduh", 1.0) for q in problems]
        ps["prior_mean"] = [q.get("prior_mean", 0.0) for q in problems]
        ps["prior_var"] = [q.get("prior_var", 1.0) for q in problems]
        ps["Ntilde"] = [q.get("Ntilde", 2) for q in problems]
        ps["max_accept_reject"] = [q.get("max_accept_reject", 0) for q in problems]
        ps["seed"] = np.array([int(q.get("seed", 0)) & 0xFFFFFFFFFFFFFFFF for q in problems], dtype=np.uint64)
        ps["stream"] = np.array([int(q.get("stream", 0)) & 0xFFFFFFFFFFFFFFFF for q in problems], dtype=np.uint64)
        ps["step"] = np.array([int(q.get("step", 0)) & 0xFFFFFFFFFFFFFFFF for q in problems], dtype=np.uint64)
        wlen = np.array([1 << 62 if t is None else t[2] for t in ws], dtype=np.int64)
        if np.any(wlen < tL - t1):
            raise ValueError("weights shorter than tL - t1")
        rs = np.zeros(B, RESULT_DTYPE)
        t_call = time.perf_counter()
        rc = self.lib.pfg_run_batch(self.handle, B, C.cast(ps.ctypes.data, C.POINTER(Problem)), C.cast(rs.ctypes.data, C.POINTER(Result)))
        self.last_call_seconds = time.perf_counter() - t_call
        self._check(rc)
        mean, ll = rs["mean_stat"], rs["loglik"]
        outs = []
        for b, q in enumerate(problems):
            # the device record is STAT_DIM[model] wide; sufficient statistics use 3 columns
            h = 3 if q.get("stat", "score") != "score" else STAT_DIM[q["model"]]
            outs.append({"mean_stat": mean[b, :h], "loglik": float(ll[b])})
        return outs

    # ---- resident path ---------------------------------------------------------------------
    def stream(self):
        """The context's own hipStream_t (as an integer handle)."""
        return self.lib.pfg_ctx_stream(self.handle) or 0

    GRID_PHASE_INIT, GRID_PHASE_FINISH = -2, -3

    def launch_device_grid_phase(self, model, kernel, dtype, rng, n_max, phase, B, dev_probs_ptr, stream_ptr=0):
        """One piece of a whole-GPU window: phase = GRID_PHASE_INIT, a timestep t >= 0, or GRID_PHASE_FINISH."""
        self._check(self.lib.pfg_launch_device_grid_phase(
            self.handle, MODEL[model], KERNEL[kernel], DTYPE[dtype], RNG[rng], int(n_max), int(phase), int(B),
            C.c_void_p(dev_probs_ptr), C.c_void_p(int(stream_ptr))))

    GRID_PHASE_ALL = -1

    def launch_device_grid_smoother(self, model, kernel, dtype, rng, smoother, n_max, T_max, phase, B, dev_probs_ptr, stream_ptr=0):
        """The whole-GPU window (phase = GRID_PHASE_ALL) or one piece of it with the batch's smoother stated: 'nemeth' /
        'filter' as launch_device_grid[_phase]; 'poyiadjis_n' (every window NEMETH, lambduh = 1, score) runs the score-only
        twin of the device-generator timestep kernel."""
        self._check(self.lib.pfg_launch_device_grid_smoother(
            self.handle, MODEL[model], KERNEL[kernel], DTYPE[dtype], RNG[rng], SMOOTHER[smoother], int(n_max), int(T_max),
            int(phase), int(B), C.c_void_p(dev_probs_ptr), C.c_void_p(int(stream_ptr))))

    def launch_device_smoother(self, model, kernel, dtype, rng, smoother, n_max, B, dev_probs_ptr, stream_ptr=0):
        self._check(self.lib.pfg_launch_device_smoother(
            self.handle, MODEL[model], KERNEL[kernel], DTYPE[dtype], RNG[rng], SMOOTHER[smoother], int(n_max),
            int(B), C.c_void_p(dev_probs_ptr), C.c_void_p(int(stream_ptr))))

    def launch_device_traced(self, model, kernel, dtype, rng, smoother, n_max, B, dev_probs_ptr, stream_ptr=0):
        """The launch that honours the trace_* / rec_* buffers of the descriptors (pfg_launch_device_traced)."""
        self._check(self.lib.pfg_launch_device_traced(
            self.handle, MODEL[model], KERNEL[kernel], DTYPE[dtype], RNG[rng], SMOOTHER[smoother], int(n_max),
            int(B), C.c_void_p(dev_probs_ptr), C.c_void_p(int(stream_ptr))))

    def last_traced(self):
        return bool(self.lib.pfg_last_traced(self.handle))

    def launch_device_grid(self, model, kernel, dtype, rng, n_max, T_max, B, dev_probs_ptr, stream_ptr=0):
        """Whole-GPU windows (N > 16384: one launch per timestep, pfg_launch_device_grid); T_max = the largest T."""
        self._check(self.lib.pfg_launch_device_grid(
            self.handle, MODEL[model], KERNEL[kernel], DTYPE[dtype], RNG[rng], int(n_max), int(T_max), int(B),
            C.c_void_p(dev_probs_ptr), C.c_void_p(int(stream_ptr))))

    def launch_device(self, model, kernel, dtype, rng, n_max, B, dev_probs_ptr, stream_ptr=0):
        """stream_ptr: a hipStream_t handle used as is (0 = HIP's default stream, which is also
        torch's default `torch.cuda.current_stream().cuda_stream`)."""
        self._check(self.lib.pfg_launch_device(self.handle, MODEL[model], KERNEL[kernel], DTYPE[dtype],
                                               RNG[rng], int(n_max), int(B), C.c_void_p(dev_probs_ptr),
                                               C.c_void_p(int(stream_ptr))))

    def sample_windows_device(self, B, dev_probs_ptr, y_ptr, weights_table_ptr, T, S, buffer, strict, seed,
                              chain_offset=0, step_ctr_ptr=None, stream_ptr=0):
        self._check(self.lib.pfg_sample_windows_device(
            self.handle, int(B), C.c_void_p(dev_probs_ptr), C.c_void_p(y_ptr), C.c_void_p(weights_table_ptr or 0),
            int(T), int(S), int(buffer), int(bool(strict)), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
            C.c_uint64(int(chain_offset)), C.c_void_p(step_ctr_ptr or 0), C.c_void_p(int(stream_ptr))))

    def sgld_update_device(self, model, B, theta_ptr, outs_ptr, hyper, epsilon, Tscale, seed,
                           chain_offset=0, step_ctr_ptr=None, stream_ptr=0):
        self._check(self.lib.pfg_sgld_update_device(
            self.handle, MODEL[model], int(B), C.c_void_p(theta_ptr), C.c_void_p(outs_ptr), C.byref(hyper),
            float(epsilon), float(Tscale), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
            C.c_uint64(int(chain_offset)), C.c_void_p(step_ctr_ptr) if step_ctr_ptr else None,
            C.c_void_p(int(stream_ptr))))

    def scratch_bytes(self, model, dtype, rng, N):
        return int(self.lib.pfg_scratch_bytes(MODEL[model], DTYPE[dtype], RNG[rng], int(N)))

    def imq_ksd(self, x, gradlogp, c=1.0, beta=0.5):
        x, g = _as_f64(x), _as_f64(gradlogp)
        if x.ndim != 2 or x.shape != g.shape:
            raise ValueError("x and gradlogp dimensions do not match")
        out = C.c_double()
        self._check(self.lib.pfg_imq_ksd(self.handle, x.shape[0], x.shape[1], _ptr(x), _ptr(g), float(c),
                                         float(beta), C.byref(out)))
        return float(out.value)

    def sghmc_update_device(self, model, B, theta_ptr, momentum_ptr, outs_ptr, hyper, epsilon, alpha, Tscale, seed,
                            chain_offset=0, step_ctr_ptr=None, stream_ptr=0):
        self._check(self.lib.pfg_sghmc_update_device(
            self.handle, MODEL[model], int(B), C.c_void_p(theta_ptr), C.c_void_p(momentum_ptr), C.c_void_p(outs_ptr),
            C.byref(hyper), float(epsilon), float(alpha), float(Tscale), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
            C.c_uint64(int(chain_offset)), C.c_void_p(step_ctr_ptr) if step_ctr_ptr else None,
            C.c_void_p(int(stream_ptr))))

    def variant_name(self, model, kernel, dtype, rng, n_max):
        return self.lib.pfg_variant_name(MODEL[model], KERNEL[kernel], DTYPE[dtype], RNG[rng], int(n_max)).decode()

    def last_variant(self):
        """Tag of the kernel variant the latest launch through this context ran."""
        return self.lib.pfg_last_variant(self.handle).decode()

    def synchronize(self):
        self._check(self.lib.pfg_synchronize(self.handle))


_default_ctx = {}


def default_context(device_id=None):
    """Process-wide context for `device_id` (default: LOCAL_RANK or 0)."""
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0"))
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]
