"""ctypes binding of libblsgpu.so (include/blsgpu.h) -- the HIP engine.

This is the product's only compute back-end for the pairing path.  There is no
CPU fallback: when the library or a GPU is missing, `engine()` raises.
"""
import ctypes
import os
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libblsgpu.so")

SYMBOLS = (
    "blsgpu_version", "blsgpu_last_error", "blsgpu_ctx_create", "blsgpu_ctx_destroy",
    "blsgpu_ctx_reserve", "blsgpu_ctx_set_mp_threshold", "blsgpu_ctx_set_miller_wide_max", "blsgpu_pairing_multi", "blsgpu_pairing_multi_dev",
    "blsgpu_miller_product_dev", "blsgpu_final_exp_product_dev", "blsgpu_final_exp",
    "blsgpu_timing_enable", "blsgpu_timing_read", "blsgpu_timing_mad_probe", "blsgpu_timing_mark",
    "blsgpu_g1_msm", "blsgpu_g2_msm", "blsgpu_g1_msm_dev", "blsgpu_g2_msm_dev",
    "blsgpu_final_exp_batch", "blsgpu_pairing_multi_batch", "blsgpu_pairing_multi_batch_dev",
    "blsgpu_map_to_g2", "blsgpu_map_to_g2_dev",
    "blsgpu_miller_product_batch_dev", "blsgpu_final_exp_product_batch_dev",
    "blsgpu_g1_decompress", "blsgpu_g2_decompress", "blsgpu_g1_decompress_dev", "blsgpu_g2_decompress_dev",
    "blsgpu_hash_to_g2", "blsgpu_hash_to_g2_dev",
    "blsgpu_miller_loop_batch", "blsgpu_miller_loop_batch_dev", "blsgpu_line_eval_batch", "blsgpu_ctx_trim",
    "blsgpu_fq12_op_batch", "blsgpu_fq12_pow_batch", "blsgpu_ctx_set_mp3_threshold", "blsgpu_ctx_set_ls_threshold", "blsgpu_ctx_set_ls_teams", "blsgpu_ctx_set_bulk_event", "blsgpu_ctx_set_fexp_team_threshold", "blsgpu_ctx_set_fexp_trace", "blsgpu_ctx_set_fexpw_stamps", "blsgpu_debug_read_lines",
    "blsgpu_ctx_workspace_bytes", "blsgpu_verify_pipeline", "blsgpu_verify_pipeline_dev",
)

_lib = None
_lock = threading.Lock()


class BlsGpuError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libblsgpu.so and declare the prototypes (no GPU call is made)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        p = path or os.environ.get("BLSGPU_LIBRARY", _LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime.  If torch is going to be
        # used in this process (device buffers, streams, RCCL) it has to be
        # loaded first so that libblsgpu.so binds to the same runtime; loaded
        # the other way round torch no longer sees the GPU.
        if "torch" not in sys.modules and not os.environ.get("BLSGPU_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(p):
            raise BlsGpuError("libblsgpu.so not found at %s -- run __graft_entry__.build() "
                              "(there is no CPU fallback)" % p)
        L = ctypes.CDLL(p)
        vp, sz, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p
        L.blsgpu_version.restype = cp
        L.blsgpu_last_error.restype = cp
        L.blsgpu_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.blsgpu_ctx_destroy.argtypes = [vp]
        L.blsgpu_ctx_destroy.restype = None
        L.blsgpu_ctx_reserve.argtypes = [vp, sz]
        L.blsgpu_ctx_set_mp_threshold.argtypes = [vp, sz]
        L.blsgpu_ctx_set_mp3_threshold.argtypes = [vp, sz]
        L.blsgpu_timing_mad_probe.argtypes = [vp, ctypes.c_double, ctypes.POINTER(ctypes.c_double), vp]
        L.blsgpu_timing_mark.argtypes = [vp, ctypes.c_uint, vp]
        L.blsgpu_ctx_set_miller_wide_max.argtypes = [vp, sz]
        L.blsgpu_ctx_set_ls_threshold.argtypes = [vp, sz, sz]
        L.blsgpu_ctx_set_ls_teams.argtypes = [vp, sz]
        L.blsgpu_ctx_set_bulk_event.argtypes = [vp, vp]
        L.blsgpu_ctx_set_fexp_team_threshold.argtypes = [vp, sz]
        L.blsgpu_ctx_set_fexp_trace.argtypes = [vp, vp]
        L.blsgpu_ctx_set_fexpw_stamps.argtypes = [vp, vp]
        L.blsgpu_debug_read_lines.argtypes = [vp, vp, sz]
        L.blsgpu_ctx_workspace_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
        L.blsgpu_verify_pipeline.argtypes = [vp, cp, cp, cp, sz, cp, cp, cp, sz, cp]
        L.blsgpu_verify_pipeline_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp, sz, vp, vp]
        L.blsgpu_ctx_trim.argtypes = [vp]
        L.blsgpu_pairing_multi.argtypes = [vp, cp, cp, cp, sz, cp]
        L.blsgpu_pairing_multi_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
        L.blsgpu_miller_loop_batch.argtypes = [vp, cp, cp, cp, sz, cp]
        L.blsgpu_miller_loop_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
        L.blsgpu_line_eval_batch.argtypes = [vp, cp, cp, cp, sz, cp]
        L.blsgpu_fq12_op_batch.argtypes = [vp, ctypes.c_int, cp, cp, sz, cp]
        L.blsgpu_fq12_pow_batch.argtypes = [vp, cp, cp, sz, sz, cp]
        L.blsgpu_miller_product_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
        L.blsgpu_final_exp_product_dev.argtypes = [vp, vp, sz, vp, vp]
        L.blsgpu_final_exp.argtypes = [vp, cp, cp]
        for f in (L.blsgpu_g1_msm, L.blsgpu_g2_msm):
            f.argtypes = [vp, cp, cp, sz, sz, cp, cp]
        for f in (L.blsgpu_g1_msm_dev, L.blsgpu_g2_msm_dev):
            f.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp]
        L.blsgpu_final_exp_batch.argtypes = [vp, cp, sz, cp]
        L.blsgpu_pairing_multi_batch.argtypes = [vp, cp, cp, cp, sz, sz, cp]
        L.blsgpu_pairing_multi_batch_dev.argtypes = [vp, vp, vp, vp, sz, sz, vp, vp]
        L.blsgpu_miller_product_batch_dev.argtypes = [vp, vp, vp, vp, sz, sz, vp, vp]
        L.blsgpu_final_exp_product_batch_dev.argtypes = [vp, vp, sz, sz, vp, vp]
        L.blsgpu_g1_decompress.argtypes = [vp, cp, sz, cp, cp]
        L.blsgpu_g2_decompress.argtypes = [vp, cp, sz, cp, cp]
        L.blsgpu_g1_decompress_dev.argtypes = [vp, vp, sz, vp, vp, vp]
        L.blsgpu_g2_decompress_dev.argtypes = [vp, vp, sz, vp, vp, vp]
        L.blsgpu_hash_to_g2.argtypes = [vp, cp, sz, cp]
        L.blsgpu_hash_to_g2_dev.argtypes = [vp, vp, sz, vp, vp]
        L.blsgpu_map_to_g2.argtypes = [vp, cp, sz, cp]
        L.blsgpu_map_to_g2_dev.argtypes = [vp, vp, sz, vp, vp]
        L.blsgpu_timing_enable.argtypes = [vp, ctypes.c_int]
        L.blsgpu_timing_read.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), sz,
                                         ctypes.POINTER(sz)]
        _lib = L
        return L


class Engine:
    """One blsgpu context on one device."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.blsgpu_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise BlsGpuError("blsgpu_ctx_create(%d) failed (%d): %s"
                              % (device, rc, self.lib.blsgpu_last_error().decode()))
        self.h = h
        self.device = device

    def _check(self, rc, what):
        if rc != 0:
            raise BlsGpuError("%s failed (%d): %s" % (what, rc, self.lib.blsgpu_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.blsgpu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def version(self):
        return self.lib.blsgpu_version().decode()

    def set_mp_threshold(self, pairs):
        """Batches >= pairs use the throughput kernel (several pairs per wavefront)."""
        self._check(self.lib.blsgpu_ctx_set_mp_threshold(self.h, pairs), "blsgpu_ctx_set_mp_threshold")

    def mad_probe(self, target_ms=30.0, stream=None):
        """the chip's v_mad_i64_i32 rate right now, in 10^12 multiply-adds per second (a probe kernel of ~target_ms)"""
        out = ctypes.c_double(0.0)
        self._check(self.lib.blsgpu_timing_mad_probe(self.h, float(target_ms), ctypes.byref(out), stream), "blsgpu_timing_mad_probe")
        return out.value

    def mark(self, tag=0, stream=None):
        """one dispatch of an empty kernel: brackets a timed region in a profile of the run"""
        self._check(self.lib.blsgpu_timing_mark(self.h, tag, stream), "blsgpu_timing_mark")

    def set_miller_wide_max(self, pairs):
        """Calls of at most `pairs` pairs run the wide Miller loop (one pair per two-wavefront workgroup); 0: never."""
        self._check(self.lib.blsgpu_ctx_set_miller_wide_max(self.h, pairs), "blsgpu_ctx_set_miller_wide_max")

    def set_mp3_threshold(self, pairs):
        """Throughput kernel: three pairs per wavefront from `pairs` pairs per call on, two below."""
        self._check(self.lib.blsgpu_ctx_set_mp3_threshold(self.h, pairs), "blsgpu_ctx_set_mp3_threshold")

    def set_ls_threshold(self, pairs, min_group=64):
        """Calls >= pairs with groups >= min_group use the line-stream kernels; pairs = None: never."""
        self._check(self.lib.blsgpu_ctx_set_ls_threshold(self.h, (1 << 64) - 1 if pairs is None else pairs, min_group),
                    "blsgpu_ctx_set_ls_threshold")

    WS_FIELDS = ("partials", "staging", "lines", "line_products", "flags_and_lists", "group_sums", "slots", "total")

    def workspace_bytes(self):
        """bytes of HBM the context's grow-only workspace holds, by purpose (include/blsgpu.h BLSGPU_WS_*)"""
        out = (ctypes.c_size_t * len(self.WS_FIELDS))()
        self._check(self.lib.blsgpu_ctx_workspace_bytes(self.h, out), "blsgpu_ctx_workspace_bytes")
        return dict(zip(self.WS_FIELDS, (int(v) for v in out)))

    def verify_pipeline(self, neg_g1, sig, hashes, n, keys_affine=None, key_pts=None, key_scalars=None, k=0):
        """blsgpu_verify_pipeline: e(-G1, sig) * prod e(P_i, H(m_i)) -- hash to G2, key sums and multi-pairing in one
        call on host buffers (ctypes only: no torch)"""
        out = ctypes.create_string_buffer(576)
        self._check(self.lib.blsgpu_verify_pipeline(self.h, neg_g1, sig, hashes if n else None, n, keys_affine, key_pts, key_scalars,
                                                    k, out), "blsgpu_verify_pipeline")
        return out.raw

    def set_ls_teams(self, teams):
        self._check(self.lib.blsgpu_ctx_set_ls_teams(self.h, teams), "blsgpu_ctx_set_ls_teams")

    def set_fexp_team_threshold(self, results):
        """calls with >= results final exponentiations run them six lanes each; None: never"""
        self._check(self.lib.blsgpu_ctx_set_fexp_team_threshold(self.h, (1 << 64) - 1 if results is None else results),
                    "blsgpu_ctx_set_fexp_team_threshold")

    def set_bulk_event(self, event_handle):
        """hipEvent_t handle (int; torch: event.cuda_event after a first record) recorded after the chip-filling
        kernels of every Miller stage, or None"""
        self._check(self.lib.blsgpu_ctx_set_bulk_event(self.h, event_handle), "blsgpu_ctx_set_bulk_event")

    def reserve(self, max_pairs):
        self._check(self.lib.blsgpu_ctx_reserve(self.h, max_pairs), "blsgpu_ctx_reserve")

    def trim(self):
        self._check(self.lib.blsgpu_ctx_trim(self.h), "blsgpu_ctx_trim")

    @staticmethod
    def _inf(inf, n):
        """n x (P flag, Q flag) bytes, or None"""
        if inf is None:
            return None
        inf = bytes(inf)
        if len(inf) != 2 * n:
            raise ValueError("inf must hold 2 flags per pair")
        return inf

    def pairing_multi(self, g1: bytes, g2: bytes, n: int, inf=None) -> bytes:
        if len(g1) != 96 * n or len(g2) != 192 * n:
            raise ValueError("g1/g2 length does not match n")
        out = ctypes.create_string_buffer(576)
        self._check(self.lib.blsgpu_pairing_multi(self.h, g1, g2, self._inf(inf, n), n, out), "blsgpu_pairing_multi")
        return out.raw

    def miller_loop_batch(self, g1: bytes, g2: bytes, n: int, inf=None) -> bytes:
        """n x 576 bytes: the reference's fq_miller_loop value of every pair."""
        if len(g1) != 96 * n or len(g2) != 192 * n:
            raise ValueError("g1/g2 length does not match n")
        out = ctypes.create_string_buffer(max(1, 576 * n))
        self._check(self.lib.blsgpu_miller_loop_batch(self.h, g1, g2, self._inf(inf, n), n, out), "blsgpu_miller_loop_batch")
        return out.raw[:576 * n]

    def line_eval_batch(self, r: bytes, q, p: bytes, n: int) -> bytes:
        """fq2_double_line_eval(R, P) (q None) / fq2_add_line_eval(R, Q, P) for n triples -> n x 576 bytes."""
        if len(r) != 192 * n or len(p) != 96 * n or (q is not None and len(q) != 192 * n):
            raise ValueError("buffer lengths do not match n")
        out = ctypes.create_string_buffer(max(1, 576 * n))
        self._check(self.lib.blsgpu_line_eval_batch(self.h, r, q, p, n, out), "blsgpu_line_eval_batch")
        return out.raw[:576 * n]

    def final_exp(self, x: bytes) -> bytes:
        if len(x) != 576:
            raise ValueError("Fq12 must be 576 bytes")
        out = ctypes.create_string_buffer(576)
        self._check(self.lib.blsgpu_final_exp(self.h, x, out), "blsgpu_final_exp")
        return out.raw

    def final_exp_batch(self, xs: bytes) -> bytes:
        if len(xs) % 576:
            raise ValueError("need m x 576 bytes")
        out = ctypes.create_string_buffer(max(1, len(xs)))
        self._check(self.lib.blsgpu_final_exp_batch(self.h, xs, len(xs) // 576, out), "blsgpu_final_exp_batch")
        return out.raw[:len(xs)]

    FQ12_OPS = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "inv": 4}

    def fq12_op(self, op: str, a: bytes, b: bytes = None) -> bytes:
        """fq12_add / sub / mul / neg / invert on n elements (n x 576 bytes each)."""
        if len(a) % 576 or (b is not None and len(b) != len(a)):
            raise ValueError("need n x 576 bytes")
        out = ctypes.create_string_buffer(max(1, len(a)))
        self._check(self.lib.blsgpu_fq12_op_batch(self.h, self.FQ12_OPS[op], a, b, len(a) // 576, out), "blsgpu_fq12_op_batch")
        return out.raw[:len(a)]

    def fq12_pow(self, a: bytes, e: int) -> bytes:
        if len(a) % 576 or e < 0:
            raise ValueError("need n x 576 bytes and a non-negative exponent")
        eb = e.to_bytes(max(1, (e.bit_length() + 7) // 8), "big")
        out = ctypes.create_string_buffer(max(1, len(a)))
        self._check(self.lib.blsgpu_fq12_pow_batch(self.h, a, eb, len(eb), len(a) // 576, out), "blsgpu_fq12_pow_batch")
        return out.raw[:len(a)]

    def pairing_multi_batch(self, g1: bytes, g2: bytes, gsz: int, groups: int, inf=None) -> bytes:
        n = gsz * groups
        if len(g1) != 96 * n or len(g2) != 192 * n:
            raise ValueError("g1/g2 length does not match gsz * groups")
        out = ctypes.create_string_buffer(max(1, 576 * groups))
        self._check(self.lib.blsgpu_pairing_multi_batch(self.h, g1, g2, self._inf(inf, n), gsz, groups, out), "blsgpu_pairing_multi_batch")
        return out.raw[:576 * groups]

    def _decompress(self, fn, name, insz, data):
        if len(data) % insz:
            raise ValueError("need n x %d bytes" % insz)
        k = len(data) // insz
        out = ctypes.create_string_buffer(max(1, 2 * len(data)))
        ok = ctypes.create_string_buffer(max(1, k))
        self._check(fn(self.h, data, k, out, ok), name)
        return out.raw[:2 * len(data)], [b != 0 for b in ok.raw[:k]]

    def g1_decompress(self, data: bytes):
        """n x 48 bytes -> (n x 96 bytes affine, [accepted])."""
        return self._decompress(self.lib.blsgpu_g1_decompress, "blsgpu_g1_decompress", 48, data)

    def g2_decompress(self, data: bytes):
        """n x 96 bytes -> (n x 192 bytes affine, [accepted])."""
        return self._decompress(self.lib.blsgpu_g2_decompress, "blsgpu_g2_decompress", 96, data)

    def hash_to_g2(self, msg_hashes: bytes) -> bytes:
        """n x 32-byte message hashes -> n x 192 bytes affine G2 (SHA-256 chain on the GPU too)."""
        if len(msg_hashes) % 32:
            raise ValueError("need n x 32 bytes")
        out = ctypes.create_string_buffer(max(1, 6 * len(msg_hashes)))
        self._check(self.lib.blsgpu_hash_to_g2(self.h, msg_hashes, len(msg_hashes) // 32, out), "blsgpu_hash_to_g2")
        return out.raw[:6 * len(msg_hashes)]

    def map_to_g2(self, t: bytes) -> bytes:
        """t: n x 192 bytes (t0.c0, t0.c1, t1.c0, t1.c1) -> n x 192 bytes affine G2."""
        if len(t) % 192:
            raise ValueError("need n x 192 bytes")
        out = ctypes.create_string_buffer(max(1, len(t)))
        self._check(self.lib.blsgpu_map_to_g2(self.h, t, len(t) // 192, out), "blsgpu_map_to_g2")
        return out.raw[:len(t)]

    def _msm(self, fn, psz, pts, scalars, k, groups):
        n = k * groups
        if len(pts) != psz * n:
            raise ValueError("point buffer length does not match k * groups")
        sb = None
        if scalars is not None:
            sb = scalars if isinstance(scalars, (bytes, bytearray)) else b"".join(int(s).to_bytes(32, "big") for s in scalars)
            if len(sb) != 32 * n:
                raise ValueError("scalar buffer length does not match k * groups")
            sb = bytes(sb)
        out = ctypes.create_string_buffer(psz * groups)
        inf = ctypes.create_string_buffer(max(1, groups))
        self._check(fn(self.h, bytes(pts), sb, k, groups, out, inf), fn.__name__)
        return out.raw, [bool(b) for b in inf.raw[:groups]]

    def g1_msm(self, pts, scalars, k, groups=1):
        """-> (groups x 96 affine bytes, [is_infinity])"""
        return self._msm(self.lib.blsgpu_g1_msm, 96, pts, scalars, k, groups)

    def g2_msm(self, pts, scalars, k, groups=1):
        return self._msm(self.lib.blsgpu_g2_msm, 192, pts, scalars, k, groups)

    def timing_enable(self, on=True):
        self._check(self.lib.blsgpu_timing_enable(self.h, int(on)), "blsgpu_timing_enable")

    def timing_read(self):
        """[(kind, ms)] for every kernel launched since the last read; kinds:
        0 k_miller, 1 k_reduce, 2 k_reduce + final exponentiation, 3 k_miller_slow."""
        cap = 1024
        ms = (ctypes.c_float * cap)()
        kind = (ctypes.c_int * cap)()
        cnt = ctypes.c_size_t(0)
        self._check(self.lib.blsgpu_timing_read(self.h, ms, kind, cap, ctypes.byref(cnt)), "blsgpu_timing_read")
        return [(kind[i], ms[i]) for i in range(cnt.value)]

    # device-pointer forms (integers: tensor.data_ptr(), stream.cuda_stream)
    # (d_inf: device pointer to n x 2 flag bytes, or None)
    def pairing_multi_dev(self, d_g1, d_g2, n, d_out, stream=0, d_inf=None):
        self._check(self.lib.blsgpu_pairing_multi_dev(self.h, d_g1, d_g2, d_inf, n, d_out, stream),
                    "blsgpu_pairing_multi_dev")

    def miller_loop_batch_dev(self, d_g1, d_g2, n, d_out, stream=0, d_inf=None):
        self._check(self.lib.blsgpu_miller_loop_batch_dev(self.h, d_g1, d_g2, d_inf, n, d_out, stream),
                    "blsgpu_miller_loop_batch_dev")

    def miller_product_dev(self, d_g1, d_g2, n, d_partial, stream=0, d_inf=None):
        self._check(self.lib.blsgpu_miller_product_dev(self.h, d_g1, d_g2, d_inf, n, d_partial, stream),
                    "blsgpu_miller_product_dev")

    def final_exp_product_dev(self, d_partials, m, d_out, stream=0):
        self._check(self.lib.blsgpu_final_exp_product_dev(self.h, d_partials, m, d_out, stream),
                    "blsgpu_final_exp_product_dev")

    def pairing_multi_batch_dev(self, d_g1, d_g2, gsz, groups, d_out, stream=0, d_inf=None):
        self._check(self.lib.blsgpu_pairing_multi_batch_dev(self.h, d_g1, d_g2, d_inf, gsz, groups, d_out, stream),
                    "blsgpu_pairing_multi_batch_dev")

    def miller_product_batch_dev(self, d_g1, d_g2, gsz, groups, d_partials, stream=0, d_inf=None):
        self._check(self.lib.blsgpu_miller_product_batch_dev(self.h, d_g1, d_g2, d_inf, gsz, groups, d_partials, stream),
                    "blsgpu_miller_product_batch_dev")

    def final_exp_product_batch_dev(self, d_partials, m, groups, d_out, stream=0):
        self._check(self.lib.blsgpu_final_exp_product_batch_dev(self.h, d_partials, m, groups, d_out, stream),
                    "blsgpu_final_exp_product_batch_dev")


_engines = {}


def engine(device=0):
    """Process-wide engine for `device`; raises BlsGpuError if unavailable."""
    with _lock:
        e = _engines.get(device)
    if e is None:
        e = Engine(device)
        with _lock:
            _engines[device] = e
    return e
