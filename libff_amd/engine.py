"""ctypes binding of libamdmsm.so -- the host-side mirror of libff's multi_exp interface.

The names follow the reference (libff/algebra/scalar_multiplication/multiexp.hpp:21-141):
``multi_exp``, ``multi_exp_filter_one_zero``, ``batch_to_special``,
``bdlo12_signed_optimal_c``, the ``multi_exp_method_*`` / ``multi_exp_base_form_*`` enums.
Everything is computed by the HIP engine; there is NO CPU fallback here -- a missing
``libamdmsm.so`` or a missing GPU raises immediately.

Array conventions (numpy ``uint64``, libff's in-memory layout, see include/amdmsm.h):
  scalars  (n, fr_limbs)      Montgomery residues (as ``std::vector<Fr>`` holds them)
  bases    (n, 3*coord_limbs) (X, Y, Z) records
  result   (3*coord_limbs,)   (X, Y, Z); ``out_form`` selects the representative
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# AMDMSM_LIBRARY: another build of the same ABI (A/B runs of experimental builds, tools/)
SO_PATH = os.environ.get("AMDMSM_LIBRARY") or os.path.join(HERE, "libamdmsm.so")

# curve / group ids (include/amdmsm.h)
ALT_BN128, BLS12_377, BW6_761, BLS12_381 = 0, 1, 2, 3
G1, G2 = 1, 2
CURVE_NAMES = {ALT_BN128: "alt_bn128", BLS12_377: "bls12_377", BW6_761: "bw6_761", BLS12_381: "bls12_381"}

# libff::multi_exp_method, multiexp.hpp:21-43
multi_exp_method_naive = 0
multi_exp_method_naive_plain = 1
multi_exp_method_bos_coster = 2
multi_exp_method_BDLO12 = 3
multi_exp_method_BDLO12_signed = 4
# libff::multi_exp_base_form, multiexp.hpp:45-51
multi_exp_base_form_normal = 0
multi_exp_base_form_special = 1

ABI_VERSION = 3   # AMDMSM_ABI_VERSION of include/amdmsm.h this file mirrors (amdmsm_opts layout)
OUT_JACOBIAN, OUT_LIBFF, OUT_AFFINE = 0, 1, 2
PH_COUNT, PH_SCATTER, PH_ACCUM, PH_REDUCE, PH_FINAL, PH_TOTAL = range(6)
MAX_PHASES = 8

EXPORTED_SYMBOLS = [
    "amdmsm_abi_version", "amdmsm_device_count", "amdmsm_ctx_create", "amdmsm_ctx_destroy", "amdmsm_strerror",
    "amdmsm_last_error", "amdmsm_sizes", "amdmsm_plan", "amdmsm_plan_ex", "amdmsm_endomorphism_info",
    "amdmsm_endomorphism_digits_device", "amdmsm_pippenger_optimal_c",
    "amdmsm_bdlo12_signed_optimal_c", "amdmsm_multi_exp", "amdmsm_multi_exp_batch", "amdmsm_multi_exp_filter_one_zero",
    "amdmsm_multi_exp_multi", "amdmsm_multi_exp_filter_one_zero_multi", "amdmsm_msm_device_multi", "amdmsm_register_bases", "amdmsm_unregister_bases",
    "amdmsm_invalidate_bases",
    "amdmsm_batch_to_special", "amdmsm_batch_exp", "amdmsm_get_batch_exp_timings", "amdmsm_multi_exp_stream", "amdmsm_multi_exp_stream_file",
    "amdmsm_multi_exp_stream_compressed", "amdmsm_multi_exp_stream_compressed_file", "amdmsm_disk_decode_device",
    "amdmsm_precompute_num_digits", "amdmsm_multi_exp_stream_with_precompute",
    "amdmsm_multi_exp_stream_with_precompute_file", "amdmsm_precompute_bases_device",
    "amdmsm_msm_precomputed_device", "amdmsm_import_bases_device", "amdmsm_export_affine_device",
    "amdmsm_msm_device", "amdmsm_msm_device_batch", "amdmsm_sum_points_device", "amdmsm_gen_bases_seq_device",
    "amdmsm_set_timing", "amdmsm_get_timings", "amdmsm_last_timing_ticket", "amdmsm_get_timings_by_ticket",
    "amdmsm_set_pipeline_depth", "amdmsm_last_slot",
    "amdmsm_get_slot_timings", "amdmsm_field_op_device", "amdmsm_group_op_device",
    "amdmsm_digits_device", "amdmsm_mul_bench_device", "amdmsm_madd_bench_device", "amdmsm_malloc", "amdmsm_free",
    "amdmsm_memcpy_h2d", "amdmsm_memcpy_d2h", "amdmsm_synchronize",
]


class AmdMsmError(RuntimeError):
    pass


class _Opts(ctypes.Structure):
    # include/amdmsm.h amdmsm_opts, AMDMSM_ABI_VERSION 3 (struct_size first)
    _fields_ = [("struct_size", ctypes.c_uint32), ("window_bits", ctypes.c_int), ("segment_len", ctypes.c_int),
                ("out_form", ctypes.c_int), ("scalars_plain", ctypes.c_int), ("endomorphism", ctypes.c_int),
                ("stream", ctypes.c_void_p)]


_lib = None


def load_library():
    """Load libamdmsm.so (in-tree).  Raises if it has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise AmdMsmError(
                f"{SO_PATH} is missing: build the HIP engine first (python -m libff_amd.build or "
                "__graft_entry__.build()); libff_amd has no CPU implementation")
        L = ctypes.CDLL(SO_PATH)
        if not hasattr(L, "amdmsm_abi_version") or L.amdmsm_abi_version() != ABI_VERSION:
            raise AmdMsmError(f"{SO_PATH} was built from another include/amdmsm.h (ABI version "
                              f"{L.amdmsm_abi_version() if hasattr(L, 'amdmsm_abi_version') else '< 3'}, this binding "
                              f"expects {ABI_VERSION}): rebuild with python -m libff_amd.build")
        L.amdmsm_strerror.restype = ctypes.c_char_p
        L.amdmsm_last_error.restype = ctypes.c_char_p
        L.amdmsm_last_error.argtypes = [ctypes.c_void_p]
        L.amdmsm_pippenger_optimal_c.restype = ctypes.c_size_t
        L.amdmsm_bdlo12_signed_optimal_c.restype = ctypes.c_size_t
        L.amdmsm_precompute_num_digits.restype = ctypes.c_size_t
        L.amdmsm_last_timing_ticket.restype = ctypes.c_longlong
        L.amdmsm_last_timing_ticket.argtypes = [ctypes.c_void_p]
        L.amdmsm_ctx_destroy.restype = None
        L.amdmsm_ctx_destroy.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def bdlo12_signed_optimal_c(num_entries):
    """multiexp.hpp:53-57 / multiexp.tcc:637-641"""
    return int(load_library().amdmsm_bdlo12_signed_optimal_c(ctypes.c_size_t(num_entries)))


def precompute_num_digits(curve, c):
    """(Fr::num_bits + c - 1) / c: multiples per base in a precompute file (multiexp_stream.tcc:205)."""
    lib = load_library()
    return int(lib.amdmsm_precompute_num_digits(curve, ctypes.c_size_t(c)))


def pippenger_optimal_c(num_elements):
    """multiexp.tcc:35-40"""
    return int(load_library().amdmsm_pippenger_optimal_c(ctypes.c_size_t(num_elements)))


def sizes(curve, group):
    out = (ctypes.c_size_t * 4)()
    rc = load_library().amdmsm_sizes(curve, group, out)
    if rc:
        raise AmdMsmError(f"amdmsm_sizes({curve},{group}): {rc}")
    return {"fr_bytes": out[0], "g_bytes": out[1], "affine_bytes": out[2], "fr_bits": out[3]}


def plan(curve, group, n, window_bits=0, endomorphism=0):
    c, w, used = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    b = ctypes.c_uint32(0)
    ws = ctypes.c_size_t(0)
    rc = load_library().amdmsm_plan_ex(curve, group, ctypes.c_size_t(n), window_bits, endomorphism, ctypes.byref(c),
                                       ctypes.byref(w), ctypes.byref(b), ctypes.byref(ws), ctypes.byref(used))
    if rc:
        raise AmdMsmError(f"amdmsm_plan_ex: {rc}")
    return {"c": c.value, "num_windows": w.value, "num_buckets": b.value, "workspace_bytes": ws.value,
            "endomorphism": bool(used.value)}


def endomorphism_info(curve, group):
    """lambda (int), log2 bound of the half scalars, whether the whole curve group has order r"""
    s = sizes(curve, group)
    lam = (ctypes.c_uint8 * s["fr_bytes"])()
    bound, prime = ctypes.c_int(0), ctypes.c_int(0)
    rc = load_library().amdmsm_endomorphism_info(curve, group, lam, ctypes.byref(bound), ctypes.byref(prime))
    if rc:
        raise AmdMsmError(f"amdmsm_endomorphism_info: {rc}")
    return {"lambda": int.from_bytes(bytes(lam), "little"), "bound_log2": bound.value / 1000.0,
            "prime_order": bool(prime.value)}


def multi_exp_multi(engines, curve, group, bases, scalars, base_form=multi_exp_base_form_normal,
                    out_form=OUT_AFFINE, window_bits=0, scalars_plain=False):
    """amdmsm_multi_exp_multi: libff's range split (multiexp.tcc:655-687) with chunk = engine context
    (one per GPU, or several on one GPU), partials combined on engines[0]'s device."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    s = sizes(curve, group)
    n = bases.shape[0] if bases.ndim == 2 else 0
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    e0 = engines[0]
    ctxs = (ctypes.c_void_p * len(engines))(*[e.h for e in engines])
    o = e0._opts(window_bits=window_bits, out_form=out_form, scalars_plain=scalars_plain)
    rc = e0.lib.amdmsm_multi_exp_multi(ctxs, len(engines), curve, group, _np_ptr(bases) if n else None,
                                       ctypes.c_size_t(s["g_bytes"]), base_form, _np_ptr(scalars) if n else None,
                                       ctypes.c_size_t(n), _np_ptr(out), ctypes.byref(o))
    e0._check(rc, "amdmsm_multi_exp_multi")
    return out


def multi_exp_filter_one_zero_multi(engines, curve, group, bases, scalars, base_form=multi_exp_base_form_normal,
                                    out_form=OUT_AFFINE, window_bits=0, scalars_plain=False):
    """amdmsm_multi_exp_filter_one_zero_multi: (result, {"skipped", "ones", "other"}) with the range split of
    multi_exp_multi; every context counts the zeros / ones of its own range."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    s = sizes(curve, group)
    n = bases.shape[0] if bases.ndim == 2 else 0
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    st = (ctypes.c_size_t * 3)()
    e0 = engines[0]
    ctxs = (ctypes.c_void_p * len(engines))(*[e.h for e in engines])
    o = e0._opts(window_bits=window_bits, out_form=out_form, scalars_plain=scalars_plain)
    rc = e0.lib.amdmsm_multi_exp_filter_one_zero_multi(ctxs, len(engines), curve, group, _np_ptr(bases) if n else None,
                                                       ctypes.c_size_t(s["g_bytes"]), base_form,
                                                       _np_ptr(scalars) if n else None, ctypes.c_size_t(n), _np_ptr(out),
                                                       ctypes.byref(o), st)
    e0._check(rc, "amdmsm_multi_exp_filter_one_zero_multi")
    return out, {"skipped": st[0], "ones": st[1], "other": st[2]}


def msm_device_multi(engines, curve, group, d_bases, d_scalars, counts, d_out_dev0, out_form=OUT_LIBFF, window_bits=0,
                     scalars_plain=False, stream=None):
    """amdmsm_msm_device_multi: range k (compact affine bases, scalars; device pointers on
    engines[k]'s GPU) is reduced there, the partials are summed on engines[0]'s GPU into d_out_dev0."""
    k = len(engines)
    e0 = engines[0]
    ctxs = (ctypes.c_void_p * k)(*[e.h for e in engines])
    pb = (ctypes.c_void_p * k)(*[_vp(x) for x in d_bases])
    ps = (ctypes.c_void_p * k)(*[_vp(x) for x in d_scalars])
    cn = (ctypes.c_size_t * k)(*counts)
    o = e0._opts(window_bits=window_bits, out_form=out_form, scalars_plain=scalars_plain, stream=stream)
    rc = e0.lib.amdmsm_msm_device_multi(ctxs, k, curve, group, pb, ps, cn, _vp(d_out_dev0), ctypes.byref(o))
    e0._check(rc, "amdmsm_msm_device_multi")


def _vp(x):
    """device address (int, e.g. torch's data_ptr()) or c_void_p (Engine.malloc) -> c_void_p"""
    return x if isinstance(x, ctypes.c_void_p) else ctypes.c_void_p(x)


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """One amdmsm context (device, stream, workspace).  Fails loudly without a GPU."""

    def __init__(self, device=0, endomorphism=0):
        self.lib = load_library()
        self.device = device
        self.endomorphism = endomorphism
        h = ctypes.c_void_p()
        rc = self.lib.amdmsm_ctx_create(device, ctypes.byref(h))
        if rc:
            raise AmdMsmError("amdmsm_ctx_create(device=%d) failed: %s" %
                              (device, self.lib.amdmsm_strerror(rc).decode()))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.amdmsm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise AmdMsmError("%s failed: %s (%s)" % (what, self.lib.amdmsm_strerror(rc).decode(),
                                                      self.lib.amdmsm_last_error(self.h).decode()))

    def _opts(self, window_bits=0, segment_len=0, out_form=OUT_LIBFF, scalars_plain=False, stream=None):
        # self.endomorphism: amdmsm_opts.endomorphism for every call of this engine (0 = only where the
        # whole curve group has order r, 1 = the bases are promised to lie in the order-r subgroup, -1 = off)
        return _Opts(ctypes.sizeof(_Opts), window_bits, segment_len, out_form, int(scalars_plain), int(self.endomorphism),
                     stream)

    # ---------------------------------------------------------------- host API
    def multi_exp(self, curve, group, bases, scalars, method=multi_exp_method_BDLO12_signed,
                  base_form=multi_exp_base_form_normal, chunks=1, out_form=OUT_AFFINE, window_bits=0,
                  scalars_plain=False, split_chunks=False):
        """libff::multi_exp<G, Fr, Method, BaseForm>(bases, scalars, chunks), multiexp.tcc:643-688.

        BDLO12 and BDLO12_signed both run the device Pippenger engine (the result is a
        group element; it does not depend on the digit convention).  ``chunks`` is the
        reference's CPU-thread hint (libsnark passes its OpenMP thread count): one GPU runs the
        whole input as ONE MSM whatever its value -- splitting would only multiply the fixed
        costs -- and across GPUs the split is ``multi_exp_multi``'s.  ``split_chunks=True``
        forces the reference's split-and-sum shape (contiguous ranges, the last one takes the
        remainder, partials summed; multiexp.tcc:655-687) on this one device, for the parity tests
        of range sharding.
        """
        if method not in (multi_exp_method_BDLO12, multi_exp_method_BDLO12_signed):
            raise NotImplementedError("only the BDLO12 / BDLO12_signed methods run on the GPU engine")
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        s = sizes(curve, group)
        n = bases.shape[0] if bases.ndim == 2 else 0
        if n:
            assert bases.shape[1] * 8 == s["g_bytes"], "bases must be (n, 3*coord_limbs) uint64"
            assert scalars.shape == (n, s["fr_bytes"] // 8), "scalars must be (n, fr_limbs) uint64"
        out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
        total = n
        if total < chunks or chunks == 1 or not split_chunks:
            o = self._opts(window_bits=window_bits, out_form=out_form, scalars_plain=scalars_plain)
            rc = self.lib.amdmsm_multi_exp(self.h, curve, group, _np_ptr(bases) if n else None,
                                           ctypes.c_size_t(s["g_bytes"]), base_form,
                                           _np_ptr(scalars) if n else None, ctypes.c_size_t(n), _np_ptr(out),
                                           ctypes.byref(o))
            self._check(rc, "amdmsm_multi_exp")
            return out
        one = total // chunks
        partials = np.zeros((chunks, s["g_bytes"] // 8), dtype=np.uint64)
        o = self._opts(window_bits=window_bits, out_form=OUT_JACOBIAN, scalars_plain=scalars_plain)
        for i in range(chunks):
            lo = i * one
            hi = total if i == chunks - 1 else (i + 1) * one
            b, sc = bases[lo:hi], scalars[lo:hi]
            rc = self.lib.amdmsm_multi_exp(self.h, curve, group, _np_ptr(b), ctypes.c_size_t(s["g_bytes"]),
                                           base_form, _np_ptr(sc), ctypes.c_size_t(hi - lo),
                                           _np_ptr(partials[i]), ctypes.byref(o))
            self._check(rc, "amdmsm_multi_exp")
        return self.sum_points(curve, group, partials, out_form=out_form)

    def multi_exp_batch(self, curve, group, bases_list, scalars_list, base_form=multi_exp_base_form_normal,
                        out_form=OUT_AFFINE, window_bits=0, scalars_plain=False):
        """amdmsm_multi_exp_batch: len(bases_list) multi_exp calls of one group and length as one batch; list of results."""
        k = len(bases_list)
        bs = [np.ascontiguousarray(b, dtype=np.uint64) for b in bases_list]
        ss = [np.ascontiguousarray(x, dtype=np.uint64) for x in scalars_list]
        s = sizes(curve, group)
        n = bs[0].shape[0]
        assert all(b.shape == bs[0].shape for b in bs) and all(x.shape[0] == n for x in ss)
        outs = [np.zeros(s["g_bytes"] // 8, dtype=np.uint64) for _ in range(k)]
        pb = (ctypes.c_void_p * k)(*[_np_ptr(b) for b in bs])
        ps = (ctypes.c_void_p * k)(*[_np_ptr(x) for x in ss])
        po = (ctypes.c_void_p * k)(*[_np_ptr(o) for o in outs])
        o = self._opts(window_bits=window_bits, out_form=out_form, scalars_plain=scalars_plain)
        rc = self.lib.amdmsm_multi_exp_batch(self.h, curve, group, k, pb, ctypes.c_size_t(s["g_bytes"]), base_form, ps,
                                             ctypes.c_size_t(n), po, ctypes.byref(o))
        self._check(rc, "amdmsm_multi_exp_batch")
        return outs

    def register_bases(self, curve, group, bases, base_form=multi_exp_base_form_normal):
        """amdmsm_register_bases: keep ``bases`` (the very numpy buffer -- the registry is keyed on its
        address) resident in HBM; later host-buffer calls on it or on row ranges of it skip the base
        transfer and import.  Returns a handle for ``unregister_bases``."""
        assert bases.dtype == np.uint64 and bases.flags["C_CONTIGUOUS"]
        s = sizes(curve, group)
        h = ctypes.c_uint64(0)
        rc = self.lib.amdmsm_register_bases(self.h, curve, group, _np_ptr(bases), ctypes.c_size_t(s["g_bytes"]),
                                            base_form, ctypes.c_size_t(bases.shape[0]), ctypes.byref(h))
        self._check(rc, "amdmsm_register_bases")
        return h.value

    def unregister_bases(self, handle):
        self._check(self.lib.amdmsm_unregister_bases(self.h, ctypes.c_uint64(handle)), "amdmsm_unregister_bases")

    def invalidate_bases(self, arr=None):
        rc = self.lib.amdmsm_invalidate_bases(self.h, _np_ptr(arr) if arr is not None else None,
                                              ctypes.c_size_t(arr.nbytes if arr is not None else 0))
        self._check(rc, "amdmsm_invalidate_bases")

    def multi_exp_filter_one_zero(self, curve, group, bases, scalars, method=multi_exp_method_BDLO12_signed,
                                  base_form=multi_exp_base_form_normal, chunks=1, out_form=OUT_AFFINE,
                                  scalars_plain=False):
        """libff::multi_exp_filter_one_zero, multiexp.tcc:690-757.  Returns (result, stats)."""
        if method not in (multi_exp_method_BDLO12, multi_exp_method_BDLO12_signed):
            raise NotImplementedError("only the BDLO12 / BDLO12_signed methods run on the GPU engine")
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        s = sizes(curve, group)
        n = bases.shape[0]
        out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
        stats = (ctypes.c_size_t * 3)()
        o = self._opts(out_form=out_form, scalars_plain=scalars_plain)
        rc = self.lib.amdmsm_multi_exp_filter_one_zero(
            self.h, curve, group, _np_ptr(bases), ctypes.c_size_t(s["g_bytes"]), base_form, _np_ptr(scalars),
            ctypes.c_size_t(n), _np_ptr(out), ctypes.byref(o), stats)
        self._check(rc, "amdmsm_multi_exp_filter_one_zero")
        return out, {"skipped": stats[0], "ones": stats[1], "other": stats[2]}

    def batch_to_special(self, curve, group, elems):
        """libff::batch_to_special<G>, multiexp.tcc:949-974 (returns a converted copy)."""
        elems = np.ascontiguousarray(elems, dtype=np.uint64).copy()
        s = sizes(curve, group)
        rc = self.lib.amdmsm_batch_to_special(self.h, curve, group, _np_ptr(elems), ctypes.c_size_t(s["g_bytes"]),
                                              ctypes.c_size_t(elems.shape[0]))
        self._check(rc, "amdmsm_batch_to_special")
        return elems

    def multi_exp_stream_file(self, curve, group, path, scalars, offset_bytes=0, chunk_points=0,
                              out_form=OUT_AFFINE, scalars_plain=False):
        """libff::multi_exp_stream<form_montgomery, compression_off> (multiexp_stream.tcc:164-191) with
        the base elements read from ``path`` in libff's on-disk format."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        s = sizes(curve, group)
        out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
        o = self._opts(out_form=out_form, scalars_plain=scalars_plain)
        rc = self.lib.amdmsm_multi_exp_stream_file(self.h, curve, group, path.encode(), ctypes.c_size_t(offset_bytes),
                                                   _np_ptr(scalars) if scalars.shape[0] else None,
                                                   ctypes.c_size_t(scalars.shape[0]), ctypes.c_size_t(chunk_points),
                                                   _np_ptr(out), ctypes.byref(o))
        self._check(rc, "amdmsm_multi_exp_stream_file")
        return out

    def multi_exp_stream_compressed_file(self, curve, group, path, scalars, offset_bytes=0, chunk_points=0,
                                         out_form=OUT_AFFINE, scalars_plain=False):
        """libff::multi_exp_stream<form_montgomery, compression_on>: ``path`` holds compressed records
        (curve_serialization.tcc:103-133); Y is recovered on the device."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        s = sizes(curve, group)
        out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
        o = self._opts(out_form=out_form, scalars_plain=scalars_plain)
        rc = self.lib.amdmsm_multi_exp_stream_compressed_file(
            self.h, curve, group, path.encode(), ctypes.c_size_t(offset_bytes),
            _np_ptr(scalars) if scalars.shape[0] else None, ctypes.c_size_t(scalars.shape[0]),
            ctypes.c_size_t(chunk_points), _np_ptr(out), ctypes.byref(o))
        self._check(rc, "amdmsm_multi_exp_stream_compressed_file")
        return out

    def disk_decode(self, curve, group, records, n, compressed):
        """group_read<encoding_binary, form_montgomery, compression_{off,on}> of ``n`` records (uint8 array)
        on the device -> (special-form (X, Y, Z) records, status)."""
        records = np.ascontiguousarray(records, dtype=np.uint8)
        s = sizes(curve, group)
        d_rec, d_aff, d_xyz = self.malloc(max(16, records.nbytes)), self.malloc(max(16, n * s["affine_bytes"])), \
            self.malloc(max(16, n * s["g_bytes"]))
        try:
            self.h2d(d_rec, records)
            st = ctypes.c_uint(0)
            self._check(self.lib.amdmsm_disk_decode_device(self.h, curve, group, d_rec, ctypes.c_size_t(n), int(compressed),
                                                           d_aff, ctypes.byref(st)), "amdmsm_disk_decode_device")
            self.export_affine_device(curve, group, d_aff.value, n, d_xyz.value)
            out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
            self.synchronize()
            self.d2h(out, d_xyz)
            return out, st.value
        finally:
            for p in (d_rec, d_aff, d_xyz):
                self.free(p)

    def multi_exp_stream_with_precompute_file(self, curve, group, path, scalars, precompute_c, offset_bytes=0,
                                              chunk_points=0, out_form=OUT_AFFINE, scalars_plain=False):
        """libff::multi_exp_stream_with_precompute<form_montgomery, compression_off>
        (multiexp_stream.tcc:193-223): ``path`` holds precompute_num_digits(curve, c) multiples
        [2^(jc)]P per base, as profile_multiexp.cpp:120-150 writes them."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        s = sizes(curve, group)
        out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
        o = self._opts(out_form=out_form, scalars_plain=scalars_plain)
        rc = self.lib.amdmsm_multi_exp_stream_with_precompute_file(
            self.h, curve, group, path.encode(), ctypes.c_size_t(offset_bytes),
            _np_ptr(scalars) if scalars.shape[0] else None, ctypes.c_size_t(scalars.shape[0]),
            ctypes.c_size_t(precompute_c), ctypes.c_size_t(chunk_points), _np_ptr(out), ctypes.byref(o))
        self._check(rc, "amdmsm_multi_exp_stream_with_precompute_file")
        return out

    def precompute_table(self, curve, group, bases, c, num_digits=None):
        """Host convenience around amdmsm_precompute_bases_device: special-form (x, y, 1) records in,
        table[i * D + j] = [2^(jc)] bases[i] out, again as special-form records."""
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        n = bases.shape[0]
        D = num_digits or precompute_num_digits(curve, c)
        s = sizes(curve, group)
        d_src = self.malloc(max(bases.nbytes, 16))
        d_aff = self.malloc(max(n * s["affine_bytes"], 16))
        d_tab = self.malloc(max(n * D * s["affine_bytes"], 16))
        d_out = self.malloc(max(n * D * s["g_bytes"], 16))
        try:
            self.h2d(d_src, bases)
            self.import_bases_device(curve, group, d_src, bases.strides[0], multi_exp_base_form_special, n, d_aff)
            self._check(self.lib.amdmsm_precompute_bases_device(
                self.h, curve, group, _vp(d_aff), ctypes.c_size_t(n), ctypes.c_size_t(c),
                ctypes.c_size_t(D), _vp(d_tab), None), "amdmsm_precompute_bases_device")
            self.export_affine_device(curve, group, d_tab, n * D, d_out)
            out = np.zeros((n * D, s["g_bytes"] // 8), dtype=np.uint64)
            self.synchronize()
            self.d2h(out, d_out)
        finally:
            for p in (d_src, d_aff, d_tab, d_out):
                self.free(p)
        return out

    def batch_exp(self, curve, group, scalar_size, window, g, v, coeff=None, scalars_plain=False, out=None):
        """libff::batch_exp / batch_exp_with_coeff (multiexp.tcc:874-947) for the table that
        get_window_table(scalar_size, window, g) would build: res[i] = (coeff *) v[i] * g.
        out: result array to fill (n, 3 * coordinate limbs) -- a caller that repeats the call passes the same
        array again and spares the page faults of a fresh 100 MB allocation."""
        g = np.ascontiguousarray(g, dtype=np.uint64)
        v = np.ascontiguousarray(v, dtype=np.uint64)
        s = sizes(curve, group)
        n = v.shape[0]
        if out is None:
            out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.flags.c_contiguous and out.shape == (n, s["g_bytes"] // 8)
        cf = np.ascontiguousarray(coeff, dtype=np.uint64) if coeff is not None else None
        rc = self.lib.amdmsm_batch_exp(self.h, curve, group, ctypes.c_size_t(scalar_size), ctypes.c_size_t(window),
                                       _np_ptr(g), _np_ptr(v) if n else None, ctypes.c_size_t(n),
                                       _np_ptr(cf) if cf is not None else None, int(scalars_plain),
                                       _np_ptr(out) if n else None)
        self._check(rc, "amdmsm_batch_exp")
        return out

    def batch_exp_timings(self):
        """device times (ms) of the last batch_exp: inputs H2D, window table (0 = reused), exponentiations, results D2H"""
        ms = (ctypes.c_float * 4)()
        self._check(self.lib.amdmsm_get_batch_exp_timings(self.h, ms), "amdmsm_get_batch_exp_timings")
        return {"h2d_ms": ms[0], "table_ms": ms[1], "exp_ms": ms[2], "d2h_ms": ms[3]}

    def sum_points(self, curve, group, points_jacobian, out_form=OUT_AFFINE):
        """Sum of engine-Jacobian partial results (the serial tail of multiexp.tcc:681-687)."""
        pts = np.ascontiguousarray(points_jacobian, dtype=np.uint64)
        s = sizes(curve, group)
        k = pts.shape[0]
        d_pts, d_out = self.malloc(max(1, pts.nbytes)), self.malloc(s["g_bytes"])
        try:
            if k:
                self.h2d(d_pts, pts)
            rc = self.lib.amdmsm_sum_points_device(self.h, curve, group, d_pts, k, out_form, d_out, None)
            self._check(rc, "amdmsm_sum_points_device")
            out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
            self.d2h(out, d_out)
            return out
        finally:
            self.free(d_pts)
            self.free(d_out)

    # -------------------------------------------------------- raw device API
    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self.lib.amdmsm_malloc(self.h, ctypes.c_size_t(nbytes), ctypes.byref(p)), "amdmsm_malloc")
        return p

    def free(self, p):
        self._check(self.lib.amdmsm_free(self.h, p), "amdmsm_free")

    def h2d(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.amdmsm_memcpy_h2d(self.h, d_ptr, _np_ptr(arr), ctypes.c_size_t(arr.nbytes)), "h2d")

    def d2h(self, arr, d_ptr):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(self.lib.amdmsm_memcpy_d2h(self.h, _np_ptr(arr), d_ptr, ctypes.c_size_t(arr.nbytes)), "d2h")

    def synchronize(self):
        self._check(self.lib.amdmsm_synchronize(self.h), "amdmsm_synchronize")

    def import_bases_device(self, curve, group, d_src_xyz, stride_bytes, base_form, n, d_dst_affine, stream=None):
        self._check(self.lib.amdmsm_import_bases_device(self.h, curve, group, _vp(d_src_xyz),
                                                        ctypes.c_size_t(stride_bytes), base_form,
                                                        ctypes.c_size_t(n), _vp(d_dst_affine),
                                                        _vp(stream)), "amdmsm_import_bases_device")

    def export_affine_device(self, curve, group, d_src_affine, n, d_dst_xyz, stream=None):
        self._check(self.lib.amdmsm_export_affine_device(self.h, curve, group, _vp(d_src_affine),
                                                         ctypes.c_size_t(n), _vp(d_dst_xyz),
                                                         _vp(stream)), "amdmsm_export_affine_device")

    def msm_device(self, curve, group, d_bases_affine, d_scalars, n, d_out_xyz, out_form=OUT_LIBFF,
                   window_bits=0, segment_len=0, scalars_plain=False, stream=None):
        o = self._opts(window_bits, segment_len, out_form, scalars_plain, stream)
        self._check(self.lib.amdmsm_msm_device(self.h, curve, group, _vp(d_bases_affine),
                                               _vp(d_scalars), ctypes.c_size_t(n),
                                               _vp(d_out_xyz), ctypes.byref(o)), "amdmsm_msm_device")

    def msm_device_batch(self, curve, group, d_bases_affine, d_scalars, n, d_out_xyz, out_form=OUT_LIBFF,
                         window_bits=0, scalars_plain=False, stream=None):
        """amdmsm_msm_device_batch: len(d_bases_affine) MSMs of n points each (lists of device pointers) in one call;
        the tails of all of them run as one set of kernels."""
        k = len(d_bases_affine)
        pb = (ctypes.c_void_p * k)(*[_vp(x) for x in d_bases_affine])
        ps = (ctypes.c_void_p * k)(*[_vp(x) for x in d_scalars])
        po = (ctypes.c_void_p * k)(*[_vp(x) for x in d_out_xyz])
        o = self._opts(window_bits, 0, out_form, scalars_plain, stream)
        self._check(self.lib.amdmsm_msm_device_batch(self.h, curve, group, k, pb, ps, ctypes.c_size_t(n), po, ctypes.byref(o)),
                    "amdmsm_msm_device_batch")

    def precompute_bases_device(self, curve, group, d_bases_affine, n, c, num_digits, d_table, stream=None):
        self._check(self.lib.amdmsm_precompute_bases_device(
            self.h, curve, group, _vp(d_bases_affine), ctypes.c_size_t(n), ctypes.c_size_t(c),
            ctypes.c_size_t(num_digits), _vp(d_table), _vp(stream)),
            "amdmsm_precompute_bases_device")

    def msm_precomputed_device(self, curve, group, d_table, d_scalars, n, c, num_digits, d_out_xyz,
                               out_form=OUT_LIBFF, segment_len=0, scalars_plain=False, stream=None):
        o = self._opts(0, segment_len, out_form, scalars_plain, stream)
        self._check(self.lib.amdmsm_msm_precomputed_device(
            self.h, curve, group, _vp(d_table), _vp(d_scalars), ctypes.c_size_t(n),
            ctypes.c_size_t(c), ctypes.c_size_t(num_digits), _vp(d_out_xyz), ctypes.byref(o)),
            "amdmsm_msm_precomputed_device")

    def sum_points_device(self, curve, group, d_points, k, out_form, d_out, stream=None):
        self._check(self.lib.amdmsm_sum_points_device(self.h, curve, group, _vp(d_points), k, out_form,
                                                      _vp(d_out), _vp(stream)),
                    "amdmsm_sum_points_device")

    def gen_bases_seq_device(self, curve, group, first, n, d_dst_affine, stream=None):
        self._check(self.lib.amdmsm_gen_bases_seq_device(self.h, curve, group, ctypes.c_uint64(first),
                                                         ctypes.c_size_t(n), _vp(d_dst_affine),
                                                         _vp(stream)), "amdmsm_gen_bases_seq_device")

    def set_timing(self, enable=True):
        self._check(self.lib.amdmsm_set_timing(self.h, int(enable)), "amdmsm_set_timing")

    def set_pipeline_depth(self, depth):
        """Number of MSMs that may be in flight (workspace slots, 1..4); see include/amdmsm.h."""
        self._check(self.lib.amdmsm_set_pipeline_depth(self.h, int(depth)), "amdmsm_set_pipeline_depth")

    def last_slot(self):
        return int(self.lib.amdmsm_last_slot(self.h))

    def last_timing_ticket(self):
        """ticket of the most recent timed MSM (see amdmsm_get_timings_by_ticket); -1 if none"""
        return int(self.lib.amdmsm_last_timing_ticket(self.h))

    def get_timings(self, slot=None, ticket=None):
        ms = (ctypes.c_float * MAX_PHASES)()
        if ticket is not None:
            self._check(self.lib.amdmsm_get_timings_by_ticket(self.h, ctypes.c_longlong(ticket), ms),
                        "amdmsm_get_timings_by_ticket")
        elif slot is None:
            self._check(self.lib.amdmsm_get_timings(self.h, ms), "amdmsm_get_timings")
        else:
            self._check(self.lib.amdmsm_get_slot_timings(self.h, int(slot), ms), "amdmsm_get_slot_timings")
        return {"count_ms": ms[PH_COUNT], "scatter_ms": ms[PH_SCATTER], "accumulate_ms": ms[PH_ACCUM],
                "reduce_ms": ms[PH_REDUCE], "final_ms": ms[PH_FINAL], "total_ms": ms[PH_TOTAL]}

    # ------------------------------------------------------------ test hooks
    def _dev_arrays(self, *arrs):
        ptrs = []
        for a in arrs:
            if a is None:
                ptrs.append(None)
                continue
            p = self.malloc(max(16, a.nbytes))
            self.h2d(p, a)
            ptrs.append(p)
        return ptrs

    def field_op(self, curve, group, op, a, b=None):
        """coordinate-field op over arrays: 0 mul 1 sqr 2 add 3 sub 4 neg 5 inverse"""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
        out = np.zeros_like(a)
        pa, pb, po = self._dev_arrays(a, b, out)
        try:
            self._check(self.lib.amdmsm_field_op_device(self.h, curve, group, op, pa, pb, po,
                                                        ctypes.c_size_t(a.shape[0])), "amdmsm_field_op_device")
            self.d2h(out, po)
        finally:
            for p in (pa, pb, po):
                if p is not None:
                    self.free(p)
        return out

    def group_op(self, curve, group, op, a, b=None, out_form=OUT_LIBFF):
        """group op over arrays of libff records: 0 add 1 mixed_add 2 dbl"""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
        out = np.zeros_like(a)
        pa, pb, po = self._dev_arrays(a, b, out)
        try:
            self._check(self.lib.amdmsm_group_op_device(self.h, curve, group, op, pa, pb, po,
                                                        ctypes.c_size_t(a.shape[0]), out_form),
                        "amdmsm_group_op_device")
            self.d2h(out, po)
        finally:
            for p in (pa, pb, po):
                if p is not None:
                    self.free(p)
        return out

    def endomorphism_digits(self, curve, group, scalars, c, num_windows, scalars_plain=False):
        """device recoding of both halves of every scalar: int32 (n, 2, num_windows)"""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        n = scalars.shape[0]
        out = np.zeros((n, 2, num_windows), dtype=np.int32)
        ps, po = self._dev_arrays(scalars, out)
        try:
            self._check(self.lib.amdmsm_endomorphism_digits_device(self.h, curve, group, ps, ctypes.c_size_t(n),
                                                                   int(scalars_plain), c, num_windows, po),
                        "amdmsm_endomorphism_digits_device")
            self.d2h(out, po)
        finally:
            self.free(ps)
            self.free(po)
        return out

    def signed_digits(self, curve, scalars, c, num_windows, scalars_plain=False):
        """device recoding of every scalar: int32 (n, num_windows), least-significant window first"""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        n = scalars.shape[0]
        out = np.zeros((n, num_windows), dtype=np.int32)
        ps, po = self._dev_arrays(scalars, out)
        try:
            self._check(self.lib.amdmsm_digits_device(self.h, curve, G1, ps, ctypes.c_size_t(n), int(scalars_plain),
                                                      c, num_windows, po), "amdmsm_digits_device")
            self.d2h(out, po)
        finally:
            self.free(ps)
            self.free(po)
        return out

    def gen_bases_seq(self, curve, group, n, first=0, as_xyz=True):
        """(first+i+1)*G for i < n, computed on the device; libff special-form records by default"""
        s = sizes(curve, group)
        d_aff = self.malloc(max(16, n * s["affine_bytes"]))
        try:
            self.gen_bases_seq_device(curve, group, first, n, d_aff.value)
            if not as_xyz:
                out = np.zeros((n, s["affine_bytes"] // 8), dtype=np.uint64)
                self.d2h(out, d_aff)
                return out
            d_xyz = self.malloc(max(16, n * s["g_bytes"]))
            try:
                self.export_affine_device(curve, group, d_aff.value, n, d_xyz.value)
                out = np.zeros((n, s["g_bytes"] // 8), dtype=np.uint64)
                self.d2h(out, d_xyz)
                return out
            finally:
                self.free(d_xyz)
        finally:
            self.free(d_aff)
