"""
ctypes binding of libbxmi.so (include/bxmi.h).

The library is the product: there is NO CPU fallback.  If it is missing, cannot
be loaded, or no MI355X is visible, every call fails loudly with BxmiError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BXMI_LIB") or os.path.join(_HERE, "libbxmi.so")  # (BXMI_LIB: another build of the same library, A/B runs of experiments)

OK, EINVAL, ENOMEM, EHIP, ESTATE, ERANGE = 0, 1, 2, 3, 4, 5


class BxmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbxmi error %d: %s" % (code, msg))
        self.code = code


vp = C.c_void_p
i32, i64 = C.c_int32, C.c_int64
_p = C.POINTER

# name -> argtypes (every function returns int status unless listed in _OTHER_RESTYPE)
_SIGNATURES = {
    "bxmi_device_count": [_p(C.c_int)],
    "bxmi_set_device": [C.c_int],
    "bxmi_get_device": [_p(C.c_int)],
    "bxmi_device_info": [C.c_int, C.c_char_p, C.c_int, _p(C.c_int), _p(i64)],
    "bxmi_mem_info": [_p(i64), _p(i64)],
    "bxmi_synchronize": [vp],
    "bxmi_malloc": [_p(vp), C.c_size_t],
    "bxmi_free": [vp],
    "bxmi_memcpy_h2d": [vp, vp, C.c_size_t],
    "bxmi_memcpy_d2h": [vp, vp, C.c_size_t],
    "bxmi_memset": [vp, C.c_int, C.c_size_t],
    "bxmi_set_option": [C.c_char_p, i64],
    "bxmi_get_option": [C.c_char_p, _p(i64)],
    "bxmi_option_at": [C.c_int, _p(C.c_char_p), _p(i64)],
    "bxmi_ivl_create": [_p(vp)],
    "bxmi_ivl_destroy": [vp],
    "bxmi_ivl_append": [vp, vp, vp, i64],
    "bxmi_ivl_append_dev": [vp, vp, vp, i64, vp],
    "bxmi_ivl_seal": [vp, vp],
    "bxmi_ivl_size": [vp, _p(i64)],
    "bxmi_ivl_has_reversed": [vp, _p(C.c_int)],
    "bxmi_ivl_order": [vp, vp],
    "bxmi_ivl_order_dev": [vp, _p(vp), _p(vp), _p(vp)],
    "bxmi_ivl_count": [vp, vp, vp, i64, vp, _p(i64)],
    "bxmi_ivl_count_dev": [vp, vp, vp, i64, vp, vp, vp],
    "bxmi_ivl_count_multi_dev": [vp, C.c_int, vp, vp, vp, vp, vp, vp],
    "bxmi_ivl_slice_state": [vp, _p(C.c_int), vp],
    "bxmi_ivl_dense_state": [vp, _p(C.c_int), vp],
    "bxmi_ivl_flat_state": [vp, _p(C.c_int), _p(i64)],
    "bxmi_ivl_sparse_state": [vp, _p(C.c_int), _p(i64), _p(C.c_int)],
    "bxmi_ivl_count_width": [vp, _p(C.c_int), _p(i64)],
    "bxmi_ivl_order_state": [vp, _p(C.c_int), _p(i64)],
    "bxmi_ivl_find": [vp, vp, vp, i64, vp, vp, i64, _p(i64)],
    "bxmi_ivl_find_dev": [vp, vp, vp, i64, vp, vp, i64, _p(i64), vp],
    "bxmi_ivl_find_one": [vp, i32, i32, vp, i64, _p(i64)],
    "bxmi_ivl_neighbors": [vp, i32, i32, C.c_int, vp, i64, _p(i64)],
    "bxmi_ivl_clusters": [vp, vp, i32, _p(i64), vp, vp, vp, vp],
    "bxmi_bits_create": [i64, i64, _p(vp)],
    "bxmi_bits_destroy": [vp],
    "bxmi_bits_info": [vp, _p(i32), _p(i32), _p(i32)],
    "bxmi_bits_words_dev": [vp, _p(vp), _p(i64)],
    "bxmi_bits_bin_states": [vp, vp],
    "bxmi_bits_get": [vp, i32, _p(C.c_int)],
    "bxmi_bits_set": [vp, i32],
    "bxmi_bits_clear": [vp, i32],
    "bxmi_bits_set_ranges": [vp, vp, vp, i64],
    "bxmi_bits_set_ranges_dev": [vp, vp, vp, i64, vp],
    "bxmi_bits_count_ranges": [vp, vp, vp, i64, vp],
    "bxmi_bits_count_ranges_dev": [vp, vp, vp, i64, vp, vp],
    "bxmi_bits_count_range": [vp, i32, i32, _p(i32)],
    "bxmi_bits_next": [vp, i32, C.c_int, _p(i32)],
    "bxmi_bits_and": [vp, vp],
    "bxmi_bits_or": [vp, vp],
    "bxmi_bits_not": [vp],
    "bxmi_bits_xor": [vp, vp],
    "bxmi_bits_and_count": [vp, vp, _p(i64)],
    "bxmi_bits_and_dev": [vp, vp, vp],
    "bxmi_bits_or_dev": [vp, vp, vp],
    "bxmi_bits_and_count_dev": [vp, vp, vp, vp],
    "bxmi_bits_popcount_dev": [vp, vp, vp],
    "bxmi_bits_runs": [vp, i32, vp, vp, i64, _p(i64)],
    "bxmi_bed_parse": [vp, i64, C.c_int, C.c_int, C.c_int, _p(vp)],
    "bxmi_bed_destroy": [vp],
    "bxmi_bed_info": [vp, _p(i64), _p(i32), _p(i64), _p(i64), _p(i64)],
    "bxmi_bed_columns": [vp, _p(vp), _p(vp), _p(vp), _p(vp), _p(vp)],
    "bxmi_bed_emit_lines": [vp, vp, vp, C.c_char_p, C.c_int],
    "bxmi_tab_parse": [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, _p(vp)],
    "bxmi_tab_destroy": [vp],
    "bxmi_tab_info": [vp, _p(i64), _p(i32), _p(i64)],
    "bxmi_tab_columns": [vp, _p(vp), _p(vp), _p(vp), _p(vp), _p(vp), _p(vp), _p(vp)],
    "bxmi_comm_unique_id": [vp],
    "bxmi_comm_create": [_p(vp), vp, C.c_int, C.c_int],
    "bxmi_comm_destroy": [vp],
    "bxmi_allreduce_i64": [vp, vp, i64, vp],
    "bxmi_bits_group_create": [_p(vp), C.c_int, _p(vp)],
    "bxmi_bits_group_destroy": [vp],
    "bxmi_bits_group_and_dev": [vp, vp, vp, vp],
    "bxmi_bits_group_or_dev": [vp, vp, vp],
    "bxmi_bits_group_popcount_dev": [vp, vp, vp],
}
_OTHER_RESTYPE = {"bxmi_version": (C.c_int, []), "bxmi_last_error": (C.c_char_p, []),
                  "bxmi_bed_chrom_name": (C.c_char_p, [vp, i32]), "bxmi_tab_chrom_name": (C.c_char_p, [vp, i32])}

EXPORTED = sorted(list(_SIGNATURES) + list(_OTHER_RESTYPE))

_lib = None


def load():
    """dlopen libbxmi.so and attach signatures (no GPU call is made here)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BxmiError(-1, "%s not found: run bx-python_amd/csrc/build.sh (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = args
        for name, (res, args) in _OTHER_RESTYPE.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        # BXMI_OPTS="ivl.bm_variant=1,bits.grid=512": tuning knobs for A/B runs of unmodified scripts (results never depend on them).
        # All of them are parsed and applied before the library is published: a malformed entry fails every load().
        for kv in filter(None, (x.strip() for x in os.environ.get("BXMI_OPTS", "").split(","))):
            key, eq, value = kv.partition("=")
            try:
                number = int(value)
            except ValueError:
                number = None
            if not eq or not key.strip() or number is None:
                raise BxmiError(EINVAL, "BXMI_OPTS: %r is not key=integer" % kv)
            if L.bxmi_set_option(key.strip().encode(), number) != OK:
                raise BxmiError(EINVAL, "BXMI_OPTS: %s" % L.bxmi_last_error().decode(errors="replace"))
        _lib = L
    return _lib


def check(rc, allow=()):
    if rc != OK and rc not in allow:
        raise BxmiError(rc, load().bxmi_last_error().decode(errors="replace"))
    return rc


def call(name, *args, allow=()):
    return check(getattr(load(), name)(*args), allow)


def device_count():
    n = C.c_int(0)
    rc = load().bxmi_device_count(C.byref(n))
    return n.value if rc == OK else 0


def require_gpu():
    if device_count() < 1:
        raise BxmiError(EHIP, "no HIP device visible: libbxmi needs an MI355X (there is no CPU fallback)")


def as_i32(a):
    """C-contiguous int32 view/copy; raises OverflowError like Cython's int coercion would."""
    arr = np.asarray(a)
    if arr.dtype != np.int32:
        if arr.dtype.kind == "f":
            arr = np.trunc(arr)
        if arr.size and (arr.max() > 2147483647 or arr.min() < -2147483648):
            raise OverflowError("value too large to convert to int")
        arr = arr.astype(np.int32)
    return np.ascontiguousarray(arr)


def ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


class DeviceArray:
    """A raw HBM allocation owned by Python (used by the bench / sharded driver)."""

    def __init__(self, nbytes):
        require_gpu()
        p = vp()
        call("bxmi_malloc", C.byref(p), max(int(nbytes), 16))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.nbytes)
        if a.nbytes:
            call("bxmi_memcpy_h2d", d.ptr, ptr(a), a.nbytes)
        return d

    def to_numpy(self, dtype, count=None):
        dt = np.dtype(dtype)
        n = self.nbytes // dt.itemsize if count is None else count
        out = np.empty(n, dtype=dt)
        if n:
            call("bxmi_memcpy_d2h", ptr(out), self.ptr, n * dt.itemsize)
        return out

    def zero(self):
        call("bxmi_memset", self.ptr, 0, self.nbytes)

    def free(self):
        if self.ptr:
            load().bxmi_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def options():
    """{key: current value} of every tuning knob (bxmi_option_at): read once at import it is the library's defaults."""
    L = load()
    out, i = {}, 0
    while True:
        k, v = C.c_char_p(), i64(0)
        if L.bxmi_option_at(i, C.byref(k), C.byref(v)) != 0:
            return out
        out[k.value.decode()] = v.value
        i += 1
