"""ctypes binding of the C-ABI library librmt_n2.so (include/rmt_n2.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C rmt_app_amd/csrc``).
There is deliberately no fallback: if the shared object is missing or no HIP device is present
the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RMT_N2_LIBRARY selects another build of the same C ABI (the host-ASan build of `make asan`)
LIB_PATH = os.environ.get("RMT_N2_LIBRARY") or os.path.join(_HERE, "librmt_n2.so")
CACHE_DIR = os.path.join(_HERE, "_kcache")

ABI_VERSION = 2


class RmtN2Error(RuntimeError):
    pass


class Plan(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_species", C.c_int32), ("n_reactions", C.c_int32),
        ("n_vars", C.c_int32), ("n_nodes", C.c_int32), ("n_members", C.c_int32),
        ("fp32", C.c_int32), ("block", C.c_int32), ("nodes_per_thread", C.c_int32),
        ("n_user_params", C.c_int32), ("ros4_nodes_per_block", C.c_int32), ("reserved", C.c_int32),
        ("code_object", C.c_void_p), ("code_size", C.c_size_t), ("members", C.POINTER(C.c_double)),
    ]


class Stats(C.Structure):
    _fields_ = [("t_end", C.c_double), ("h_last", C.c_double), ("accepted", C.c_int64),
                ("rejected", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RmtN2Error("%s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(or make -C rmt_app_amd/csrc); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, cp, dbl, i64 = C.c_void_p, C.c_char_p, C.c_double, C.c_int64
    L.rmt_n2_last_error.restype = cp
    L.rmt_n2_abi_version.restype = C.c_int
    L.rmt_n2_kernel_template.restype = cp
    L.rmt_n2_compile.argtypes = [cp, cp, cp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
    L.rmt_n2_free.argtypes = [vp]
    L.rmt_n2_free.restype = None
    L.rmt_n2_create.argtypes = [C.POINTER(Plan), C.POINTER(vp)]
    L.rmt_n2_destroy.argtypes = [vp]
    L.rmt_n2_destroy.restype = None
    L.rmt_n2_set_stream.argtypes = [vp, vp]
    L.rmt_n2_set_mode.argtypes = [vp, C.c_int]
    L.rmt_n2_set_members.argtypes = [vp, C.POINTER(dbl)]
    L.rmt_n2_rhs.argtypes = [vp, dbl, vp, vp]
    L.rmt_n2_rk4.argtypes = [vp, vp, dbl, dbl, i64]
    L.rmt_n2_multistep.argtypes = [vp, vp, dbl, dbl, i64, C.c_int]
    L.rmt_n2_rk45.argtypes = [vp, vp, dbl, dbl, dbl, dbl, dbl, i64, vp]
    L.rmt_n2_ros4.argtypes = [vp, vp, dbl, dbl, dbl, dbl, dbl, i64, vp]
    L.rmt_n1_profile.argtypes = [vp, C.POINTER(dbl), vp, C.c_int, dbl, dbl, dbl, i64, vp]
    L.rmt_n2_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.rmt_n2_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.rmt_n2_last_geometry.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rmt_n2_fallbacks.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rmt_n2_hiprtc_path.restype = cp
    L.rmt_n2_compile_options.restype = cp
    if L.rmt_n2_abi_version() != ABI_VERSION:
        raise RmtN2Error("librmt_n2.so ABI %d != binding ABI %d" % (L.rmt_n2_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RmtN2Error(lib().rmt_n2_last_error().decode(errors="replace"))


def kernel_template():
    return lib().rmt_n2_kernel_template().decode()


def compile_source(source, arch="gfx950", extra_opts=""):
    """hipRTC compile -> bytes of the code object (works without a GPU)."""
    L = lib()
    code, size, log = C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = L.rmt_n2_compile(source.encode(), arch.encode(), extra_opts.encode(), C.byref(code),
                          C.byref(size), C.byref(log))
    logtxt = C.string_at(log.value).decode(errors="replace") if log.value else ""
    if log.value:
        L.rmt_n2_free(log)
    if rc != 0:
        raise RmtN2Error(L.rmt_n2_last_error().decode(errors="replace"))
    blob = C.string_at(code.value, size.value)
    L.rmt_n2_free(code)
    return blob, logtxt


_RTC_TAG = None


def hiprtc_tag():
    """Short identity of the hipRTC in this process.  The image has two - /opt/rocm's and the one bundled with
    PyTorch (used by every process that imported torch before this library); both call themselves 9.0 but
    generate different code - so the tag is derived from the library file that is actually loaded."""
    global _RTC_TAG
    if _RTC_TAG is None:
        import hashlib
        path = os.path.realpath(lib().rmt_n2_hiprtc_path().decode() or "unknown")
        try:
            size = os.path.getsize(path)
        except OSError:
            size = 0
        # ... and from the options the library compiles with by default
        _RTC_TAG = hashlib.sha256(("%s:%d:%s" % (path, size, lib().rmt_n2_compile_options().decode())).encode()
                                  ).hexdigest()[:8]
    return _RTC_TAG


def compile_cached(source, key, arch="gfx950", extra_opts=""):
    """Code objects are cached in-tree (rmt_app_amd/_kcache/<key>-rtc<compiler tag>-<arch>.hsaco): the directory
    travels with the repo snapshot, a cache under $HOME would not.  The compiler's identity is part of the name,
    so objects of the two hipRTCs of this image never stand in for each other."""
    os.makedirs(CACHE_DIR, exist_ok=True)
    key = "%s-rtc%s" % (key, hiprtc_tag())
    if extra_opts:
        import hashlib
        key = "%s-%s" % (key, hashlib.sha256(extra_opts.encode()).hexdigest()[:8])
    path = os.path.join(CACHE_DIR, "%s-%s.hsaco" % (key, arch))
    if os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    blob, _ = compile_source(source, arch, extra_opts)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, path)
    return blob
