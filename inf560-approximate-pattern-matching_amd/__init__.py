"""ctypes binding of the C ABI in include/apm.h (libapm_hip.so).

This is glue for tests/ and bench.py only: the product is the C ABI and the C
host `host/apm_parallel`.  It mirrors the ABI one to one (same names, argument
meaning and error behaviour) and raises ApmError on any negative apm_status.
There is NO CPU fallback: if the HIP library is missing or no device is
visible, everything here fails loudly.

Load order matters in a process that also uses torch: torch bundles its own
libamdhip64.so.7; importing torch FIRST makes libapm_hip.so bind to that same
runtime instance (same SONAME), so torch device pointers and streams can be
handed to the C ABI.  The package therefore imports torch (if present) before
dlopen()ing the library.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APM_LIB_PATH") or os.path.join(_HERE, "libapm_hip.so")  # APM_LIB_PATH: A/B builds

APM_KERNEL_AUTO, APM_KERNEL_GENERIC, APM_KERNEL_WAVEFRONT, APM_KERNEL_BITPAR, APM_KERNEL_BANDED, APM_KERNEL_NFA = range(6)
KERNEL_NAMES = {0: "auto", 1: "generic", 2: "wavefront", 3: "bitpar", 4: "banded", 5: "nfa"}
KERNEL_IDS = {v: k for k, v in KERNEL_NAMES.items()}

# every symbol include/apm.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "apm_device_count", "apm_abi_version", "apm_create", "apm_create_on_device", "apm_destroy",
    "apm_last_error", "apm_set_stream", "apm_set_patterns", "apm_set_kernel", "apm_set_partition", "apm_count_buffer",
    "apm_count_file", "apm_find_buffer", "apm_count_shard_device", "apm_shard_range", "apm_synth_fill_device",
    "apm_synth_fill_host", "apm_count_synthetic", "apm_set_timing", "apm_get_timing", "apm_get_launch_times", "apm_get_stat",
    "apm_pattern_kernel",
    "apm_device_alloc", "apm_device_free", "apm_device_upload", "apm_device_download",
    "apm_device_memset", "apm_synchronize",
]


class ApmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("apm status %d: %s" % (status, message))
        self.status = status


class ApmTiming(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("h2d_ms", ctypes.c_double), ("kernel_ms", ctypes.c_double),
                ("reduce_ms", ctypes.c_double), ("main_kernel_ms", ctypes.c_double),
                ("text_bytes", ctypes.c_uint64), ("windows", ctypes.c_uint64),
                ("cells_algorithmic", ctypes.c_double), ("cells_evaluated", ctypes.c_double),
                ("n_devices", ctypes.c_int), ("n_launches", ctypes.c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load_library():
    """dlopen libapm_hip.so (after torch, see module docstring)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ApmError(-2, "HIP extension %s is missing: run `make -C %s` (no CPU fallback exists)" % (LIB_PATH, _HERE))
    try:
        import torch  # noqa: F401  (binds libamdhip64.so.7 first)
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    c = ctypes
    vp, u64, i32 = c.c_void_p, c.c_uint64, c.c_int
    sig = {
        "apm_device_count": (i32, []),
        "apm_abi_version": (i32, []),
        "apm_create": (i32, [c.POINTER(vp), i32]),
        "apm_create_on_device": (i32, [c.POINTER(vp), i32]),
        "apm_destroy": (None, [vp]),
        "apm_last_error": (c.c_char_p, [vp]),
        "apm_set_stream": (i32, [vp, vp]),
        "apm_set_patterns": (i32, [vp, i32, c.POINTER(c.c_char_p), c.POINTER(i32), i32]),
        "apm_set_kernel": (i32, [vp, i32]),
        "apm_set_partition": (i32, [vp, i32]),
        "apm_count_buffer": (i32, [vp, vp, u64, c.POINTER(u64)]),
        "apm_count_file": (i32, [vp, c.c_char_p, c.POINTER(u64)]),
        "apm_find_buffer": (i32, [vp, vp, u64, i32, c.POINTER(u64), u64, c.POINTER(u64)]),
        "apm_count_shard_device": (i32, [vp, vp, u64, u64, u64, u64, u64, vp]),
        "apm_shard_range": (i32, [u64, i32, i32, i32, c.POINTER(u64), c.POINTER(u64)]),
        "apm_synth_fill_device": (i32, [vp, vp, u64, u64, u64]),
        "apm_synth_fill_host": (None, [vp, u64, u64, u64]),
        "apm_count_synthetic": (i32, [vp, u64, u64, c.POINTER(u64)]),
        "apm_set_timing": (i32, [vp, i32]),
        "apm_get_timing": (i32, [vp, c.POINTER(ApmTiming)]),
        "apm_get_launch_times": (i32, [vp, i32, c.POINTER(c.c_double), c.POINTER(c.c_char_p)]),
        "apm_get_stat": (i32, [vp, c.c_char_p, c.POINTER(c.c_double)]),
        "apm_pattern_kernel": (i32, [vp, i32]),
        "apm_device_alloc": (i32, [vp, c.POINTER(vp), u64]),
        "apm_device_free": (i32, [vp, vp]),
        "apm_device_upload": (i32, [vp, vp, vp, u64]),
        "apm_device_download": (i32, [vp, vp, vp, u64]),
        "apm_device_memset": (i32, [vp, vp, i32, u64]),
        "apm_synchronize": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count():
    n = load_library().apm_device_count()
    if n < 0:
        raise ApmError(n, (load_library().apm_last_error(None) or b"").decode())
    return n


def shard_range(n_total, k, shard, n_shards):
    b, e = ctypes.c_uint64(), ctypes.c_uint64()
    rc = load_library().apm_shard_range(n_total, k, shard, n_shards, ctypes.byref(b), ctypes.byref(e))
    if rc:
        raise ApmError(rc, "apm_shard_range: invalid argument")
    return b.value, e.value


def synth_fill_host(global_off, length, seed):
    """The synthetic DNA generator's bytes for [global_off, global_off+length) (host side)."""
    buf = ctypes.create_string_buffer(length)
    load_library().apm_synth_fill_host(ctypes.cast(buf, ctypes.c_void_p), global_off, length, seed)
    return buf.raw


class ApmContext:
    """Thin object wrapper over apm_ctx*.  device=None -> apm_create(n_devices)."""

    def __init__(self, n_devices=1, device=None):
        self._lib = load_library()
        self._ctx = ctypes.c_void_p()
        if device is None:
            rc = self._lib.apm_create(ctypes.byref(self._ctx), n_devices)
        else:
            rc = self._lib.apm_create_on_device(ctypes.byref(self._ctx), device)
        if rc:
            raise ApmError(rc, (self._lib.apm_last_error(None) or b"").decode())
        self.n_patterns = 0

    # -- plumbing --
    def _check(self, rc):
        if rc:
            raise ApmError(rc, (self._lib.apm_last_error(self._ctx) or b"").decode())

    def close(self):
        if self._ctx:
            self._lib.apm_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ABI mirror --
    def set_patterns(self, patterns, k):
        pats = [bytes(p) for p in patterns]
        n = len(pats)
        arr = (ctypes.c_char_p * max(n, 1))(*pats)
        lens = (ctypes.c_int * max(n, 1))(*[len(p) for p in pats])
        self._check(self._lib.apm_set_patterns(self._ctx, n, arr, lens, k))
        self.n_patterns = n

    def set_kernel(self, kernel):
        if isinstance(kernel, str):
            kernel = KERNEL_IDS[kernel]
        self._check(self._lib.apm_set_kernel(self._ctx, kernel))

    def set_partition(self, partition):
        """"text" (default: owner ranges of the text, counts summed) or "patterns" (slices of the pattern list, every
        device scans the whole text) -- multi-device contexts only"""
        self._check(self._lib.apm_set_partition(self._ctx, {"text": 0, "patterns": 1}.get(partition, partition)))

    def set_stream(self, stream_ptr):
        self._check(self._lib.apm_set_stream(self._ctx, ctypes.c_void_p(stream_ptr)))

    def pattern_kernel(self, i):
        return self._lib.apm_pattern_kernel(self._ctx, i)

    def count_buffer(self, text):
        text = bytes(text)
        out = (ctypes.c_uint64 * max(self.n_patterns, 1))()
        buf = ctypes.create_string_buffer(text, len(text)) if text else None
        ptr = ctypes.cast(buf, ctypes.c_void_p) if text else ctypes.c_void_p()
        self._check(self._lib.apm_count_buffer(self._ctx, ptr, len(text), out))
        return list(out)[: self.n_patterns]

    def find_buffer(self, text, pattern_index, capacity=1 << 16):
        """(positions ascending, total matches) of one pattern of the current set"""
        text = bytes(text)
        out = (ctypes.c_uint64 * max(capacity, 1))()
        found = ctypes.c_uint64()
        buf = ctypes.create_string_buffer(text, len(text)) if text else None
        ptr = ctypes.cast(buf, ctypes.c_void_p) if text else ctypes.c_void_p()
        self._check(self._lib.apm_find_buffer(self._ctx, ptr, len(text), pattern_index, out, capacity,
                                              ctypes.byref(found)))
        return list(out)[: min(found.value, capacity)], found.value

    def count_file(self, path):
        out = (ctypes.c_uint64 * max(self.n_patterns, 1))()
        self._check(self._lib.apm_count_file(self._ctx, os.fsencode(path), out))
        return list(out)[: self.n_patterns]

    def count_synthetic(self, n, seed):
        out = (ctypes.c_uint64 * max(self.n_patterns, 1))()
        self._check(self._lib.apm_count_synthetic(self._ctx, n, seed, out))
        return list(out)[: self.n_patterns]

    def count_shard_device(self, d_text, text_off, text_len, n_total, own_begin, own_end, d_counts):
        self._check(self._lib.apm_count_shard_device(self._ctx, ctypes.c_void_p(d_text), text_off, text_len,
                                                     n_total, own_begin, own_end, ctypes.c_void_p(d_counts)))

    def synth_fill_device(self, d_dst, global_off, length, seed):
        self._check(self._lib.apm_synth_fill_device(self._ctx, ctypes.c_void_p(d_dst), global_off, length, seed))

    def set_timing(self, enabled):
        self._check(self._lib.apm_set_timing(self._ctx, 1 if enabled else 0))

    def timing(self):
        t = ApmTiming()
        self._check(self._lib.apm_get_timing(self._ctx, ctypes.byref(t)))
        return t.as_dict()

    def launch_times(self, max_launches=32):
        """[(label, ms)] of the scan-kernel launches of the last counting call (HIP events on the launch stream)"""
        ms = (ctypes.c_double * max_launches)()
        labels = (ctypes.c_char_p * max_launches)()
        n = self._lib.apm_get_launch_times(self._ctx, max_launches, ms, labels)
        if n < 0:
            self._check(n)
        return [((labels[i] or b"?").decode(), ms[i]) for i in range(n)]

    def stat(self, name):
        v = ctypes.c_double()
        self._check(self._lib.apm_get_stat(self._ctx, name.encode(), ctypes.byref(v)))
        return v.value

    def synchronize(self):
        self._check(self._lib.apm_synchronize(self._ctx))

    def device_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self._lib.apm_device_alloc(self._ctx, ctypes.byref(p), nbytes))
        return p.value

    def device_free(self, ptr):
        self._check(self._lib.apm_device_free(self._ctx, ctypes.c_void_p(ptr)))

    def device_upload(self, d_dst, data):
        data = bytes(data)
        self._check(self._lib.apm_device_upload(self._ctx, ctypes.c_void_p(d_dst), data, len(data)))

    def device_download(self, d_src, nbytes):
        buf = ctypes.create_string_buffer(nbytes)
        self._check(self._lib.apm_device_download(self._ctx, ctypes.cast(buf, ctypes.c_void_p),
                                                  ctypes.c_void_p(d_src), nbytes))
        return buf.raw

    def device_memset(self, d_dst, value, nbytes):
        self._check(self._lib.apm_device_memset(self._ctx, ctypes.c_void_p(d_dst), value, nbytes))
