"""ctypes binding of libldpc_toolbox.so (include/ldpc_toolbox.h).

This is the same stub a maintainer of a Python program using the reference's cdylib would
write; it binds the nine reference symbols and the batched extension.  There is no
fallback: if the HIP library is missing, importing callers get an ImportError, and on a
machine without a GPU the constructors return NULL (surfaced as DecoderUnavailable).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDPC_TOOLBOX_LIB: load another build of the same library (A/B timing of two builds)
LIB_PATH = os.environ.get("LDPC_TOOLBOX_LIB") or os.path.join(_HERE, "lib", "libldpc_toolbox.so")

# every symbol include/ldpc_toolbox.h declares (tests check the library exports all of them)
SYMBOLS = [
    "ldpc_toolbox_decoder_ctor",
    "ldpc_toolbox_decoder_ctor_alist_string",
    "ldpc_toolbox_decoder_dtor",
    "ldpc_toolbox_decoder_decode_f64",
    "ldpc_toolbox_decoder_decode_f32",
    "ldpc_toolbox_encoder_ctor",
    "ldpc_toolbox_encoder_ctor_alist_string",
    "ldpc_toolbox_encoder_dtor",
    "ldpc_toolbox_encoder_encode",
    "ldpc_toolbox_decoder_ctor_alist_string_on_device",
    "ldpc_toolbox_decoder_decode_batch_f32",
    "ldpc_toolbox_decoder_decode_batch_f64",
    "ldpc_toolbox_decoder_decode_batch_f32_device",
    "ldpc_toolbox_decoder_decode_batch_f64_device",
    "ldpc_toolbox_decoder_syndrome",
    "ldpc_toolbox_decoder_syndrome_device",
    "ldpc_toolbox_decoder_get",
    "ldpc_toolbox_decoder_set",
    "ldpc_toolbox_decoder_kernel_stats",
    "ldpc_toolbox_sim_ctor",
    "ldpc_toolbox_sim_dtor",
    "ldpc_toolbox_sim_run",
    "ldpc_toolbox_sim_run_bch",
    "ldpc_toolbox_sim_generate",
    "ldpc_toolbox_sim_pool",
    "ldpc_toolbox_sim_get",
    "ldpc_toolbox_sim_set",
    "ldpc_toolbox_code_alist",
    "ldpc_toolbox_alist_normalize",
    "ldpc_toolbox_device_count",
    "ldpc_toolbox_last_error",
]

_lib = None


def _hip_sonames(path):
    """the libamdhip64.so.N names an ELF file carries in its string tables (its own SONAME, or what it NEEDs)"""
    import re
    names = set()
    try:
        with open(path, "rb") as f:
            tail = b""
            while True:
                chunk = f.read(1 << 22)
                if not chunk:
                    break
                names.update(m.decode() for m in re.findall(rb"libamdhip64\.so\.\d+", tail + chunk))
                tail = chunk[-32:]
    except OSError:
        pass
    return names


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order.  PyTorch's ROCm wheel bundles its own libamdhip64.so.7 (and
    the HSA runtime beside it); libldpc_toolbox.so names the same soname and finds the system's through its RUNPATH.  The
    first one loaded serves both -- and if that is the system's, a later `import torch` pairs it with torch's bundled HSA
    libraries and reports "No HIP GPUs are available".  So: when a torch installation is present and not loaded yet, its
    runtime is loaded first (without importing torch); the library then binds to it, exactly as when torch came first.
    Only when the wheel's runtime carries the SONAME this library needs: with another major version the pre-load would put
    TWO runtimes into the process -- the situation this function exists to avoid -- so it is skipped, with a warning.
    LDPC_TOOLBOX_SYSTEM_HIP=1 keeps the system runtime (a process that will never import torch; a C caller of the library
    gets the system runtime in any case)."""
    import sys
    if "torch" in sys.modules or os.environ.get("LDPC_TOOLBOX_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if not os.path.exists(path):
            return
        need, have = _hip_sonames(LIB_PATH), _hip_sonames(path)
        if need and have and not (need & have):
            import warnings
            warnings.warn(f"ldpc_toolbox_amd: torch bundles {sorted(have)} but {os.path.basename(LIB_PATH)} needs {sorted(need)}: "
                          "not pre-loading torch's HIP runtime; import torch first, or not at all, in this process")
            return
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:
        import warnings
        warnings.warn(f"ldpc_toolbox_amd: could not pre-load torch's HIP runtime ({e}); the library's own RUNPATH is used")
    except Exception:
        pass      # the library's own RUNPATH still finds a runtime


def lib():
    """Loads the HIP library; raises ImportError (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ldpc_toolbox_amd/csrc` (there is no CPU fallback)")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, cp, sz, u32, i32 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int32
    L.ldpc_toolbox_decoder_ctor.restype = vp
    L.ldpc_toolbox_decoder_ctor.argtypes = [cp, cp, cp]
    L.ldpc_toolbox_decoder_ctor_alist_string.restype = vp
    L.ldpc_toolbox_decoder_ctor_alist_string.argtypes = [cp, cp, cp]
    L.ldpc_toolbox_decoder_ctor_alist_string_on_device.restype = vp
    L.ldpc_toolbox_decoder_ctor_alist_string_on_device.argtypes = [cp, cp, cp, i32]
    L.ldpc_toolbox_decoder_dtor.restype = None
    L.ldpc_toolbox_decoder_dtor.argtypes = [vp]
    for name in ("f64", "f32"):
        f = getattr(L, "ldpc_toolbox_decoder_decode_" + name)
        f.restype = i32
        f.argtypes = [vp, vp, sz, vp, sz, u32]
        f = getattr(L, "ldpc_toolbox_decoder_decode_batch_" + name)
        f.restype = i32
        f.argtypes = [vp, vp, sz, vp, sz, sz, u32, vp, vp]
        f = getattr(L, "ldpc_toolbox_decoder_decode_batch_" + name + "_device")
        f.restype = i32
        f.argtypes = [vp, vp, sz, vp, sz, sz, u32, vp, vp, vp]
    L.ldpc_toolbox_encoder_ctor.restype = vp
    L.ldpc_toolbox_encoder_ctor.argtypes = [cp, cp]
    L.ldpc_toolbox_encoder_ctor_alist_string.restype = vp
    L.ldpc_toolbox_encoder_ctor_alist_string.argtypes = [cp, cp]
    L.ldpc_toolbox_encoder_dtor.restype = None
    L.ldpc_toolbox_encoder_dtor.argtypes = [vp]
    L.ldpc_toolbox_encoder_encode.restype = None
    L.ldpc_toolbox_encoder_encode.argtypes = [vp, vp, sz, vp, sz]
    L.ldpc_toolbox_decoder_syndrome.restype = i32
    L.ldpc_toolbox_decoder_syndrome.argtypes = [vp, vp, sz, sz, vp, vp]
    L.ldpc_toolbox_decoder_syndrome_device.restype = i32
    L.ldpc_toolbox_decoder_syndrome_device.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.ldpc_toolbox_decoder_get.restype = i32
    L.ldpc_toolbox_decoder_get.argtypes = [vp, cp, C.POINTER(C.c_int64)]
    L.ldpc_toolbox_decoder_set.restype = i32
    L.ldpc_toolbox_decoder_set.argtypes = [vp, cp, C.c_int64]
    L.ldpc_toolbox_decoder_kernel_stats.restype = i32
    L.ldpc_toolbox_decoder_kernel_stats.argtypes = [vp, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_double), i32]
    u64 = C.c_uint64
    L.ldpc_toolbox_sim_ctor.restype = vp
    L.ldpc_toolbox_sim_ctor.argtypes = [cp, cp, cp, i32, u32, u64]
    L.ldpc_toolbox_sim_dtor.restype = None
    L.ldpc_toolbox_sim_dtor.argtypes = [vp]
    L.ldpc_toolbox_sim_run.restype = i32
    L.ldpc_toolbox_sim_run.argtypes = [vp, C.c_double, u64, u64, sz, u32, vp]
    L.ldpc_toolbox_sim_run_bch.restype = i32
    L.ldpc_toolbox_sim_run_bch.argtypes = [vp, C.c_double, u64, u64, sz, u32, u64, vp]
    L.ldpc_toolbox_sim_generate.restype = i32
    L.ldpc_toolbox_sim_generate.argtypes = [vp, C.c_double, u64, u64, sz, vp, vp]
    L.ldpc_toolbox_sim_pool.restype = i32
    L.ldpc_toolbox_sim_pool.argtypes = [vp, vp, vp]
    L.ldpc_toolbox_sim_get.restype = i32
    L.ldpc_toolbox_sim_get.argtypes = [vp, cp, C.POINTER(C.c_int64)]
    L.ldpc_toolbox_sim_set.restype = i32
    L.ldpc_toolbox_sim_set.argtypes = [vp, cp, C.c_int64]
    L.ldpc_toolbox_code_alist.restype = sz
    L.ldpc_toolbox_code_alist.argtypes = [cp, vp, sz]
    L.ldpc_toolbox_alist_normalize.restype = sz
    L.ldpc_toolbox_alist_normalize.argtypes = [cp, i32, vp, sz]
    L.ldpc_toolbox_device_count.restype = i32
    L.ldpc_toolbox_device_count.argtypes = []
    L.ldpc_toolbox_last_error.restype = cp
    L.ldpc_toolbox_last_error.argtypes = []
    _lib = L
    return L


def last_error() -> str:
    return lib().ldpc_toolbox_last_error().decode(errors="replace")


def code_alist(spec: str) -> str:
    """alist text of a standard code: "dvbs2:R1_2", "nr5g:1:384", "ar4ja:1/2:1024", "c2"."""
    L = lib()
    need = L.ldpc_toolbox_code_alist(spec.encode(), None, 0)
    if need == 0:
        raise ValueError(f"unknown code spec {spec!r}")
    buf = C.create_string_buffer(need + 1)
    L.ldpc_toolbox_code_alist(spec.encode(), buf, need + 1)
    return buf.value.decode()


def alist_normalize(alist: str, padding: bool = True) -> str:
    """Parse + re-write an alist with the host library's SparseMatrix."""
    L = lib()
    need = L.ldpc_toolbox_alist_normalize(alist.encode(), int(padding), None, 0)
    if need == 0:
        raise ValueError(last_error() or "malformed alist")
    buf = C.create_string_buffer(need + 1)
    L.ldpc_toolbox_alist_normalize(alist.encode(), int(padding), buf, need + 1)
    return buf.value.decode()
