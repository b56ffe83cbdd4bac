"""ctypes binding of libarchive_hip.so (include/archive_hip.h).

There is deliberately no fallback: if the shared library is missing or no GPU is usable,
calls raise -- nothing in this package decodes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AHIP_LIB") or os.path.join(_HERE, "lib", "libarchive_hip.so")

AHIP_OK, AHIP_FALSE, AHIP_RANGE, AHIP_HANG = 0, 1, 2, 3
AHIP_E_CAP, AHIP_E_DEVICE, AHIP_E_UNSUPPORTED, AHIP_E_ARG = -1, -2, -3, -4

# every symbol include/archive_hip.h declares
EXPORTS = [
    "ahip_init", "ahip_init_devices", "ahip_device_count", "ahip_debug_last_shards", "ahip_shutdown", "ahip_last_error", "ahip_abi_version",
    "ahip_inflate_raw", "ahip_gzip_decode", "ahip_zlib_decode",
    "ahip_gzip_decode_device", "ahip_gzip_plan_create", "ahip_gzip_plan_info", "ahip_gzip_plan_run",
    "ahip_gzip_plan_status", "ahip_gzip_plan_destroy", "ahip_debug_plan_results",
    "ahip_bzip2_decode", "ahip_bzip2_decode_device", "ahip_crc32_device", "ahip_adler32_device", "ahip_inflate_batch", "ahip_inflate_batch_device", "ahip_deflate_raw", "ahip_gzip_encode", "ahip_zlib_encode", "ahip_deflate_raw_device", "ahip_deflate_bound",
    "ahip_crc32", "ahip_adler32", "ahip_decode_bound",
    "ahip_gzip_decode_shards", "ahip_debug_last_exchange", "ahip_gzip_encode_device", "ahip_zlib_encode_device",
    "ahip_debug_last_chunks", "ahip_deflate_shards", "ahip_bzip2_decode_shards", "ahip_debug_bz_reruns", "ahip_last_consumed",
    "ahip_stream_split_create", "ahip_stream_split_candidates", "ahip_stream_split_size", "ahip_stream_split_chain", "ahip_stream_split_map_bytes",
    "ahip_stream_split_resolve", "ahip_stream_split_finish", "ahip_stream_split_destroy", "ahip_debug_stream_split_chain", "ahip_inflate_stream_shards", "ahip_deflate_piece_device", "ahip_bzip2_decode_range_device",
]

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "%s not built -- run `python -m archive_amd.build` (needs hipcc); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: when PyTorch-ROCm is installed its bundled libamdhip64 must be
    # the one this library binds to (same SONAME), otherwise device pointers and streams could not
    # be shared with torch and the second runtime to start finds no GPU.  Dart/FFI processes have
    # no torch and simply use the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C-ABI
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i32, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64
    szp = ctypes.POINTER(sz)
    L.ahip_init.argtypes = [i32]; L.ahip_init.restype = i32
    L.ahip_init_devices.argtypes = [u64]; L.ahip_init_devices.restype = i32
    L.ahip_device_count.argtypes = []; L.ahip_device_count.restype = i32
    L.ahip_debug_last_shards.argtypes = []; L.ahip_debug_last_shards.restype = i32
    L.ahip_debug_last_chunks.argtypes = []; L.ahip_debug_last_chunks.restype = i32
    L.ahip_shutdown.argtypes = []; L.ahip_shutdown.restype = None
    L.ahip_last_error.argtypes = []; L.ahip_last_error.restype = ctypes.c_char_p
    L.ahip_abi_version.argtypes = []; L.ahip_abi_version.restype = u32
    L.ahip_inflate_raw.argtypes = [vp, sz, vp, sz, szp, szp]; L.ahip_inflate_raw.restype = i32
    L.ahip_gzip_decode.argtypes = [vp, sz, i32, i32, vp, sz, szp]; L.ahip_gzip_decode.restype = i32
    L.ahip_zlib_decode.argtypes = [vp, sz, i32, i32, vp, sz, szp]; L.ahip_zlib_decode.restype = i32
    L.ahip_gzip_decode_device.argtypes = [vp, sz, vp, sz, szp, vp]; L.ahip_gzip_decode_device.restype = i32
    L.ahip_gzip_plan_create.argtypes = [vp, sz, vp, ctypes.POINTER(vp)]; L.ahip_gzip_plan_create.restype = i32
    L.ahip_gzip_plan_info.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.ahip_gzip_plan_info.restype = i32
    L.ahip_gzip_plan_run.argtypes = [vp, vp, sz, vp]; L.ahip_gzip_plan_run.restype = i32
    L.ahip_gzip_plan_status.argtypes = [vp, szp]; L.ahip_gzip_plan_status.restype = i32
    L.ahip_gzip_plan_destroy.argtypes = [vp]; L.ahip_gzip_plan_destroy.restype = None
    L.ahip_debug_plan_results.argtypes = [vp, vp, sz, szp]; L.ahip_debug_plan_results.restype = i32
    L.ahip_bzip2_decode.argtypes = [vp, sz, i32, vp, sz, szp]; L.ahip_bzip2_decode.restype = i32
    L.ahip_inflate_batch.argtypes = [vp, sz, u32, vp, vp, vp, vp, sz, vp, vp, vp, szp]; L.ahip_inflate_batch.restype = i32
    L.ahip_inflate_batch_device.argtypes = [vp, sz, u32, vp, vp, vp, vp, sz, vp, vp, vp, szp, vp]; L.ahip_inflate_batch_device.restype = i32
    L.ahip_crc32_device.argtypes = [vp, sz, u32, ctypes.POINTER(u32), vp]; L.ahip_crc32_device.restype = i32
    L.ahip_adler32_device.argtypes = [vp, sz, u32, ctypes.POINTER(u32), vp]; L.ahip_adler32_device.restype = i32
    L.ahip_bzip2_decode_device.argtypes = [vp, sz, i32, vp, sz, szp, vp]; L.ahip_bzip2_decode_device.restype = i32
    L.ahip_deflate_raw.argtypes = [vp, sz, i32, i32, vp, sz, szp, ctypes.POINTER(u32)]; L.ahip_deflate_raw.restype = i32
    L.ahip_gzip_encode.argtypes = [vp, sz, i32, i32, u32, vp, sz, szp]; L.ahip_gzip_encode.restype = i32
    L.ahip_zlib_encode.argtypes = [vp, sz, i32, i32, vp, sz, szp]; L.ahip_zlib_encode.restype = i32
    L.ahip_deflate_raw_device.argtypes = [vp, sz, i32, i32, vp, sz, szp, vp]; L.ahip_deflate_raw_device.restype = i32
    L.ahip_deflate_bound.argtypes = [sz]; L.ahip_deflate_bound.restype = sz
    L.ahip_decode_bound.argtypes = [vp, sz]; L.ahip_decode_bound.restype = sz
    L.ahip_gzip_decode_shards.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp, vp]; L.ahip_gzip_decode_shards.restype = i32
    L.ahip_deflate_shards.argtypes = [u32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]; L.ahip_deflate_shards.restype = i32
    L.ahip_bzip2_decode_shards.argtypes = [u32, vp, vp, sz, i32, vp, vp, vp, vp, vp]; L.ahip_bzip2_decode_shards.restype = i32
    L.ahip_debug_last_exchange.argtypes = []; L.ahip_debug_last_exchange.restype = i32
    L.ahip_debug_bz_reruns.argtypes = []; L.ahip_debug_bz_reruns.restype = i32
    L.ahip_last_consumed.argtypes = []; L.ahip_last_consumed.restype = sz
    L.ahip_gzip_encode_device.argtypes = [vp, sz, i32, i32, u32, vp, sz, szp, vp]; L.ahip_gzip_encode_device.restype = i32
    L.ahip_zlib_encode_device.argtypes = [vp, sz, i32, i32, vp, sz, szp, vp]; L.ahip_zlib_encode_device.restype = i32
    i32p, u64p = ctypes.POINTER(i32), ctypes.POINTER(u64)
    L.ahip_stream_split_create.argtypes = [vp, sz, sz, u32, u32, vp, ctypes.POINTER(vp)]; L.ahip_stream_split_create.restype = i32
    L.ahip_stream_split_candidates.argtypes = [vp, vp, sz, szp]; L.ahip_stream_split_candidates.restype = i32
    L.ahip_stream_split_size.argtypes = [vp, vp, sz, vp, sz, i32p]; L.ahip_stream_split_size.restype = i32
    L.ahip_stream_split_chain.argtypes = [vp, vp, sz, i32p, u64p, u64p, u64p, u64p]; L.ahip_stream_split_chain.restype = i32
    L.ahip_stream_split_map_bytes.argtypes = []; L.ahip_stream_split_map_bytes.restype = sz
    L.ahip_stream_split_resolve.argtypes = [vp, vp]; L.ahip_stream_split_resolve.restype = i32
    L.ahip_stream_split_finish.argtypes = [vp, vp, vp, sz, szp, i32p]; L.ahip_stream_split_finish.restype = i32
    L.ahip_stream_split_destroy.argtypes = [vp]; L.ahip_stream_split_destroy.restype = None
    L.ahip_inflate_stream_shards.argtypes = [u32, vp, vp, sz, sz, vp, vp, vp, vp, u64p, i32p]; L.ahip_inflate_stream_shards.restype = i32
    L.ahip_deflate_piece_device.argtypes = [vp, sz, i32, i32, i32, vp, sz, szp, ctypes.POINTER(u32), vp]; L.ahip_deflate_piece_device.restype = i32
    L.ahip_bzip2_decode_range_device.argtypes = [vp, sz, i32, u32, u32, u64, vp, sz, szp, vp, vp]; L.ahip_bzip2_decode_range_device.restype = i32
    L.ahip_debug_stream_split_chain.argtypes = [vp, vp, sz, u32, u32, vp]; L.ahip_debug_stream_split_chain.restype = i32
    L.ahip_crc32.argtypes = [vp, sz, u32]; L.ahip_crc32.restype = u32
    L.ahip_adler32.argtypes = [vp, sz, u32]; L.ahip_adler32.restype = u32
    _lib = L
    return L


def last_error():
    return lib().ahip_last_error().decode("utf-8", "replace")
