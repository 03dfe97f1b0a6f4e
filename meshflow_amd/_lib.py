"""ctypes binding of libmeshflow_hip.so (C ABI: include/meshflow_hip.h).

There is no CPU fallback: if the shared library is missing, importing this module raises, and every
compute call fails with the library's own error when no GPU is present."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmeshflow_hip.so')

MF_OK = 0
MF_ERR_INVALID_ARG = -1
MF_ERR_HIP = -2
MF_ERR_DEGENERATE = -3
CELL_DOUBLES = 32
CELL_OFF_M, CELL_OFF_HI, CELL_OFF_RECT, CELL_OFF_BBOX, CELL_OFF_STATUS = 0, 9, 18, 22, 26

_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/meshflow_hip.h declares.
SIGNATURES = {
    'mf_abi_version': (_i, []),
    'mf_last_error': (ctypes.c_char_p, []),
    'mf_device_count': (_i, [ctypes.POINTER(_i)]),
    'mf_set_device': (_i, [_i]),
    'mf_malloc': (_i, [ctypes.POINTER(_vp), _sz]),
    'mf_free': (_i, [_vp]),
    'mf_malloc_host': (_i, [ctypes.POINTER(_vp), _sz]),
    'mf_free_host': (_i, [_vp]),
    'mf_memcpy_h2d': (_i, [_vp, _vp, _sz, _vp]),
    'mf_memcpy_d2h': (_i, [_vp, _vp, _sz, _vp]),
    'mf_stream_synchronize': (_i, [_vp]),
    'mf_jacobi_f64': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mf_cell_table_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'mf_cell_table_bounds_offset': (_sz, [_i, _i, _i, _i, _i]),
    'mf_cell_table_f64': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'mf_warp_u8c3': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mf_crop_scan_f64': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'mf_cell_table_bounds_f64': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'mf_warp_bounds_u8c3': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'mf_crop_scan_bounds_f64': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mf_warp_clip_u8c3': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'mf_crop_reduce': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'mf_crop_resize_workspace_bytes': (_sz, [_i, _i]),
    'mf_crop_resize_u8c3': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mf_vertex_motion_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'mf_vertex_motion_f64': (_i, [_vp, _vp, _vp, _vp] + [_i] * 9 + [_vp, _vp, _vp, _vp, _vp]),
    'mf_stability_score_f64': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'mf_selftest_sqrt': (_i, [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    'mf_selftest_recip': (_i, [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    'mf_selftest_fast64': (_i, [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    'mf_selftest_fast64_margin': (_i, [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double)]),
    'mf_jacobi_f64_host': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_float)]),
    'mf_warp_u8c3_host': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    'mf_warp_u8c3_host_frames': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    'mf_warp_crop_u8c3_host_frames': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    'mf_crop_resize_u8c3_host_frames': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_float)]),
    'mf_host_cache_release': (_i, []),
    'mf_comm_init_all': (_i, [_i]),
    'mf_comm_size': (_i, [ctypes.POINTER(_i)]),
    'mf_allreduce_crop': (_i, [_vp]),
    'mf_gather_frames': (_i, [_vp, _vp, _vp, _i]),
    'mf_comm_destroy': (_i, []),
}


class MeshflowHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libmeshflow_hip error {code}: {message}')
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C meshflow_amd/csrc` (needs hipcc). There is no CPU fallback.')
    # The process must hold ONE HIP runtime.  torch ships its own libamdhip64; when this library is loaded first it pulls
    # in /opt/rocm's copy, initialises the GPU through it, and torch then finds "no HIP GPUs" (observed on MI355X / ROCm 7).
    # Loading torch first makes the dynamic loader bind this library to the runtime torch already holds.
    try:
        import torch  # noqa: F401
    except ImportError:      # a pure-ctypes host without torch: /opt/rocm's runtime is the only one
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mf_abi_version() != 1:
        raise ImportError('libmeshflow_hip.so ABI version mismatch; rebuild it')
    return lib


lib = _load()


def check(code):
    """Raise for a negative status (ValueError for bad arguments / degenerate meshes, like the Python
    exceptions the reference's NumPy/OpenCV calls would raise)."""
    if code == MF_OK:
        return
    msg = lib.mf_last_error().decode('utf-8', 'replace')
    if code in (MF_ERR_INVALID_ARG, MF_ERR_DEGENERATE):
        raise ValueError(f'libmeshflow_hip: {msg}')
    raise MeshflowHipError(code, msg)
