"""ctypes loader for oracle/_build/liboracle.so (C part of the CPU oracle).

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, "csrc", f) for f in ("nmr_raster.c", "sdf.c", "objchain.c", "lbs_exact.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        ci, cf = ctypes.c_int, ctypes.c_float
        _lib.orc_nmr_face_index_map.argtypes = [fp, ci, ci, ci, cf, cf, ip, fp]
        _lib.orc_nmr_face_index_map.restype = None
        _lib.orc_nmr_grad_faces_alpha.argtypes = [fp, ip, fp, ci, ci, ci, cf, fp]
        _lib.orc_nmr_grad_faces_alpha.restype = None
        _lib.orc_nmr_grad_faces_depth.argtypes = [fp, ip, fp, ci, ci, ci, fp]
        _lib.orc_nmr_grad_faces_depth.restype = None
        _lib.orc_sdf_grid.argtypes = [ip, fp, ci, ci, ci, ci, ci, fp]
        _lib.orc_sdf_grid.restype = None
        _lib.orc_point_triangle_distance.argtypes = [fp, fp, fp, fp]
        _lib.orc_point_triangle_distance.restype = cf
        dp, vp = ctypes.POINTER(ctypes.c_double), ctypes.c_void_p
        _lib.orc_nmr_grad_faces_alpha_exact.argtypes = [fp, ip, fp, ci, ci, ci, cf, ci, dp]
        _lib.orc_nmr_grad_faces_alpha_exact.restype = None
        _lib.orc_rigid_bwd_sil_exact.argtypes = [fp, fp, cf, ci, vp, fp, ci, dp, ip, ip, fp, fp, cf, ci, ci, ci, ci, fp, fp,
                                                 fp, fp]
        _lib.orc_rigid_bwd_sil_exact.restype = None
        _lib.orc_mano_forward.argtypes = [fp, fp, fp, fp, fp, fp, fp, ip, fp, ci, fp, fp, ci, fp]
        _lib.orc_mano_forward.restype = None
        _lib.orc_hand_chain.argtypes = [fp, fp, fp, fp, fp, fp, fp, ip, fp, ci, fp, fp, fp, fp, cf, vp, fp, ci, fp, ci, cf, fp, cf, ci,
                                        fp, fp, fp, fp, fp, fp]
        _lib.orc_hand_chain.restype = None
        _lib.orc_v2d_unit_grad.argtypes = [fp, fp, fp, cf, ci, ci, ci, fp]
        _lib.orc_mano_bwd_rows.argtypes = [fp, fp, fp, fp, fp, fp, fp, ip, fp, ci, fp, fp, fp, fp, cf, ci, ci, ci, fp, fp, fp, fp]
        _lib.orc_mano_bwd_rows.restype = None
        _lib.orc_rigid_bwd_rows.argtypes = [fp, fp, cf, vp, fp, ci, fp, ci, cf, ci, ci, ci, fp, fp, fp]
        _lib.orc_rigid_bwd_rows.restype = None
        _lib.orc_v2d_unit_grad.restype = None
        _lib.orc_inter_rec.argtypes = [fp, fp, fp, ci, ci, ci, cf, cf, ci, fp]
        _lib.orc_inter_rec.restype = None
        _lib.orc_nn_search_d2.argtypes = [fp, fp, ci, ci, ci, ip, fp]
        _lib.orc_nn_search_d2.restype = None
        _lib.orc_hand_chain_rigid.argtypes = [fp, fp, fp, fp, fp, fp, fp, ip, fp, ci, fp, fp, fp, fp, cf, vp, fp, ci, fp, fp, cf, ci,
                                              fp, fp, fp, fp, fp, fp]
        _lib.orc_hand_chain_rigid.restype = None
        _lib.orc_nn_search.argtypes = [fp, fp, ci, ci, ci, ip]
        _lib.orc_nn_search.restype = None
        _lib.orc_contact_grads.argtypes = [fp, fp, ip, ci, ci, ci, cf, ci, fp, fp]
        _lib.orc_contact_grads.restype = None
        _lib.orc_sdf_sample_grad.argtypes = [fp, fp, fp, ci, ci, ci, fp]
        _lib.orc_sdf_sample_grad.restype = None
        u8p = ctypes.POINTER(ctypes.c_uint8)
        _lib.orc_ordinal_depth_grad.argtypes = [fp, fp, u8p, u8p, u8p, u8p, ci, ci, cf, fp, fp, fp]
        _lib.orc_ordinal_depth_grad.restype = None
        _lib.orc_depth_bwd_faces.argtypes = [fp, ip, fp, ci, ci, ci, fp]
        _lib.orc_depth_bwd_faces.restype = None
        _lib.orc_depth_bwd_gather.argtypes = [fp, ip, ip, fp, fp, ci, ci, ci, cf, fp]
        _lib.orc_depth_bwd_gather.restype = None
        _lib.orc_sigmoid.argtypes = [cf]
        _lib.orc_sigmoid.restype = cf
        _lib.orc_offscreen.argtypes = [fp, fp, ci, ci, cf, cf, ci, fp, fp]
        _lib.orc_offscreen.restype = None
        _lib.orc_block_sum.argtypes = [fp, ci, ci]
        _lib.orc_block_sum.restype = cf
        _lib.orc_tanh.argtypes = [cf]
        _lib.orc_tanh.restype = cf
        _lib.orc_sincos.argtypes = [cf, fp, fp]
        _lib.orc_sincos.restype = None
        _lib.orc_sum_magic.argtypes = [ci]
        _lib.orc_sum_magic.restype = ctypes.c_double
    return _lib


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def dptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
