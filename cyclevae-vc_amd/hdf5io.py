"""The reference's HDF5 helpers (src/utils/utils.py:18-126: `check_hdf5`, `read_hdf5`, `shape_hdf5`, `write_hdf5`) on the HDF5 C
library itself, bound with ctypes -- no h5py needed.  (the recipe's feature files `*.h5` with
`/feat_org_lf0 [T,54]`, `/cvuvlogf0fil_ap [T,4]`, `/spcidx_range [1,S]`, and the statistics files with `/mean_feat_org_lf0_jnt` ...)

h5py is a wrapper around the same C library and `create_dataset(path, data=array)` uses the library's default creation
properties, so a file written here (H5Dcreate2 with default properties: contiguous layout, native little-endian type of the
array) is the file the reference's `write_hdf5` writes, and anything h5py wrote is read back by `H5Dread` with the conversion
to the native type done by the library (any byte order, any layout / filter the library was built with).

Differences from the reference on purpose: errors are Python exceptions (FileNotFoundError / KeyError / FileExistsError)
instead of print + sys.exit, and a dataset is returned with the numpy dtype matching its stored class / size / sign.

The library is found at the first call: `use_library(path)` if given, else `ctypes.util.find_library("hdf5")`, else the usual
prefixes (`/opt/conda/lib`, `/usr/lib/x86_64-linux-gnu/{,hdf5/serial/}`, `/usr/local/lib`).  Not finding one raises ImportError.
The HDF5 C library is not thread-safe in its default build: call from one thread per process (DataLoader workers are processes).
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_hid = ctypes.c_int64          # hid_t of HDF5 >= 1.10
_hsize = ctypes.c_uint64
_LIB = None
_LIB_PATH = None

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0


def use_library(path):
    """Bind THIS libhdf5 (absolute path) instead of searching for one."""
    global _LIB, _LIB_PATH
    _LIB, _LIB_PATH = None, path


def _candidates():
    if _LIB_PATH:
        yield _LIB_PATH
        return
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*",
                "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/usr/lib64/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def _sig(lib, name, res, *args):
    f = getattr(lib, name)
    f.restype, f.argtypes = res, list(args)
    return f


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    errs = []
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
            lib.H5open
        except (OSError, AttributeError) as e:
            errs.append("%s: %s" % (path, e))
            continue
        break
    else:
        raise ImportError("no HDF5 C library found (tried: %s); install libhdf5 or h5py, or call hdf5io.use_library(path)"
                          % ("; ".join(errs) or "nothing on the search path"))
    c_int, c_uint, c_char_p, c_void_p, c_size_t = ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t
    _sig(lib, "H5open", c_int)
    _sig(lib, "H5get_libversion", c_int, ctypes.POINTER(c_uint), ctypes.POINTER(c_uint), ctypes.POINTER(c_uint))
    _sig(lib, "H5Eset_auto2", c_int, _hid, c_void_p, c_void_p)
    _sig(lib, "H5Fopen", _hid, c_char_p, c_uint, _hid)
    _sig(lib, "H5Fcreate", _hid, c_char_p, c_uint, _hid, _hid)
    _sig(lib, "H5Fflush", c_int, _hid, c_int)
    _sig(lib, "H5Fclose", c_int, _hid)
    _sig(lib, "H5Lexists", c_int, _hid, c_char_p, _hid)
    _sig(lib, "H5Ldelete", c_int, _hid, c_char_p, _hid)
    _sig(lib, "H5Oexists_by_name", c_int, _hid, c_char_p, _hid)
    _sig(lib, "H5Dopen2", _hid, _hid, c_char_p, _hid)
    _sig(lib, "H5Dcreate2", _hid, _hid, c_char_p, _hid, _hid, _hid, _hid, _hid)
    _sig(lib, "H5Dclose", c_int, _hid)
    _sig(lib, "H5Dget_space", _hid, _hid)
    _sig(lib, "H5Dget_type", _hid, _hid)
    _sig(lib, "H5Dread", c_int, _hid, _hid, _hid, _hid, _hid, c_void_p)
    _sig(lib, "H5Dwrite", c_int, _hid, _hid, _hid, _hid, _hid, c_void_p)
    _sig(lib, "H5Screate", _hid, c_int)
    _sig(lib, "H5Screate_simple", _hid, c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize))
    _sig(lib, "H5Sget_simple_extent_ndims", c_int, _hid)
    _sig(lib, "H5Sget_simple_extent_dims", c_int, _hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize))
    _sig(lib, "H5Sclose", c_int, _hid)
    _sig(lib, "H5Tget_class", c_int, _hid)
    _sig(lib, "H5Tget_size", c_size_t, _hid)
    _sig(lib, "H5Tget_sign", c_int, _hid)
    _sig(lib, "H5Tclose", c_int, _hid)
    _sig(lib, "H5Pcreate", _hid, _hid)
    _sig(lib, "H5Pset_create_intermediate_group", c_int, _hid, c_uint)
    _sig(lib, "H5Pclose", c_int, _hid)
    if lib.H5open() < 0:
        raise ImportError("H5open() failed in %s" % path)
    maj, mnr, rel = c_uint(), c_uint(), c_uint()
    lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise ImportError("%s is HDF5 %d.%d.%d; this binding needs >= 1.10 (64-bit hid_t)" % (path, maj.value, mnr.value, rel.value))
    lib.H5Eset_auto2(0, None, None)          # errors come back as return codes -> exceptions below, not as stderr dumps
    lib._version = (maj.value, mnr.value, rel.value)
    lib._path = path
    _LIB = lib
    return lib


def library_version():
    """(major, minor, release) of the bound HDF5 library and its path."""
    lib = _lib()
    return lib._version, lib._path


def _glob(name):
    """A library global of type hid_t (the H5T_NATIVE_* / H5P_CLS_* macros of the C headers are these variables)."""
    return _hid.in_dll(_lib(), name).value


_NATIVE = {np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
           np.dtype(np.int8): "H5T_NATIVE_INT8_g", np.dtype(np.int16): "H5T_NATIVE_INT16_g",
           np.dtype(np.int32): "H5T_NATIVE_INT32_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g",
           np.dtype(np.uint8): "H5T_NATIVE_UINT8_g", np.dtype(np.uint16): "H5T_NATIVE_UINT16_g",
           np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g"}


def _b(s):
    return s if isinstance(s, bytes) else os.fsencode(s)


def _has(lib, f, path):
    """`path in f` of h5py: every link along the path exists and resolves."""
    parts = [p for p in path.split("/") if p]
    if not parts:
        return True
    cur = ""
    for p in parts:
        cur += "/" + p
        if lib.H5Lexists(f, _b(cur), H5P_DEFAULT) <= 0:
            return False
    return lib.H5Oexists_by_name(f, _b(cur), H5P_DEFAULT) > 0


def _open(hdf5_name, flags):
    f = _lib().H5Fopen(_b(hdf5_name), flags, H5P_DEFAULT)
    if f < 0:
        raise OSError("cannot open %s as an HDF5 file" % hdf5_name)
    return f


def check_hdf5(hdf5_name, hdf5_path):
    """True when the file exists and holds `hdf5_path` (src/utils/utils.py:18-35)."""
    if not os.path.exists(hdf5_name):
        return False
    lib = _lib()
    f = _open(hdf5_name, H5F_ACC_RDONLY)
    try:
        return _has(lib, f, hdf5_path)
    finally:
        lib.H5Fclose(f)


def _dataset(lib, f, hdf5_name, hdf5_path):
    if not _has(lib, f, hdf5_path):
        raise KeyError("there is no such data in %s: %s" % (hdf5_name, hdf5_path))
    d = lib.H5Dopen2(f, _b(hdf5_path), H5P_DEFAULT)
    if d < 0:
        raise KeyError("%s in %s is not a dataset" % (hdf5_path, hdf5_name))
    return d


def _shape(lib, d):
    sp = lib.H5Dget_space(d)
    try:
        nd = lib.H5Sget_simple_extent_ndims(sp)
        if nd < 0:
            raise OSError("H5Sget_simple_extent_ndims failed")
        dims = (_hsize * max(nd, 1))()
        if nd:
            lib.H5Sget_simple_extent_dims(sp, dims, None)
        return tuple(int(dims[i]) for i in range(nd))
    finally:
        lib.H5Sclose(sp)


def _dtype(lib, d, what):
    t = lib.H5Dget_type(d)
    try:
        cls, size = lib.H5Tget_class(t), int(lib.H5Tget_size(t))
        if cls == H5T_FLOAT and size in (4, 8):
            return np.dtype("f%d" % size)
        if cls == H5T_INTEGER and size in (1, 2, 4, 8):
            return np.dtype(("u%d" if lib.H5Tget_sign(t) == H5T_SGN_NONE else "i%d") % size)
        raise TypeError("%s: only integer and 4 / 8-byte floating datasets are supported (class %d, %d bytes)" % (what, cls, size))
    finally:
        lib.H5Tclose(t)


def read_hdf5(hdf5_name, hdf5_path):
    """The dataset's values as a numpy array of its stored type (src/utils/utils.py:38-60)."""
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError("there is no such hdf5 file: %s" % hdf5_name)
    lib = _lib()
    f = _open(hdf5_name, H5F_ACC_RDONLY)
    try:
        d = _dataset(lib, f, hdf5_name, hdf5_path)
        try:
            dt = _dtype(lib, d, "%s:%s" % (hdf5_name, hdf5_path))
            out = np.empty(_shape(lib, d), dtype=dt)
            if out.size and lib.H5Dread(d, _glob(_NATIVE[dt]), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise OSError("H5Dread failed on %s:%s" % (hdf5_name, hdf5_path))
            return out if out.ndim else out[()]
        finally:
            lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)


def shape_hdf5(hdf5_name, hdf5_path):
    """The dataset's shape (src/utils/utils.py:63-79)."""
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError("there is no such hdf5 file: %s" % hdf5_name)
    lib = _lib()
    f = _open(hdf5_name, H5F_ACC_RDONLY)
    try:
        d = _dataset(lib, f, hdf5_name, hdf5_path)
        try:
            return _shape(lib, d)
        finally:
            lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)


def write_hdf5(hdf5_name, hdf5_path, write_data, is_overwrite=True):
    """Create (or replace) dataset `hdf5_path` with `write_data` (src/utils/utils.py:82-126): the folder is made when missing, an
    existing file is opened read-write and keeps its other datasets, an existing dataset is unlinked and recreated when
    `is_overwrite`, else FileExistsError."""
    data = np.require(np.array(write_data), requirements="C")     # keeps a 0-d array 0-d (scalar dataspace, like h5py)
    if data.dtype == np.bool_:
        data = data.astype(np.int8)              # h5py stores bools as an enum of int8; the recipe never writes one
    if data.dtype not in _NATIVE:
        raise TypeError("write_hdf5: dtype %s is not supported (integers, float32, float64)" % data.dtype)
    folder = os.path.dirname(hdf5_name)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)
    lib = _lib()
    if os.path.exists(hdf5_name):
        f = _open(hdf5_name, H5F_ACC_RDWR)
        if _has(lib, f, hdf5_path):
            if not is_overwrite:
                lib.H5Fclose(f)
                raise FileExistsError("%s already holds %s (is_overwrite=False)" % (hdf5_name, hdf5_path))
            if lib.H5Ldelete(f, _b(hdf5_path), H5P_DEFAULT) < 0:
                lib.H5Fclose(f)
                raise OSError("cannot unlink %s in %s" % (hdf5_path, hdf5_name))
    else:
        f = lib.H5Fcreate(_b(hdf5_name), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        if f < 0:
            raise OSError("cannot create %s" % hdf5_name)
    sp = d = lcpl = -1
    try:
        if data.ndim:
            dims = (_hsize * data.ndim)(*data.shape)
            sp = lib.H5Screate_simple(data.ndim, dims, None)
        else:
            sp = lib.H5Screate(0)                # H5S_SCALAR
        lcpl = lib.H5Pcreate(_glob("H5P_CLS_LINK_CREATE_ID_g"))
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        tid = _glob(_NATIVE[data.dtype])
        d = lib.H5Dcreate2(f, _b(hdf5_path), tid, sp, lcpl, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            raise OSError("cannot create dataset %s in %s" % (hdf5_path, hdf5_name))
        if data.size and lib.H5Dwrite(d, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.ctypes.data_as(ctypes.c_void_p)) < 0:
            raise OSError("H5Dwrite failed on %s:%s" % (hdf5_name, hdf5_path))
    finally:
        if d >= 0:
            lib.H5Dclose(d)
        if lcpl >= 0:
            lib.H5Pclose(lcpl)
        if sp >= 0:
            lib.H5Sclose(sp)
        lib.H5Fflush(f, 1)                       # H5F_SCOPE_GLOBAL
        lib.H5Fclose(f)
    return 1
