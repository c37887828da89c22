"""Test infrastructure: a minimal ctypes walker over an HDF5 file (no h5py / h5dump in the image).

`describe(path)` returns {object path: {"kind", "type", "shape", "data", "attrs": {name: {"type", "shape", "data"}}}} for
every group and dataset, with `type` = (class, size, sign, order, strpad, cset) and `data` = the raw bytes in the file's own
type.  Two files are the same to HDF5's data model iff their descriptions are equal.  It binds the same libhdf5 the product
library dlopens (fdes_amd/csrc/emd.cpp); used to pin the EMD writer to the file the reference wrote
(ExampleSpecimens/Au_cubeoctahedron_emd/Auparticle.emd, writer src/rwHdf5.cu:1085-1944)."""
import ctypes as C
import os

_NAMES = ["libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so", "libhdf5_serial.so.103",
          "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"]
hid_t = C.c_int64
hsize_t = C.c_uint64
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    names = ([os.environ["FDES_HDF5_LIB"]] if os.environ.get("FDES_HDF5_LIB") else []) + _NAMES
    for n in names:
        try:
            lib = C.CDLL(n)
            break
        except OSError:
            continue
    else:
        raise OSError("libhdf5 not loadable")
    lib.H5open()

    def fn(name, res, *args):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, list(args)
        return f
    fn("H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
    fn("H5Fclose", C.c_int, hid_t)
    fn("H5Oopen", hid_t, hid_t, C.c_char_p, hid_t)
    fn("H5Oclose", C.c_int, hid_t)
    fn("H5Iget_type", C.c_int, hid_t)
    fn("H5Dget_type", hid_t, hid_t)
    fn("H5Dget_space", hid_t, hid_t)
    fn("H5Dread", C.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    fn("H5Aopen_by_idx", hid_t, hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, hid_t, hid_t)
    fn("H5Aget_name", C.c_ssize_t, hid_t, C.c_size_t, C.c_char_p)
    fn("H5Aget_type", hid_t, hid_t)
    fn("H5Aget_space", hid_t, hid_t)
    fn("H5Aread", C.c_int, hid_t, hid_t, C.c_void_p)
    fn("H5Aclose", C.c_int, hid_t)
    fn("H5Tget_class", C.c_int, hid_t)
    fn("H5Tget_size", C.c_size_t, hid_t)
    fn("H5Tget_sign", C.c_int, hid_t)
    fn("H5Tget_order", C.c_int, hid_t)
    fn("H5Tget_strpad", C.c_int, hid_t)
    fn("H5Tget_cset", C.c_int, hid_t)
    fn("H5Tclose", C.c_int, hid_t)
    fn("H5Sget_simple_extent_type", C.c_int, hid_t)
    fn("H5Sget_simple_extent_ndims", C.c_int, hid_t)
    fn("H5Sget_simple_extent_dims", C.c_int, hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t))
    fn("H5Sget_simple_extent_npoints", C.c_int64, hid_t)
    fn("H5Sclose", C.c_int, hid_t)
    fn("H5Eset_auto2", C.c_int, hid_t, C.c_void_p, C.c_void_p)
    lib.H5Eset_auto2(0, None, None)
    lib._iterate = getattr(lib, "H5Literate", None) or getattr(lib, "H5Literate1")
    lib._iterate.restype = C.c_int
    _lib = lib
    return lib


_CB = C.CFUNCTYPE(C.c_int, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)
_CLASS = {0: "int", 1: "float", 3: "string"}


def _type(lib, t):
    cls = lib.H5Tget_class(t)
    size = lib.H5Tget_size(t)
    if cls == 3:
        return ("string", size, None, None, lib.H5Tget_strpad(t), lib.H5Tget_cset(t))
    sign = lib.H5Tget_sign(t) if cls == 0 else None
    return (_CLASS.get(cls, cls), size, sign, lib.H5Tget_order(t), None, None)


def _space(lib, sp):
    kind = lib.H5Sget_simple_extent_type(sp)  # 0 scalar, 1 simple
    nd = lib.H5Sget_simple_extent_ndims(sp)
    dims = (hsize_t * max(nd, 1))()
    if nd > 0:
        lib.H5Sget_simple_extent_dims(sp, dims, None)
    shape = "scalar" if kind == 0 else tuple(int(dims[i]) for i in range(nd))
    return shape, int(lib.H5Sget_simple_extent_npoints(sp))


def _attrs(lib, obj):
    out = {}
    i = 0
    while True:
        a = lib.H5Aopen_by_idx(obj, b".", 0, 0, i, 0, 0)  # H5_INDEX_NAME, H5_ITER_INC
        if a < 0:
            break
        n = lib.H5Aget_name(a, 0, None)
        buf = C.create_string_buffer(n + 1)
        lib.H5Aget_name(a, n + 1, buf)
        t = lib.H5Aget_type(a)
        sp = lib.H5Aget_space(a)
        ty = _type(lib, t)
        shape, npts = _space(lib, sp)
        raw = C.create_string_buffer(max(1, npts * ty[1]))
        assert lib.H5Aread(a, t, raw) >= 0
        out[buf.value.decode()] = {"type": ty, "shape": shape, "data": raw.raw[:npts * ty[1]]}
        lib.H5Sclose(sp)
        lib.H5Tclose(t)
        lib.H5Aclose(a)
        i += 1
    return out


def describe(path):
    lib = _load()
    f = lib.H5Fopen(str(path).encode(), 0, 0)
    if f < 0:
        raise OSError(f"cannot open {path}")
    out = {}

    def visit(loc, prefix):
        names = []

        def cb(_g, name, _info, _data):
            names.append(name)
            return 0
        idx = hsize_t(0)
        lib._iterate(hid_t(loc), C.c_int(0), C.c_int(0), C.byref(idx), _CB(cb), None)
        for name in names:
            o = lib.H5Oopen(loc, name, 0)
            assert o >= 0, name
            full = prefix + "/" + name.decode()
            kind = lib.H5Iget_type(o)  # H5I_GROUP = 2, H5I_DATASET = 5
            entry = {"kind": {2: "group", 5: "dataset"}.get(kind, kind), "attrs": _attrs(lib, o)}
            if kind == 5:
                t = lib.H5Dget_type(o)
                sp = lib.H5Dget_space(o)
                ty = _type(lib, t)
                shape, npts = _space(lib, sp)
                raw = C.create_string_buffer(max(1, npts * ty[1]))
                assert lib.H5Dread(o, t, 0, 0, 0, raw) >= 0
                entry.update(type=ty, shape=shape, data=raw.raw[:npts * ty[1]])
                lib.H5Sclose(sp)
                lib.H5Tclose(t)
            out[full] = entry
            if kind == 2:
                visit(o, full)
            lib.H5Oclose(o)
    root = lib.H5Oopen(f, b"/", 0)
    out["/"] = {"kind": "group", "attrs": _attrs(lib, root)}
    visit(root, "")
    lib.H5Oclose(root)
    lib.H5Fclose(f)
    return out


def diff(a, b):
    """Human-readable list of differences between two descriptions (empty = identical)."""
    msgs = []
    for k in sorted(set(a) | set(b)):
        if k not in a or k not in b:
            msgs.append(f"{k}: only in {'first' if k in a else 'second'}")
            continue
        ea, eb = a[k], b[k]
        for f in ("kind", "type", "shape", "data"):
            if ea.get(f) != eb.get(f):
                va, vb = ea.get(f), eb.get(f)
                if f == "data":
                    va, vb = va[:32], vb[:32]
                msgs.append(f"{k}: {f} {va!r} != {vb!r}")
        for n in sorted(set(ea["attrs"]) | set(eb["attrs"])):
            if n not in ea["attrs"] or n not in eb["attrs"]:
                msgs.append(f"{k}@{n}: only in {'first' if n in ea['attrs'] else 'second'}")
            elif ea["attrs"][n] != eb["attrs"][n]:
                msgs.append(f"{k}@{n}: {ea['attrs'][n]!r} != {eb['attrs'][n]!r}")
    return msgs
