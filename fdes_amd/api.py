"""Host-side Python interface over the C-ABI (include/fdes_abi.h).

Mirrors the reference's call surface for the forward path: `read_cnf` = getParams
(src/paramStructure.cu:600-635), `Engine.build_measurements` = buildMeasurements
(src/crystalMaker.cu:227-424), `run_file` = the exported FDES(...) (src/FDESExport.cu:59-178) that
Python/pyFDES.py:100 calls.  Everything executes in the HIP library; there is no Python or CPU
implementation behind these calls.
"""
import ctypes as C
import weakref

import numpy as np

from . import abi
from .abi import HostAtoms, HostParams, fptr

ERRORS = {0: "ok", -1: "invalid argument", -2: "I/O error", -3: "GPU runtime error", -4: "out of memory",
          -5: "unsupported"}


class FdesError(RuntimeError):
    def __init__(self, code, detail=""):
        super().__init__(f"fdes error {code} ({ERRORS.get(code, '?')}) {detail}")
        self.code = code


def _chk(rc, detail=""):
    if rc != 0:
        raise FdesError(rc, detail)


def gpu_available():
    return bool(abi.load_library().fdes_gpu_available())


def consistent(hp):
    """consitentParams (src/paramStructure.cu:637-673) on a HostParams, in place."""
    _chk(abi.load_library().fdes_params_consistent(hp.ptr))
    return hp


def sub_sliced(hp):
    q = hp.copy()
    ratio = abi.load_library().fdes_params_sub_slices(q.ptr)
    if ratio < 0:
        raise FdesError(ratio)
    return q, ratio


def emd_available():
    return bool(abi.load_library().fdes_emd_available())


def write_emd(path, hp, atoms=None, image=None, potential=None, exitwave=None, print_level=0):
    """writeHdf5 (src/rwHdf5.cu:27-1083); image [n3,n2,n1], potential [m3,m2,m1,2], exitwave [n3,m2,m1,2] float32."""
    f = lambda a: fptr(np.ascontiguousarray(a, np.float32)) if a is not None else None
    _chk(abi.load_library().fdes_write_emd(str(path).encode(), hp.ptr, atoms.ptr if atoms is not None else None, f(image),
                                           f(potential), f(exitwave), print_level), str(path))


def read_emd(path, skip_atoms=False, capacity=1000):
    """readHdf5 (src/rwHdf5.cu:1946-2570): returns (HostParams, HostAtoms or None), parameters made consistent."""
    return _read_file(path, "fdes_read_emd", 2 if skip_atoms else 0, skip_atoms, capacity)


def build_measurements_multi(devices, hp, atoms, want_potential=False, want_exitwave=False):
    """fdes_build_measurements_multi: one host thread per entry of `devices`.  Returns image[n3, n2, n1], or the same
    dict as Engine.build_measurements when a potential / exit-wave output is asked for."""
    lib = abi.load_library()
    dev = (C.c_int * len(devices))(*devices)
    c = hp.c
    img = np.zeros((c.n3, c.n2, c.n1), np.float32)
    pot = np.zeros((c.m3, c.m2, c.m1, 2), np.float32) if want_potential else None
    ew = np.zeros((c.n3, c.m2, c.m1, 2), np.float32) if want_exitwave else None
    _chk(lib.fdes_build_measurements_multi(len(devices), dev, hp.ptr, atoms.ptr, fptr(img), fptr(pot) if want_potential else None,
                                           fptr(ew) if want_exitwave else None))
    if not (want_potential or want_exitwave):
        return img
    return {"image": img, "potential": pot, "exitwave": ew}


def read_qsc(path, skip_atoms=False, capacity=1000):
    """readQsc (src/rwQsc.cu:8-1101): QSTEM .qsc + the .cfg cell it names -> (HostParams, HostAtoms or None)."""
    return _read_file(path, "fdes_read_qsc", 2 if skip_atoms else 0, skip_atoms, capacity)


def read_cnf(path, bug_compatible=True, skip_atoms=False, capacity=1000):
    """getParams: returns (HostParams, HostAtoms or None). Parameters are made consistent."""
    return _read_file(path, "fdes_read_cnf", (1 if bug_compatible else 0) | (2 if skip_atoms else 0), skip_atoms, capacity)


def _read_file(path, fn, flags, skip_atoms, capacity):
    lib = abi.load_library()
    p = abi.Params()
    _chk(lib.fdes_params_init(C.byref(p), capacity))
    a = abi.Atoms()
    try:
        _chk(getattr(lib, fn)(str(path).encode(), C.byref(p), C.byref(a), flags), str(path))
        n3 = p.n3
        hp = HostParams(n3)
        ts = np.ctypeslib.as_array(p.tiltspec, (2 * n3,)).copy()
        tb = np.ctypeslib.as_array(p.tiltbeam, (2 * n3,)).copy()
        df = np.ctypeslib.as_array(p.defoci, (n3,)).copy()
        C.memmove(C.byref(hp.c), C.byref(p), C.sizeof(abi.Params))
        hp.c.cap = n3  # re-point the three arrays at the numpy-owned storage (p's are freed below)
        hp.c.tiltspec = hp.tiltspec.ctypes.data_as(C.POINTER(C.c_float))
        hp.c.tiltbeam = hp.tiltbeam.ctypes.data_as(C.POINTER(C.c_float))
        hp.c.defoci = hp.defoci.ctypes.data_as(C.POINTER(C.c_float))
        hp.tiltspec[:] = ts
        hp.tiltbeam[:] = tb
        hp.defoci[:] = df
        atoms = None
        if not skip_atoms:
            n = a.nAt
            atoms = HostAtoms(np.ctypeslib.as_array(a.Z, (n,)).copy() if n else np.zeros(0, np.int32),
                              np.ctypeslib.as_array(a.xyz, (3 * n,)).copy() if n else np.zeros((0, 3), np.float32),
                              np.ctypeslib.as_array(a.dwf, (n,)).copy() if n else np.zeros(0, np.float32),
                              np.ctypeslib.as_array(a.occ, (n,)).copy() if n else np.zeros(0, np.float32))
    finally:
        lib.fdes_atoms_release(C.byref(a))
        lib.fdes_params_release(C.byref(p))
    consistent(hp)
    return hp, atoms


def write_cnf(path, hp, atoms=None):
    _chk(abi.load_library().fdes_write_cnf(str(path).encode(), hp.ptr, atoms.ptr if atoms is not None else None))


def run_file(input_name, image_name="Measurements.bin", emd_name="results.emd", atoms=None, gpu_index=0,
             print_level=0, out=None):
    """The exported FDES(...) call of Python/pyFDES.py:100 with an int status. `atoms`: HostAtoms or
    float32[n,6] = [Z,x,y,z,DWF,occ]; `out`: float32 buffer of n1*n2*n3."""
    lib = abi.load_library()
    arr = None
    n = 0
    if atoms is not None:
        arr = np.ascontiguousarray(atoms.as_array6() if isinstance(atoms, HostAtoms) else atoms, np.float32)
        n = arr.shape[0]
    rc = lib.fdes_run_file(gpu_index, print_level, str(input_name).encode(),
                           str(image_name).encode() if image_name else None,
                           str(emd_name).encode() if emd_name else None,
                           fptr(arr) if arr is not None else None, n, fptr(out) if out is not None else None)
    _chk(rc, str(input_name))


class Engine:
    """One GPU context (fdes_ctx)."""

    def __init__(self, gpu_index=0, **options):
        self.lib = abi.load_library()
        self.h = C.c_void_p()
        self._plans = weakref.WeakSet()
        _chk(self.lib.fdes_create(C.byref(self.h), gpu_index), "fdes_create: no usable GPU")
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, key, value):
        _chk(self.lib.fdes_set_option(self.h, key.encode(), int(value)), key)

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id made by one rank (fdes_comm_unique_id); hand it to the other ranks."""
        buf = C.create_string_buffer(128)
        _chk(abi.load_library().fdes_comm_unique_id(buf), "fdes_comm_unique_id (librccl.so)")
        return buf.raw

    def comm_create(self, nranks, rank, uid):
        """This context's rank in an RCCL communicator (fdes_comm_create; blocks until all ranks joined). Returns the handle."""
        h = C.c_void_p()
        _chk(self.lib.fdes_comm_create(self.h, int(nranks), int(rank), uid, C.byref(h)), self.err())
        return h

    def comm_destroy(self, comm):
        _chk(self.lib.fdes_comm_destroy(comm), "fdes_comm_destroy")

    def set_progress(self, fn, min_interval_ms=200):
        """fn(done, total) in slice-propagations, called between configurations of build_measurements; None removes it."""
        self._progress_cb = abi.PROGRESS_FN(lambda _u, d, t: fn(int(d), int(t))) if fn else None  # keep the thunk alive
        _chk(self.lib.fdes_set_progress(self.h, C.cast(self._progress_cb, C.c_void_p) if fn else None, None,
                                        int(min_interval_ms)))

    def err(self):
        return (self.lib.fdes_last_error(self.h) or b"").decode()

    def close(self):
        if self.h:
            for pl in list(self._plans):  # plans hold a pointer to this context
                pl.close()
            self.lib.fdes_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_measurements(self, hp, atoms, want_potential=False, want_exitwave=False):
        """hp: consistent parameters BEFORE sub-slicing."""
        c = hp.c
        img = np.zeros((c.n3, c.n2, c.n1), np.float32)
        pot = np.zeros((c.m3, c.m2, c.m1, 2), np.float32) if want_potential else None
        ew = np.zeros((c.n3, c.m2, c.m1, 2), np.float32) if want_exitwave else None
        rc = self.lib.fdes_build_measurements(self.h, hp.ptr, atoms.ptr, fptr(img), fptr(pot) if want_potential else None,
                                              fptr(ew) if want_exitwave else None)
        _chk(rc, self.err())
        return {"image": img, "potential": pot, "exitwave": ew}

    def fft2(self, f, inverse=False, backend=0):
        """Unnormalised 2-D FFT of a complex array [m2, m1] through the engine's FFT back-end."""
        m2, m1 = f.shape
        buf = np.empty((m2, m1, 2), np.float32)
        buf[..., 0] = f.real
        buf[..., 1] = f.imag
        rc = self.lib.fdes_fft2d_host(self.h, fptr(buf), m1, m2, int(inverse), backend)
        if rc < 0:
            raise FdesError(rc, self.err())
        return buf[..., 0] + 1j * buf[..., 1], rc

    def bench_pass(self, n, pre, mid, post, store_t, iters=50, streams=1):
        us = C.c_double()
        _chk(self.lib.fdes_bench_pass(self.h, n, pre, mid, post, int(store_t), iters, streams, C.byref(us)), self.err())
        return us.value

    def plan(self, hp, atoms):
        return Plan(self, hp, atoms)


class Plan:
    """Device-resident simulation state (fdes_plan)."""

    def __init__(self, eng, hp, atoms):
        self.eng = eng
        self.lib = eng.lib
        self.hp = hp
        self.atoms = atoms
        self.h = C.c_void_p()
        _chk(self.lib.fdes_plan_create(eng.h, hp.ptr, atoms.ptr, C.byref(self.h)), eng.err())
        eng._plans.add(self)
        q, _ = sub_sliced(hp)
        self.m1, self.m2, self.m3 = q.c.m1, q.c.m2, q.c.m3
        self.n1, self.n2, self.n3 = q.c.n1, q.c.n2, q.c.n3

    def _c(self, rc):
        _chk(rc, self.eng.err())

    def close(self):
        if self.h and self.eng.h:
            self.lib.fdes_plan_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin_measurement(self, k):
        self._c(self.lib.fdes_plan_begin_measurement(self.h, k))

    def run_config(self, k, j, weight):
        self._c(self.lib.fdes_plan_run_config(self.h, k, j, weight))

    def end_measurement(self, k):
        self._c(self.lib.fdes_plan_end_measurement(self.h, k))

    def run_measurements(self, ks):
        """Complete measurements `ks` (every configuration, detector chain); a series with one configuration per
        measurement runs in gangs of measurements."""
        ks = [int(k) for k in ks]
        arr = (C.c_int * max(len(ks), 1))(*ks)
        self._c(self.lib.fdes_plan_run_measurements(self.h, arr, len(ks)))

    def sync(self):
        self._c(self.lib.fdes_plan_sync(self.h))

    def intensity_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._c(self.lib.fdes_plan_intensity_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def copy_intensity(self, dev_ptr, to_plan):
        self._c(self.lib.fdes_plan_copy_intensity(self.h, C.c_void_p(dev_ptr), int(to_plan)))

    def copy_intensity_real(self, dev_ptr, to_plan):
        """float[m1*m2] view of the running intensity sum (its imaginary part is identically zero)."""
        self._c(self.lib.fdes_plan_copy_intensity_real(self.h, C.c_void_p(dev_ptr), int(to_plan)))

    def images_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._c(self.lib.fdes_plan_images_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_images(self):
        img = np.zeros((self.n3, self.n2, self.n1), np.float32)
        self._c(self.lib.fdes_plan_get_images(self.h, fptr(img)))
        return img

    def slice_loop_ms(self):
        t, n = C.c_double(), C.c_int64()
        self._c(self.lib.fdes_plan_slice_loop_ms(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value

    def probe_ms(self):
        t, n = C.c_double(), C.c_int64()
        self._c(self.lib.fdes_plan_probe_ms(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value

    def accumulate_from(self, other):
        """I (and the exit-wave sum) += other's, device to device (fdes_plan_accumulate_from)."""
        self._c(self.lib.fdes_plan_accumulate_from(self.h, other.h))

    def reduce_intensity(self, comm, root):
        """Sum of the ranks' running intensity sums onto `root` through RCCL (fdes_plan_reduce_intensity); every rank calls it."""
        self._c(self.lib.fdes_plan_reduce_intensity(self.h, comm, int(root)))

    def reduce_intensity_span(self, comm, root, lo, hi):
        """The same for a measurement that spans the ranks lo .. hi only (fdes_plan_reduce_intensity_span); those ranks call it."""
        self._c(self.lib.fdes_plan_reduce_intensity_span(self.h, comm, int(root), int(lo), int(hi)))

    def want_exitwave(self, on=True):
        self._c(self.lib.fdes_plan_want_exitwave(self.h, int(on)))

    def get_exitwave(self):
        out = np.zeros((self.m2, self.m1, 2), np.float32)
        self._c(self.lib.fdes_plan_get_exitwave(self.h, fptr(out)))
        return out[..., 0] + 1j * out[..., 1]

    def fft_backend(self):
        return int(self.lib.fdes_plan_fft_backend(self.h))

    def jit_kernels(self):
        """Grid axes whose row passes run kernels compiled for that length at plan creation (0 ... 2)."""
        return int(self.lib.fdes_plan_jit_kernels(self.h))

    def lanes(self):
        return int(self.lib.fdes_plan_lanes(self.h))

    def gang(self):
        return int(self.lib.fdes_plan_gang(self.h))

    def slices_done(self):
        return int(self.lib.fdes_plan_slices_done(self.h))

    def empty_queries(self):
        return int(self.lib.fdes_plan_empty_queries(self.h))

    def tap_coords(self, k, j):
        out = np.zeros((self.atoms.n, 3), np.float32)
        self._c(self.lib.fdes_plan_tap_coords(self.h, k, j, fptr(out)))
        return out

    def tap_potential(self, k, j, s):
        out = np.zeros((self.m2, self.m1, 2), np.float32)
        self._c(self.lib.fdes_plan_tap_potential(self.h, k, j, s, fptr(out)))
        return out[..., 0] + 1j * out[..., 1]

    def tap_wave(self, k, j, nslices=None):
        out = np.zeros((self.m2, self.m1, 2), np.float32)
        self._c(self.lib.fdes_plan_tap_wave(self.h, k, j, self.m3 if nslices is None else nslices, fptr(out)))
        return out[..., 0] + 1j * out[..., 1]

    def tap_propagator(self):
        out = np.zeros((self.m2, self.m1, 2), np.float32)
        self._c(self.lib.fdes_plan_tap_propagator(self.h, fptr(out)))
        return out[..., 0] + 1j * out[..., 1]

    def propagate_dev(self, psi_ptr, t_ptr, batch=1, t_per_wave=False):
        self._c(self.lib.fdes_plan_propagate_dev(self.h, C.c_void_p(psi_ptr), C.c_void_p(t_ptr), batch, int(t_per_wave)))
