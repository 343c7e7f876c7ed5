"""ctypes binding of libjurassic_hip.so (include/jurassic_hip.h).

The library is the product; this module only marshals arguments.  It raises if
the shared object is missing -- there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np
from . import abi

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("JURASSIC_HIP_SO",       # override: A/B builds
                    os.path.join(PKG, "libjurassic_hip%s.so" % os.environ.get("JUR_SUFFIX", "")))
_lib = None
dp = C.POINTER(C.c_double)


class JurassicError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise JurassicError(f"{SO} not built: run `make -C jurassic-gpu_amd/csrc` "
                                "or __graft_entry__.build(); no CPU fallback exists")
        L = C.CDLL(SO)
        L.jur_last_error.restype = C.c_char_p
        L.jur_fov_read_shape.argtypes = [C.c_char_p, C.POINTER(C.c_int), dp, dp]
        L.jur_fov_apply.argtypes = [C.c_int, C.c_long, dp, dp, dp, dp, C.c_long, C.c_int, dp, dp]
        L.formod_fov.argtypes = [C.c_void_p, C.c_void_p]
        L.formod_fov.restype = None
        L.jur_tables_new.restype = C.c_void_p
        L.jur_tables_new.argtypes = [C.c_int, C.c_int]
        L.jur_tables_free.argtypes = [C.c_void_p]
        L.jur_tables_feed_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, dp, dp, dp, dp]
        L.jur_tables_read_ascii.argtypes = [C.c_void_p, C.c_void_p]
        L.jur_tables_set_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp]
        L.jur_tables_read_filters.argtypes = [C.c_void_p, C.c_void_p]
        L.jur_tables_save.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.jur_tables_load.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
        L.jur_tables_cache_filename.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.jur_tables_checksum.restype = C.c_ulonglong
        L.jur_tables_checksum.argtypes = [C.c_void_p]
        L.jur_tables_entries.restype = C.c_long
        L.jur_tables_entries.argtypes = [C.c_void_p]
        L.jur_model_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int]
        L.jur_model_create_from_files.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.jur_model_destroy.argtypes = [C.c_void_p]
        L.jur_model_set_atm.argtypes = [C.c_void_p, C.c_void_p]
        L.jur_formod_host.argtypes = [C.c_void_p, C.c_long, C.POINTER(dp), dp, dp, C.POINTER(dp), C.POINTER(C.c_int)]
        L.jur_curtis_godson_host.argtypes = [C.c_void_p, C.c_long, C.POINTER(dp), dp, dp, dp, C.POINTER(dp), C.POINTER(C.c_int)]
        L.jur_fov_apply_device.argtypes = [C.c_void_p, C.c_long] + [C.c_void_p] * 4 + [C.c_int, dp, dp, C.c_void_p]
        L.jur_intpol_atm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.intpol_atm.argtypes = [C.c_void_p] * 3
        L.intpol_atm.restype = None
        L.jur_host_alloc.restype = C.c_void_p
        L.jur_host_alloc.argtypes = [C.c_size_t]
        L.jur_host_free.argtypes = [C.c_void_p]
        L.jur_formod_device.argtypes = [C.c_void_p, C.c_long] + [C.c_void_p] * 7
        L.jur_model_reserve.argtypes = [C.c_void_p, C.c_long]
        L.jur_model_workspace_bytes.restype = C.c_long
        L.jur_model_workspace_bytes.argtypes = [C.c_void_p]
        L.jur_model_table_bytes.restype = C.c_long
        L.jur_model_table_bytes.argtypes = [C.c_void_p]
        L.jur_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.jur_model_chunk_rays.argtypes = [C.c_void_p]
        L.jur_model_last_launches.restype = C.c_long
        L.jur_model_last_launches.argtypes = [C.c_void_p]
        L.jur_model_set_compact_workspace.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_set_chunk_rays.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_set_sort_rays.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_set_trace_multiple.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_set_workspace_budget.argtypes = [C.c_void_p, C.c_long]
        L.jur_model_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_last_kernel_ms.argtypes = [C.c_void_p, dp, C.POINTER(C.c_long)]
        L.jur_model_last_pencil_ms.argtypes = [C.c_void_p, dp, C.POINTER(C.c_long)]
        L.jur_model_set_pencil.argtypes = [C.c_void_p, C.c_long, C.c_int]
        L.jur_model_set_arithmetic.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_arithmetic.argtypes = [C.c_void_p]
        L.jur_model_set_ega_group.argtypes = [C.c_void_p, C.c_int]
        L.jur_model_ega_group.argtypes = [C.c_void_p]
        L.jur_multi_balance.argtypes = [C.c_void_p, C.c_long, C.POINTER(dp), C.c_int, C.POINTER(C.c_long)]
        L.jur_estimate_los_points.argtypes = [C.c_double] * 4 + [C.c_long, C.POINTER(dp), dp]
        L.jur_balance_rays.argtypes = [C.c_double] * 4 + [C.c_long, C.POINTER(dp), C.c_int, C.POINTER(C.c_long)]
        L.jur_models_set_atm.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        L.jur_formod_host_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_long, C.POINTER(dp), dp, dp, C.POINTER(dp), C.POINTER(C.c_int)]
        L.jur_formod_device_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_long, C.POINTER(C.c_long)] + [C.c_void_p] * 7
        L.jur_tune_combine.argtypes = [C.c_int, C.c_int, C.c_long]
        L.jur_tune_combine.restype = None
        L.jur_tune_trace.argtypes = [C.c_int]
        L.jur_tune_trace.restype = None
        L.jur_state_size.restype = C.c_size_t
        L.jur_state_size.argtypes = [C.c_void_p, C.c_void_p]
        L.jur_measurement_size.restype = C.c_size_t
        L.jur_measurement_size.argtypes = [C.c_void_p, C.c_void_p]
        L.jur_kernel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, dp, C.c_size_t, C.c_size_t]
        L.jur_abi_sizes.argtypes = [C.POINTER(C.c_size_t)]
        L.jur_kat_ega_eps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, dp, dp, dp, dp, C.c_int, C.c_int, dp]
        L.jur_kat_continua.argtypes = [C.c_void_p, C.c_int, C.c_long] + [dp] * 6
        L.jur_kat_update.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int] + [dp] * 6
        for name in ("formod", "formod_GPU"):
            getattr(L, name).argtypes = [C.c_void_p] * 3
            getattr(L, name).restype = None
        L.kernel.argtypes = [C.c_void_p] * 4
        L.kernel.restype = None
        L.formod_pencil.argtypes = [C.c_void_p] * 3 + [C.c_int]
        L.formod_pencil.restype = None
        _lib = L
    return _lib


def _chk(rc):
    if rc < 0:
        raise JurassicError(f"jurassic_hip error {rc}: {lib().jur_last_error().decode()}")
    return rc


def _p(a):
    return a.ctypes.data_as(dp)


class Tables:
    def __init__(self, ng, nd):
        self.h = lib().jur_tables_new(ng, nd)
        if not self.h:
            raise JurassicError(lib().jur_last_error().decode())
        self.ng, self.nd = ng, nd

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().jur_tables_free(self.h)
            except TypeError:          # interpreter shutdown
                pass
            self.h = None

    def feed_rows(self, ig, id_, rows):
        r = np.ascontiguousarray(rows, dtype=np.float64)
        cols = [np.ascontiguousarray(r[:, k]) for k in range(4)]
        _chk(lib().jur_tables_feed_rows(self.h, ig, id_, len(r), *[_p(c) for c in cols]))

    def read_ascii(self, ctl):
        return _chk(lib().jur_tables_read_ascii(self.h, C.byref(ctl)))

    def set_filter(self, id_, nu, f):
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        f = np.ascontiguousarray(f, dtype=np.float64)
        _chk(lib().jur_tables_set_filter(self.h, id_, len(nu), _p(nu), _p(f)))

    def read_filters(self, ctl):
        _chk(lib().jur_tables_read_filters(self.h, C.byref(ctl)))

    def entries(self):
        return lib().jur_tables_entries(self.h)

    def checksum(self):
        return lib().jur_tables_checksum(self.h)

    def save(self, ctl, path):
        _chk(lib().jur_tables_save(self.h, C.byref(ctl), os.fsencode(path)))

    @classmethod
    def load(cls, ctl, path):
        h = C.c_void_p()
        _chk(lib().jur_tables_load(C.byref(h), C.byref(ctl), os.fsencode(path)))
        self = cls.__new__(cls)
        self.h, self.ng, self.nd = h, ctl.ng, ctl.nd
        return self


def _free_pinned(p):
    try:
        lib().jur_host_free(p)
    except TypeError:                  # interpreter shutdown: the module globals are gone already
        pass


class HostBuffers:
    """The arrays of one jur_formod_host call, allocated once and reused: geom (7, nr), rad/tau (nr, nd),
    tp (3, nr), np (nr,) -- in pinned host memory (jur_host_alloc) or as ordinary numpy arrays.

    A pinned block lives as long as ANY numpy view of it: the block is freed by a finalizer on the buffer object
    the arrays are views of, not by close() -- `rad = b.rad; b.close()` leaves `rad` valid."""

    def __init__(self, nr, nd, pinned=True):
        self.nr, self.nd, self.pinned = nr, nd, pinned
        self.geom = self._alloc((7, nr), np.float64)
        self.rad = self._alloc((nr, nd), np.float64)
        self.tau = self._alloc((nr, nd), np.float64)
        self.tp = self._alloc((3, nr), np.float64)
        self.np = self._alloc((nr,), np.int32)

    def _alloc(self, shape, dtype):
        if not self.pinned:
            return np.zeros(shape, dtype=dtype)
        import weakref
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = lib().jur_host_alloc(n)
        if not p:
            raise JurassicError(lib().jur_last_error().decode())
        base = (C.c_char * n).from_address(p)
        weakref.finalize(base, _free_pinned, p)          # runs when the last array that views `base` is gone
        a = np.frombuffer(base, dtype=dtype).reshape(shape)
        a[...] = 0
        return a

    def set_geometry(self, geom):
        self.geom[...] = np.asarray(geom, dtype=np.float64).T

    def close(self):
        """Drop this object's references; pinned blocks go when no array views them any more."""
        self.geom = self.rad = self.tau = self.tp = self.np = None


class Model:
    """Control block + tables resident on one GPU."""

    def __init__(self, ctl, tables=None, device=0):
        h = C.c_void_p()
        if tables is None:
            _chk(lib().jur_model_create_from_files(C.byref(h), C.byref(ctl), device))
        else:
            _chk(lib().jur_model_create(C.byref(h), C.byref(ctl), tables.h, device))
        self.h = h
        self.nd = ctl.nd
        self.ng = ctl.ng

    def close(self):
        if getattr(self, "h", None):
            try:
                lib().jur_model_destroy(self.h)
            except TypeError:          # interpreter shutdown: the module globals are gone already
                pass
            self.h = None

    __del__ = close

    def set_atm(self, atm):
        _chk(lib().jur_model_set_atm(self.h, C.byref(atm)))

    def set_chunk_rays(self, n):
        _chk(lib().jur_model_set_chunk_rays(self.h, n))

    def set_sort_rays(self, on):
        _chk(lib().jur_model_set_sort_rays(self.h, int(on)))

    def reserve(self, nr):
        _chk(lib().jur_model_reserve(self.h, nr))

    def set_pencil(self, max_rays, rays_per_group=0):
        """Calls of up to max_rays rays run as one fused kernel (0: never)."""
        _chk(lib().jur_model_set_pencil(self.h, max_rays, rays_per_group))

    def set_arithmetic(self, mode):
        """ARITH_FAST (default) or ARITH_EXACT: the look-up's arithmetic on strictly increasing tables."""
        _chk(lib().jur_model_set_arithmetic(self.h, int(mode)))

    def set_ega_group(self, nch):
        """nch in 2..4: the channel-group look-up kernel (round-4 experiment); < 2: one pair per workgroup (default)."""
        _chk(lib().jur_model_set_ega_group(self.h, int(nch)))
        return lib().jur_model_ega_group(self.h)       # channels per lane the next call will walk (0: default kernel)

    def set_trace_multiple(self, mult):
        _chk(lib().jur_model_set_trace_multiple(self.h, mult))

    def set_workspace_budget(self, nbytes):
        _chk(lib().jur_model_set_workspace_budget(self.h, nbytes))

    def formod_host(self, geom, rad_in=None):
        """geom: (nr, 7).  -> dict(rad, tau, tp (nr,3), np)."""
        g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
        nr, nd = g.shape[1], self.nd
        rad = np.zeros((nr, nd)) if rad_in is None else np.ascontiguousarray(rad_in, dtype=np.float64).copy()
        tau = np.zeros((nr, nd))
        tp = np.zeros((3, nr))
        npts = np.zeros(nr, dtype=np.int32)
        garr = (dp * 7)(*[_p(g[k]) for k in range(7)])
        tarr = (dp * 3)(*[_p(tp[k]) for k in range(3)])
        _chk(lib().jur_formod_host(self.h, nr, garr, _p(rad), _p(tau), tarr, npts.ctypes.data_as(C.POINTER(C.c_int))))
        return dict(rad=rad, tau=tau, tp=np.ascontiguousarray(tp.T), np=npts)

    def host_buffers(self, nr, pinned=True):
        return HostBuffers(nr, self.nd, pinned)

    def formod_host_buffers(self, b):
        """jur_formod_host on preallocated arrays (b.rad is read for the NaN mask, then overwritten)."""
        import time
        garr = (dp * 7)(*[_p(b.geom[k]) for k in range(7)])
        tarr = (dp * 3)(*[_p(b.tp[k]) for k in range(3)])
        args = (self.h, b.nr, garr, _p(b.rad), _p(b.tau), tarr, b.np.ctypes.data_as(C.POINTER(C.c_int)))
        t0 = time.perf_counter()
        rc = lib().jur_formod_host(*args)
        dt = time.perf_counter() - t0
        _chk(rc)
        return dt                                  # seconds inside the library call

    def curtis_godson(self, geom):
        """-> dict(cgp, cgt, cgu (nr, ng, NLOS), np)."""
        g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
        nr = g.shape[1]
        out = [np.zeros((nr, max(self.ng, 1), abi.NLOS)) for _ in range(3)]
        npts = np.zeros(nr, dtype=np.int32)
        garr = (dp * 7)(*[_p(g[k]) for k in range(7)])
        _chk(lib().jur_curtis_godson_host(self.h, nr, garr, _p(out[0]), _p(out[1]), _p(out[2]), None,
                                          npts.ctypes.data_as(C.POINTER(C.c_int))))
        return dict(cgp=out[0], cgt=out[1], cgu=out[2], np=npts)

    def formod_device(self, nr, d_geom, d_rad, d_tau, d_tp, d_np=0, d_status=0, stream=0):
        """All arguments are raw device addresses (ints), e.g. torch_tensor.data_ptr()."""
        _chk(lib().jur_formod_device(self.h, nr, d_geom, d_rad, d_tau, d_tp, d_np, d_status, stream))

    def fov_apply_device(self, nr, d_time, d_vpz, d_rad, d_tau, dz, w, stream=0):
        """Field-of-view convolution of device arrays in place (raw device addresses as for formod_device)."""
        dz, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (dz, w))
        _chk(lib().jur_fov_apply_device(self.h, nr, d_time, d_vpz, d_rad, d_tau, len(dz), _p(dz), _p(w), stream))

    def kernel(self, atm, obs):
        """Forward-difference Jacobian (m, n); obs receives the unperturbed forward model."""
        n = lib().jur_state_size(self.h, C.byref(atm))
        m = lib().jur_measurement_size(self.h, C.byref(obs))
        k = np.zeros((m, n))
        _chk(lib().jur_kernel(self.h, C.byref(atm), C.byref(obs), _p(k), m, n))
        return k

    # known-answer hooks: device functions on arrays (include/jurassic_hip.h)
    def kat_ega_eps(self, ig, id_, tau, t, u, p, mode=3, chain=False):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (tau, t, u, p)]
        out = np.zeros(len(a[0]))
        _chk(lib().jur_kat_ega_eps(self.h, ig, id_, len(out), *[_p(x) for x in a], mode, int(chain), _p(out)))
        return out

    def kat_continua(self, id_, p, t, q, u_co2, u_h2o):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (p, t, q, u_co2, u_h2o)]
        out = np.zeros((4, len(a[0])))
        _chk(lib().jur_kat_continua(self.h, id_, out.shape[1], *[_p(x) for x in a], _p(out)))
        return out

    def kat_update(self, id_, what, a, b, c, rad, tau):
        """what 0: one segment (a = T, b = tau_gas, c = beta_ds); what 1: epilogue (a = tsurf, b = bbt flag).
        -> (rad, tau, src)"""
        a, b, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (a, b, c))
        rad, tau = np.array(rad, dtype=np.float64), np.array(tau, dtype=np.float64)
        src = np.zeros(len(a))
        _chk(lib().jur_kat_update(self.h, id_, len(a), what, _p(a), _p(b), _p(c), _p(rad), _p(tau), _p(src)))
        return rad, tau, src

    def enable_timing(self, on=True):
        _chk(lib().jur_model_enable_timing(self.h, int(on)))

    def kernel_ms(self):
        ms = (C.c_double * 3)()
        n = (C.c_long * 3)()
        _chk(lib().jur_model_last_kernel_ms(self.h, ms, n))
        pm, pn = C.c_double(0), C.c_long(0)
        _chk(lib().jur_model_last_pencil_ms(self.h, C.byref(pm), C.byref(pn)))
        return dict(trace_ms=ms[0], ega_ms=ms[1], combine_ms=ms[2], trace_launches=n[0], ega_launches=n[1],
                    combine_launches=n[2], pencil_ms=pm.value, pencil_launches=pn.value)

    def workspace_bytes(self):
        return lib().jur_model_workspace_bytes(self.h)

    def last_launches(self):
        """Integration launches of the last batched call."""
        return lib().jur_model_last_launches(self.h)

    def set_compact_workspace(self, on):
        _chk(lib().jur_model_set_compact_workspace(self.h, int(on)))

    def table_bytes(self):
        return lib().jur_model_table_bytes(self.h)


ARITH_FAST, ARITH_EXACT = 0, 1


class GslMatrix(C.Structure):
    """Layout of GSL's gsl_matrix (jur_gsl_matrix_t): what the reference's kernel() takes."""
    _fields_ = [("size1", C.c_size_t), ("size2", C.c_size_t), ("tda", C.c_size_t), ("data", dp), ("block", C.c_void_p),
                ("owner", C.c_int)]


def kernel(ctl, atm, obs, m, n, tda=None):
    """Drop-in kernel() (reference jurassic.c:812): -> (m, n) Jacobian; obs receives the unperturbed forward model.
    tda > n exercises a matrix whose rows are longer than its width (a sub-matrix view)."""
    tda = tda or n
    store = np.full((m, tda), -7.0)
    mat = GslMatrix(m, n, tda, _p(store), None, 0)
    lib().kernel(C.byref(ctl), C.byref(atm), C.byref(obs), C.byref(mat))
    assert np.all(store[:, n:] == -7.0)               # nothing written beyond the matrix's width
    return store[:, :n].copy()


def device_info(device):
    """-> dict(pci_bus_id, free, total) of a HIP device (jur_device_info)."""
    buf = C.create_string_buffer(64)
    f, t = C.c_size_t(0), C.c_size_t(0)
    _chk(lib().jur_device_info(device, buf, 64, C.byref(f), C.byref(t)))
    return dict(pci_bus_id=buf.value.decode(), free=int(f.value), total=int(t.value))


def _handles(models):
    return (C.c_void_p * len(models))(*[m.h for m in models])


def multi_balance(model, geom, nparts):
    """Boundaries of nparts contiguous ray ranges with equal estimated LOS points (jur_multi_balance)."""
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
    garr = (dp * 7)(*[_p(g[k]) for k in range(7)])
    b = (C.c_long * (nparts + 1))()
    _chk(lib().jur_multi_balance(model.h, g.shape[1], garr, nparts, b))
    return list(b)


def estimate_los_points(ctl, atm, geom):
    """Estimated LOS points per ray (jur_estimate_los_points): host arithmetic, no GPU."""
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
    z = np.ctypeslib.as_array(atm.z)[:atm.np]
    garr = (dp * 7)(*[_p(g[k]) for k in range(7)])
    out = np.zeros(g.shape[1])
    _chk(lib().jur_estimate_los_points(ctl.rayds, ctl.raydz, float(z.min()), float(z.max()), g.shape[1], garr, _p(out)))
    return out


def balance_rays(ctl, atm, geom, nparts):
    """Boundaries of nparts contiguous ray ranges with equal estimated LOS points (jur_balance_rays): no GPU."""
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
    z = np.ctypeslib.as_array(atm.z)[:atm.np]
    garr = (dp * 7)(*[_p(g[k]) for k in range(7)])
    b = (C.c_long * (nparts + 1))()
    _chk(lib().jur_balance_rays(ctl.rayds, ctl.raydz, float(z.min()), float(z.max()), g.shape[1], garr, nparts, b))
    return list(b)


def models_set_atm(models, atm):
    _chk(lib().jur_models_set_atm(_handles(models), len(models), C.byref(atm)))


def formod_host_multi(models, geom, rad_in=None):
    """jur_formod_host_multi: the rays of one call dealt to several models (one per device).  Same result dict as
    Model.formod_host."""
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64).T)
    nr, nd = g.shape[1], models[0].nd
    rad = np.zeros((nr, nd)) if rad_in is None else np.ascontiguousarray(rad_in, dtype=np.float64).copy()
    tau = np.zeros((nr, nd))
    tp = np.zeros((3, nr))
    npts = np.zeros(nr, dtype=np.int32)
    garr = (dp * 7)(*[_p(g[k]) for k in range(7)])
    tarr = (dp * 3)(*[_p(tp[k]) for k in range(3)])
    _chk(lib().jur_formod_host_multi(_handles(models), len(models), nr, garr, _p(rad), _p(tau), tarr,
                                     npts.ctypes.data_as(C.POINTER(C.c_int))))
    return dict(rad=rad, tau=tau, tp=np.ascontiguousarray(tp.T), np=npts)


def formod_device_multi(models, nr, d_geom, d_rad, d_tau, d_tp, d_np=0, d_status=0, stream=0, bounds=None):
    """jur_formod_device_multi: raw device addresses on models[0]'s GPU; bounds: nmodel + 1 ray indices or None."""
    b = None if bounds is None else (C.c_long * (len(models) + 1))(*bounds)
    _chk(lib().jur_formod_device_multi(_handles(models), len(models), nr, b, d_geom, d_rad, d_tau, d_tp, d_np, d_status, stream))


def formod(ctl, atm, obs):
    """Drop-in entry (reference CPUdrivers.c:179): tables from ctl.tblbase files."""
    lib().formod(C.byref(ctl), C.byref(atm), C.byref(obs))


def dropin_finalize():
    """Free the lanes and tables behind formod() (jur_dropin_finalize) -> number of lanes freed."""
    return lib().jur_dropin_finalize()


def intpol_atm(ctl, dest, src, device=0):
    """Regrid src onto dest's points (reference intpol_atm, jurassic.c:675): fills dest.p, t, q, k."""
    _chk(lib().jur_intpol_atm(C.byref(ctl), C.byref(dest), C.byref(src), device))


def formod_fov(ctl, obs):
    """Drop-in field-of-view convolution (reference jurassic.c:214): shape file named by ctl.fov."""
    lib().formod_fov(C.byref(ctl), C.byref(obs))


def fov_read_shape(path):
    dz, w = np.zeros(abi.NSHAPE), np.zeros(abi.NSHAPE)
    n = C.c_int(0)
    _chk(lib().jur_fov_read_shape(path.encode(), C.byref(n), _p(dz), _p(w)))
    return dz[:n.value].copy(), w[:n.value].copy()


def fov_apply(time, vpz, rad, tau, dz, w):
    """Field-of-view convolution on flat arrays, in place: rad/tau (nr, nd) C-contiguous float64."""
    time, vpz, dz, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (time, vpz, dz, w))
    assert rad.flags.c_contiguous and tau.flags.c_contiguous and rad.dtype == np.float64 and rad.shape == tau.shape
    nr, nd = rad.shape
    _chk(lib().jur_fov_apply(nd, nr, _p(time), _p(vpz), _p(rad), _p(tau), nd, len(dz), _p(dz), _p(w)))


def formod_pencil(ctl, atm, obs, ir):
    lib().formod_pencil(C.byref(ctl), C.byref(atm), C.byref(obs), ir)


def tune_trace(lanes_per_ray=0):
    """Process-wide: lanes per ray of the batched ray tracer (0: chosen per launch)."""
    lib().jur_tune_trace(lanes_per_ray)


def tune_combine(channels_per_group=4, sync_segments=8, min_lanes=1_000_000):
    """Process-wide arrangement of the radiance-update kernel of the batched path (jur_tune_combine)."""
    lib().jur_tune_combine(channels_per_group, sync_segments, min_lanes)
