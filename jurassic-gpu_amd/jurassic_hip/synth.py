"""Seeded synthetic inputs: emissivity tables, filter functions, observation
geometries and stacked atmosphere profiles.

The geometry formulas restate the reference's generators: limb tangent-height
scan src/limb.c:49-59, nadir latitude sweep src/nadir.c:51-58; the profile
perturbation follows src/climatology.c:65-78 (p*(1+dp), T+dT per profile).
"""
import numpy as np
from . import abi

RE = 6367.421

# absorption strength per molecule/cm^2 at 1000 hPa, 250 K, by emitter
K0 = {"CO2": 5e-23, "H2O": 2e-22, "O3": 5e-21, "F11": 1e-17, "CCL4": 2e-17}


def table_rows(emitter, nu, id_=0, nlev=33, ntemp=10, descending=False, umax_eps=0.99999, ratio=1.122, dup_every=0):
    """Rows (p, T, u, eps) of one synthetic table in file order.

    p: nlev levels 0.016..1000 hPa (log-spaced, ascending unless `descending`);
    T: ntemp values per level, 15 K apart, the axis shifted by 2 K*(level%3) so
       neighbouring levels do not share their temperature brackets;
    u: geometric grid with the given ratio from eps~1e-6 to eps>umax_eps;
    eps = 1 - exp(-(k u)^0.7), k = k0 sqrt(p/1000) 250/T (1+0.3 id_) (nu/800)^2;
    dup_every = k > 0: after every k-th row a row 1e-10 (relative) above it -- larger as a double, so the
       loader keeps it, equal once stored as fp32: curves that are sorted but not strictly increasing."""
    k0 = K0.get(emitter.upper(), 1e-21) * (1 + 0.3 * id_) * (nu / 800.0) ** 2
    plev = np.exp(np.linspace(np.log(0.016), np.log(1000.0), nlev))
    if descending:
        plev = plev[::-1]
    nmax = 2 + int(np.ceil(np.log(1e12) / np.log(ratio))) if ratio > 1 else 400
    steps = np.full(nmax, ratio)
    blocks = []
    for il, p in enumerate(plev):
        for t in 180.0 + 2.0 * (il % 3) + 15.0 * np.arange(ntemp):
            k = k0 * np.sqrt(p / 1000.0) * 250.0 / t
            steps[0] = (1e-6 ** (1 / 0.7)) / k
            u = np.cumprod(steps)                      # u0, u0*r, (u0*r)*r, ... left to right
            eps = 1.0 - np.exp(-(k * u) ** 0.7)
            hit = np.nonzero(eps > umax_eps)[0]
            n = (hit[0] + 1) if len(hit) else nmax     # the row that crosses the limit is the last one
            blk = np.empty((n, 4))
            blk[:, 0] = p
            blk[:, 1] = t
            blk[:, 2] = u[:n]
            blk[:, 3] = eps[:n]
            if dup_every > 0:
                twin = blk[dup_every - 1::dup_every].copy()
                twin[:, 2:] *= 1.0 + 1e-10
                blk = np.insert(blk, np.arange(dup_every, n + 1, dup_every)[:len(twin)], twin, axis=0)
            blocks.append(blk)
    return np.vstack(blocks)


def write_table_file(path, rows):
    with open(path, "w") as fh:
        fh.write("# $1 = pressure [hPa]\n# $2 = temperature [K]\n"
                 "# $3 = column density [molecules/cm^2]\n# $4 = emissivity\n\n")
        for p, t, u, e in rows:
            fh.write("%.9g %.9g %.9g %.9g\n" % (p, t, u, e))


def parse_table_file(path):
    out = []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if len(tok) >= 4:
                try:
                    out.append([float(x) for x in tok[:4]])
                except ValueError:
                    pass
    return np.array(out)


def boxcar_filter(nu, halfwidth=0.5, n=21):
    x = np.linspace(nu - halfwidth, nu + halfwidth, n)
    f = np.ones(n)
    f[0] = f[-1] = 0.0
    return x, f


def write_filter_file(path, x, f):
    with open(path, "w") as fh:
        fh.write("# $1 = wavenumber [cm^-1]\n# $2 = filter function\n\n")
        for a, b in zip(x, f):
            fh.write("%.4f %g\n" % (a, b))


def limb_geometry(nr, seed=0, obsz=780.0, zmin=3.0, zmax=68.0, nprofiles=1, scan=False):
    """(nr, 7) limb rays.  scan=True gives the reference's regular tangent-height
    scan (limb.c), else vpz ~ U[zmin, zmax]."""
    rng = np.random.default_rng(seed)
    vpz = np.linspace(zmin, zmax, nr) if scan else rng.uniform(zmin, zmax, nr)
    g = np.zeros((nr, 7))
    g[:, 0] = np.arange(nr) % nprofiles
    g[:, 1] = obsz
    g[:, 4] = vpz
    g[:, 6] = 180.0 / np.pi * np.arccos((RE + vpz) / (RE + obsz))
    return g


def nadir_geometry(nr, seed=0, obsz=700.0, lat0=-8.01, lat1=8.01, nprofiles=1):
    rng = np.random.default_rng(seed)
    g = np.zeros((nr, 7))
    g[:, 0] = np.arange(nr) % nprofiles
    g[:, 1] = obsz
    g[:, 6] = rng.uniform(lat0, lat1, nr)
    return g


# ---- index-addressable geometry ("splitmix64 seed 0x4A55524153534943") -----------------------------
# Ray i of a workload is a function of (seed, i) alone: output i of the splitmix64 stream started at `seed`, whose
# state after i + 1 steps is seed + (i + 1) * golden -- no sequential generator state.  A rank of a sharded run builds
# exactly its rows [lo, hi), rank 0 builds the handful of sampled global rows it re-computes, nobody builds the whole set.
SURVEY_SEED = 0x4A55524153534943
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_uniform(seed, idx):
    """U[0, 1) doubles number idx (array of non-negative ints) of the splitmix64 stream with the given seed."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def limb_rays(idx, seed=SURVEY_SEED, obsz=780.0, zmin=3.0, zmax=68.0, nprofiles=1):
    """(len(idx), 7) limb rays number idx of the workload: view-point altitude ~ U[zmin, zmax] from obsz km, looking
    at the tangent point (limb.c:49-59), profile idx mod nprofiles."""
    idx = np.asarray(idx, dtype=np.int64)
    vpz = zmin + (zmax - zmin) * splitmix64_uniform(seed, idx)
    g = np.zeros((len(idx), 7))
    g[:, 0] = idx % nprofiles
    g[:, 1] = obsz
    g[:, 4] = vpz
    g[:, 6] = 180.0 / np.pi * np.arccos((RE + vpz) / (RE + obsz))
    return g


def nadir_rays(idx, seed=SURVEY_SEED, obsz=700.0, lat0=-8.01, lat1=8.01, nprofiles=1):
    """(len(idx), 7) nadir observations number idx: sub-satellite latitude ~ U[lat0, lat1] (nadir.c:51-58)."""
    idx = np.asarray(idx, dtype=np.int64)
    g = np.zeros((len(idx), 7))
    g[:, 0] = idx % nprofiles
    g[:, 1] = obsz
    g[:, 6] = lat0 + (lat1 - lat0) * splitmix64_uniform(seed, idx)
    return g


def stack_profiles(atm, ctl, nprofiles, seed=0, dp=0.05, dt=30.0):
    """Atmosphere holding `nprofiles` perturbed copies of the first profile of
    `atm`, time stamps 0..nprofiles-1 (profile 0 unperturbed)."""
    rng = np.random.default_rng(seed)
    n = atm.np
    assert n * nprofiles <= abi.NP
    out = abi.atm_t()
    out.np = n * nprofiles
    fields = ("z", "lon", "lat", "p", "t")
    src = {f: np.ctypeslib.as_array(getattr(atm, f))[:n].copy() for f in fields}
    q = np.ctypeslib.as_array(atm.q)[:, :n].copy()
    k = np.ctypeslib.as_array(atm.k)[:, :n].copy()
    for i in range(nprofiles):
        s = slice(i * n, (i + 1) * n)
        fp = 1.0 + (rng.uniform(-dp, dp) if i else 0.0)
        ft = rng.uniform(-dt, dt) if i else 0.0
        np.ctypeslib.as_array(out.time)[s] = float(i)
        for f in ("z", "lon", "lat"):
            np.ctypeslib.as_array(getattr(out, f))[s] = src[f]
        np.ctypeslib.as_array(out.p)[s] = src["p"] * fp
        np.ctypeslib.as_array(out.t)[s] = src["t"] + ft
        np.ctypeslib.as_array(out.q)[:, s] = q
        np.ctypeslib.as_array(out.k)[:, s] = k
    return out
