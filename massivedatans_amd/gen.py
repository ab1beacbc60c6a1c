"""Synthetic inputs named by the benchmark configs.

These restate the RNG call ORDER of the reference generators so that the arrays are
bit-identical to what the reference scripts write (numpy legacy ``RandomState`` streams are
frozen by numpy's compatibility policy):

* :func:`horns`   <- gensimple_horns.py:15-39  (``data_widths_N.hdf5``: datasets ``x``, ``y``)
* :func:`nothing` <- gennothing.py:7-12        (``data_nothing_N.hdf5``)
* :func:`muse_like` has no generator in the reference (musefuse.py:31-154 reads a real MUSE
  cube); it is the synthetic stand-in SURVEY.md section 8(d) defines for config C5.

The reference stores ``y`` as ``[n_channels, n_datasets]`` (C order); that is what these
functions return, so they can be fed to the drop-in ``like()`` entry points unchanged.  The
device-resident layout ``[n_datasets, n_channels]`` is produced at upload time
(:class:`massivedatans_amd.like.GaussLineSpectra`).

File containers: a path ending in ``.hdf5`` / ``.h5`` is read and written through h5py exactly as
the reference does (same dataset names, gzip + shuffle), so files of the reference's generators
can be fed in directly -- where h5py is installed.  It is not in this image (nor on the GPU
box), so everything here and in the tests uses ``.npz`` with the reference's dataset names.
"""
import numpy as np

#: wavelength grid of the toy problem (gensimple_horns.py:15, gennothing.py:7)
N_CHANNELS = 200
NOISE_LEVEL = 0.01   # gensimple_horns.py:26, sample.py:45
REST_WAVE = 656      # gensimple_horns.py:21
LINE_WIDTH = 5.0     # gensimple_horns.py:23


def wavelength_grid():
    return np.linspace(400, 800, N_CHANNELS)


def horns(n):
    """``gensimple_horns.py N``: one narrow emission line per spectrum, redshifts piling up in
    two "horns".  Returns ``dict(x, y, z, mean_narrow, width_narrow, height_narrow)`` with
    ``y`` of shape ``[200, n]``.

    RNG order (gensimple_horns.py:19-39): ``seed(n)``; ``uniform(-pi, pi, n)``;
    ``power(3, n)``; then, spectrum by spectrum, ``normal(0, 0.01, 200)``.
    """
    n = int(n)
    x = wavelength_grid()
    rng = np.random.RandomState(n)          # numpy.random.seed(N) on the global stream
    z = np.arctan(rng.uniform(-np.pi, np.pi, size=n)) * 0.1
    centre = REST_WAVE * (1 + z)
    width = LINE_WIDTH * np.ones(n)
    height = 0.02 / rng.power(3, size=n)
    # noiseless lines, built [n, 200] then transposed exactly as gensimple_horns.py:8-13,31-32
    clean = height.reshape((-1, 1)) * np.exp(
        -0.5 * ((centre.reshape((-1, 1)) - x.reshape((1, -1))) / width.reshape((-1, 1))) ** 2)
    y = np.transpose(clean).copy()
    for i in range(n):                      # column-by-column draw order matters
        y[:, i] += rng.normal(0, NOISE_LEVEL, size=len(x))
    return dict(x=x, y=np.ascontiguousarray(y), z=z, mean_narrow=centre, width_narrow=width,
                height_narrow=height)


def nothing(n):
    """``gennothing.py N``: pure noise spectra (low-acceptance RadFriends stress)."""
    n = int(n)
    x = wavelength_grid()
    rng = np.random.RandomState(n)
    y = rng.normal(0, NOISE_LEVEL, size=(len(x), n))
    return dict(x=x, y=np.ascontiguousarray(y))


# --- MUSE-style synthetic cube (config C5) -------------------------------

#: rest wavelengths [Angstrom], relative amplitudes and widths of the three template lines
MUSE_LINES = ((4861.3, 0.35, 4.0), (5006.8, 1.0, 4.0), (6562.8, 0.8, 5.0))


def muse_template(x, params):
    """Three-Gaussian emission template on a flat continuum.  ``params`` = (log_amp, z,
    log_width_scale, ratio1, ratio3): ``ypred = 1 + 10**log_amp * sum_g r_g A_g
    exp(-0.5 ((x - mu_g (1+z)) / (sigma_g 10**log_width_scale))**2)`` with ``r_2 = 1``.
    Host statement of what the device template kernel evaluates
    (``mdns_muse_template_batch``)."""
    log_amp, z, log_ws, r1, r3 = params
    ratios = (r1, 1.0, r3)
    y = np.ones_like(x)
    for (mu, a, sg), r in zip(MUSE_LINES, ratios):
        y = y + (10 ** log_amp) * r * a * np.exp(-0.5 * ((x - mu * (1 + z)) / (sg * 10 ** log_ws)) ** 2)
    return y


def muse_like(n, nx=4096):
    """Synthetic IFU cube: ``n`` spaxels x ``nx`` channels.  Returns ``dict(x, y, v, z, scale)``
    with ``y`` and ``v`` (per-pixel variance) of shape ``[nx, n]``, the layout cmuselike.c:54
    indexes (``i + j*ndata``)."""
    n = int(n)
    x = np.linspace(4750, 9350, nx)
    rng = np.random.RandomState(n)
    z = rng.uniform(0.0, 0.02, size=n)
    scale = 10 ** rng.uniform(-1, 1, size=n)
    y = np.empty((nx, n))
    v = np.empty((nx, n))
    for i in range(n):
        truth = scale[i] * muse_template(x, (0.0, z[i], 0.0, 1.0, 1.0))
        v[:, i] = rng.uniform(0.5, 2.0, size=nx) * NOISE_LEVEL ** 2
        y[:, i] = truth + rng.normal(0, 1, size=nx) * np.sqrt(v[:, i])
    return dict(x=x, y=y, v=v, z=z, scale=scale)


def _is_hdf5(path):
    return str(path).endswith(('.hdf5', '.h5'))


def _h5py():
    try:
        import h5py
    except ImportError:
        raise RuntimeError("reading/writing .hdf5 needs h5py, which is not installed; "
                           "use a .npz file (gen.save / gen.load) instead")
    return h5py


def write_datasets(path, data):
    """``data``: name -> array.  ``.hdf5``: one gzip + shuffle dataset per name (scalars plain),
    as gensimple_horns.py:61-67 and sample.py:202-211 write them; otherwise a compressed .npz."""
    if _is_hdf5(path):
        with _h5py().File(path, 'w') as f:
            for k, v in data.items():
                v = np.asarray(v)
                if v.ndim == 0:
                    f.create_dataset(k, data=v)
                else:
                    f.create_dataset(k, data=v, compression='gzip', shuffle=True)
    else:
        np.savez_compressed(path, **data)


def read_datasets(path):
    if _is_hdf5(path):
        with _h5py().File(path, 'r') as f:
            return {k: np.asarray(f[k][()]) for k in f.keys()}
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


def save(path, data):
    write_datasets(path, data)


def load(path, ndata=None):
    """Counterpart of sample.py:28-31: ``x`` and the first ``ndata`` columns of ``y``."""
    out = read_datasets(path)
    if ndata is not None:
        for k in ("y", "v"):
            if k in out:
                out[k] = np.ascontiguousarray(out[k][:, :int(ndata)])
    return out
