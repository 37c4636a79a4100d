"""Field-like proxies over device buffers: the slice of the Taichi field API that the
reference's caller uses (main.py:111,159,173,190): `.to_numpy()`, `.from_numpy()`, `[None]`."""
import numpy as np

from . import _native as nat


class DeviceField:
    """A per-particle field living in HBM (cell-sorted); crosses the ABI in original particle order."""

    def __init__(self, owner, field, species=nat.SPECIES_FLUID, writable=False):
        self._owner, self._field, self._species, self._writable = owner, field, species, writable

    @property
    def shape(self):
        sim = self._owner._sim
        if self._species == nat.SPECIES_RIGID:
            return (sim.n_vertices if self._field == nat.F_RIGID_VERT else sim.n_rigid,)
        return (sim.n_wall if self._species == nat.SPECIES_WALL else sim.n_fluid,)

    def to_numpy(self):
        return self._owner._sim.download(self._field, self._species)

    def from_numpy(self, arr):
        if not self._writable:
            raise TypeError("this field is read-only on the device")
        self._owner._sim.upload(self._field, arr, self._species)

    def __getitem__(self, i):
        return self.to_numpy()[i]

    def __len__(self):
        return self.shape[0]


class ConstField:
    """Host-resident constant-per-particle field (colours)."""

    def __init__(self, n, value):
        self._n, self._value = n, np.asarray(value, dtype=np.float32)

    def to_numpy(self):
        return np.broadcast_to(self._value, (self._n,) + self._value.shape).copy()

    def fill(self, value):
        self._value = np.asarray(value, dtype=np.float32)

    def __getitem__(self, i):
        return self._value.copy()


class ScalarField:
    """0-d field: value = f[None]."""

    def __init__(self, getter, setter=None):
        self._get, self._set = getter, setter

    def __getitem__(self, key):
        if key is not None:
            raise IndexError("0-d field: index with [None]")
        return self._get()

    def __setitem__(self, key, value):
        if key is not None or self._set is None:
            raise TypeError("read-only 0-d field")
        self._set(value)


class ParticleFields:
    """`ps.fluid_particles` / `ps.boundary_particles`: attribute access to per-particle fields."""

    def __init__(self, **fields):
        self.__dict__.update(fields)
