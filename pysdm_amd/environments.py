"""`Box` environment and `Mesh`: what the collision path needs to know about space.

cf. PySDM/environments/box.py:12-42 and PySDM/impl/mesh.py:9-87 (same names and meaning).
A multi-cell box is obtained the way the reference's own multi-cell collision test does it
(tests/unit_tests/dynamics/collisions/test_sdm_multi_cell.py:24-34): `env.mesh = Mesh(grid, size)`
plus an explicit "cell id" attribute.
"""
import numpy as np


class Mesh:
    def __init__(self, grid, size, n_cell=None, dv=None, n_dims=None, strides=None):
        self.grid = grid
        self.size = size
        if strides is None:
            # C-order strides in elements; a 1-D grid keeps a (1, 1) shape
            dims = tuple(int(g) for g in grid)
            flat = [int(np.prod(dims[d + 1:])) for d in range(len(dims))]
            strides = np.asarray(flat, dtype=np.int64).reshape(1, -1)
        self.strides = strides
        self.n_cell = n_cell or int(np.prod(grid))
        self.dv = dv or np.prod(np.asarray(size) / np.asarray(grid))
        self.n_dims = len(grid) if n_dims is None else n_dims

    @property
    def dz(self):
        return self.size[-1] / self.grid[-1]

    @property
    def dimension(self):
        return self.n_dims

    @property
    def dim(self):
        return self.n_dims

    @staticmethod
    def mesh_0d(dv=None):
        return Mesh(grid=(1,), size=tuple(), n_cell=1, dv=dv or np.nan, n_dims=0, strides=(-1,))

    def cellular_attributes(self, positions):
        """cell id, cell origin and position in cell from positions in grid coordinates"""
        cell_origin = positions.astype(dtype=np.int64)
        position_in_cell = positions - np.floor(positions)
        cell_id = np.empty(positions.shape[1], dtype=np.int64)
        cell_id[:] = np.dot(self.strides, cell_origin)
        return cell_id, cell_origin, position_in_cell


class Box:
    def __init__(self, dt, dv):
        self.dt = dt
        self.mesh = Mesh.mesh_0d(dv)
        self.particulator = None
        self._ambient_air = {}

    def __getitem__(self, item):
        return self._ambient_air[item]

    def __setitem__(self, key, value):
        if key not in self._ambient_air:
            self._ambient_air[key] = self.particulator.backend.Storage.from_ndarray(
                np.array([value])
            )
        else:
            self._ambient_air[key][:] = value

    def register(self, builder):
        self.particulator = builder.particulator

    def instantiate(self, *, builder):
        self.register(builder)
        return self

    def init_attributes(self, *, spectral_discretisation):
        volume, multiplicity = spectral_discretisation.sample(
            backend=self.particulator.backend, n_sd=self.particulator.n_sd
        )
        return {"volume": volume, "multiplicity": multiplicity}
