"""Catalogue container: named float64 columns with optional units (reference: utils/files/data_reader.py,
an ``astropy.table.QTable`` wrapper).  Columns the hot path uses: ``ra, dec`` [deg], ``v, verr`` [km/s],
optional ``density``, ``pmember`` (dimensionless) and ``bin`` (int16, from ``make_radial_bins``)."""
import logging
from collections import OrderedDict

import numpy as np

from .. import units
from .coordinates import calc_xy_offset

logger = logging.getLogger(__name__)


class ColumnTable(object):
    """Minimal struct-of-arrays table: ``t['v']``, ``t.columns``, ``len(t)``, ``t[mask]``."""

    def __init__(self, data=None, units_=None):
        self._cols = OrderedDict()
        self.units = dict(units_ or {})
        if data is None:
            return
        if isinstance(data, ColumnTable):
            for k in data.columns:
                self._cols[k] = np.array(data[k])
            self.units.update(data.units)
            return
        if hasattr(data, "colnames"):                       # astropy Table / QTable
            data = {k: data[k] for k in data.colnames}
        if isinstance(data, np.ndarray) and data.dtype.names:
            data = {k: data[k] for k in data.dtype.names}
        for key, col in dict(data).items():
            self[key] = col

    @property
    def columns(self):
        return list(self._cols.keys())

    colnames = columns

    def __len__(self):
        return len(next(iter(self._cols.values()))) if self._cols else 0

    def __contains__(self, key):
        return key in self._cols

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._cols[key]
        out = ColumnTable()
        for k, col in self._cols.items():
            out._cols[k] = col[key]
        out.units = dict(self.units)
        return out

    def __setitem__(self, key, col):
        plain, unit = units.split(col)
        arr = np.array(plain)
        if arr.dtype.kind == "f" or arr.dtype.kind in "iu" and key != "bin":
            arr = arr.astype(np.float64)
        arr = np.atleast_1d(arr)
        if self._cols and len(arr) != len(self):
            raise ValueError("column '{0}' has length {1}, table has {2}".format(key, len(arr), len(self)))
        self._cols[key] = arr
        if unit is not None:
            self.units[key] = unit

    def unit(self, key):
        return self.units.get(key)

    def __repr__(self):
        return "<ColumnTable {0} rows: {1}>".format(len(self), ", ".join(self.columns))


class DataReader(object):

    def __init__(self, data, **kwargs):
        """``data``: dict of columns (arrays or Quantities), structured array, astropy table or another
        ColumnTable.  Extra keyword arguments are accepted for signature compatibility and ignored."""
        self.data = ColumnTable(data)

    @property
    def sample_size(self):
        return len(self.data)

    @property
    def has_ra(self):
        return "ra" in self.data.columns

    @property
    def has_dec(self):
        return "dec" in self.data.columns

    @property
    def has_coordinates(self):
        return self.has_ra & self.has_dec

    def column(self, name, unit=None):
        """Plain float64 column expressed in ``unit`` (bare columns are assumed to be in it already)."""
        col = self.data[name]
        src = self.data.unit(name)
        if unit is None or src is None:
            return np.asarray(col, dtype=np.float64)
        return np.asarray(col, dtype=np.float64) * units.conversion_factor(src, unit)

    def compute_distances(self, ra_center, dec_center):
        """Distances [arcmin] of the data points from a reference position (data_reader.py:47-69)."""
        if not self.has_coordinates:
            logger.error("Cannot calculate distances as world coordinates are missing.")
            return None
        x, y = calc_xy_offset(self.column("ra", "deg"), self.column("dec", "deg"), ra_center, dec_center)
        return np.sqrt(x ** 2 + y ** 2)

    def make_radial_bins(self, ra_center, dec_center, nstars=50, dlogr=0.2):
        """Greedy radial bins of at least ``nstars`` stars and ``dlogr`` dex; a short tail is merged into
        the last bin (data_reader.py:71-120).  Writes the int16 column ``bin``."""
        if not self.has_coordinates:
            logger.error("Cannot create radial profile. WCS coordinates of data points unknown.")
            return
        r = self.compute_distances(ra_center, dec_center)
        n = self.sample_size
        order = np.argsort(r)
        log_r = np.log10(r[order])
        bin_sorted = -np.ones(n, dtype=np.int16)
        current = -1
        i = 0
        while i < (n - nstars):
            j = min(n, i + nstars)
            while (log_r[j] - log_r[i]) < dlogr:
                j += 1
                if j >= n:
                    break
            current += 1
            bin_sorted[i:j] = current
            i = j
        if (n - i) > 0.5 * nstars or current == -1:
            current += 1
        bin_sorted[i:] = current
        bins = np.empty(n, dtype=np.int16)
        bins[order] = bin_sorted
        self.data["bin"] = bins

    def fetch_radial_bin(self, i):
        """Sub-catalogue of radial bin ``i`` (data_reader.py:122-140)."""
        if "bin" not in self.data.columns:
            logger.error("No information about bins available.")
            return None
        bins = self.data["bin"]
        if i < bins.min() or i > bins.max():
            logger.error("Requested bin %s does not exist.", i)
            return None
        return self.__class__(self.data[bins == i])

    def sorted_by_bin(self):
        """(DataReader sorted by bin, offsets[B + 1]): the layout the binned kernel launch expects."""
        bins = np.asarray(self.data["bin"], dtype=np.int64)
        order = np.argsort(bins, kind="stable")
        counts = np.bincount(bins, minlength=int(bins.max()) + 1)
        return self.__class__(self.data[order]), np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
