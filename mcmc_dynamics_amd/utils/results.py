"""Small three-row results table (median / uperr / loerr per parameter) standing in for the
``astropy.table.QTable`` with a ``value`` index that the reference returns from its chain statistics
(analysis/runner.py:625-660; utils/coordinates/get_amplitude_and_angle.py:33-46)."""
from collections import OrderedDict

import numpy as np

ROWS = ("median", "uperr", "loerr")


class _Row(object):
    def __init__(self, table, i):
        self._table, self._i = table, i

    def __getitem__(self, name):
        return self._table._cols[name][self._i]

    def __setitem__(self, name, value):
        self._table._cols[name][self._i] = value


class _Loc(object):
    def __init__(self, table):
        self._table = table

    def __getitem__(self, row):
        return _Row(self._table, ROWS.index(row))


class ResultsTable(object):
    def __init__(self):
        self._cols = OrderedDict()
        self._cols["value"] = np.array(ROWS, dtype=object)
        self.units = {}

    def add_column(self, name, median, uperr, loerr, unit=None):
        self._cols[name] = np.array([median, uperr, loerr], dtype=np.float64)
        self.units[name] = unit

    @property
    def columns(self):
        return list(self._cols.keys())

    colnames = columns

    @property
    def loc(self):
        return _Loc(self)

    def __getitem__(self, name):
        return self._cols[name]

    def __setitem__(self, name, values):
        self._cols[name] = np.asarray(values, dtype=np.float64)

    def __repr__(self):
        names = [c for c in self._cols if c != "value"]
        lines = ["{0:>8s} ".format("value") + " ".join("{0:>14s}".format(n) for n in names)]
        for i, row in enumerate(ROWS):
            lines.append("{0:>8s} ".format(row) + " ".join("{0:14.6g}".format(self._cols[n][i]) for n in names))
        return "\n".join(lines)
