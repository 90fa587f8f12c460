"""Minimal unit handling for the host side.

The reference wraps every number in ``astropy.units.Quantity`` (L0 of SURVEY.md section 1); astropy is
not a dependency of this package and is absent on the GPU box.  The hot path works on plain float64
in the reference's canonical units, so all that is needed here is (a) a name for each unit and
(b) conversion of user input into the canonical unit.  astropy ``Quantity``/``Unit`` objects are
accepted by duck typing (``.to_value`` / ``.to_string``) when a user has astropy installed.
"""
import numpy as np

_ANGLE = {"deg": np.pi / 180.0, "rad": 1.0, "arcmin": np.pi / 180.0 / 60.0, "arcsec": np.pi / 180.0 / 3600.0,
          "mas": np.pi / 180.0 / 3.6e6}
_SPEED = {"km/s": 1.0e3, "m/s": 1.0, "cm/s": 1.0e-2}
_FAMILIES = (_ANGLE, _SPEED)
_ALIASES = {"km / s": "km/s", "m / s": "m/s", "cm / s": "cm/s", "degree": "deg", "radian": "rad",
            "": None, "dimensionless": None, "None": None}


def unit_name(unit):
    """Normalise a unit given as None, str or an astropy unit to a canonical string (or None)."""
    if unit is None:
        return None
    if not isinstance(unit, str):
        to_string = getattr(unit, "to_string", None)
        unit = to_string() if to_string is not None else str(unit)
    unit = unit.strip()
    unit = _ALIASES.get(unit, unit)
    if unit is None:
        return None
    compact = unit.replace(" ", "")
    for fam in _FAMILIES:
        if compact in fam:
            return compact
    return unit


def conversion_factor(src, dst):
    """Multiplicative factor taking values in ``src`` to ``dst``; both unit names or None."""
    src, dst = unit_name(src), unit_name(dst)
    if src == dst or src is None or dst is None:
        return 1.0
    for fam in _FAMILIES:
        if src in fam and dst in fam:
            return fam[src] / fam[dst]
    raise ValueError("cannot convert unit '{0}' to '{1}'".format(src, dst))


def split(value):
    """Return (plain ndarray/float, unit name or None) for a number, array or Quantity-like object."""
    unit = getattr(value, "unit", None)
    if unit is not None and hasattr(value, "value"):
        return np.asarray(value.value, dtype=np.float64), unit_name(unit)
    return value, None


def to_unit(value, unit, default_unit=None):
    """Plain float64 value(s) of ``value`` expressed in ``unit``.

    Bare numbers are taken to be in ``default_unit`` (or already in ``unit``), mirroring the
    reference's "Missing units ... Assuming" behaviour (analysis/runner.py:78-80)."""
    plain, src = split(value)
    if src is None:
        src = default_unit if default_unit is not None else unit
    arr = np.asarray(plain, dtype=np.float64)
    f = conversion_factor(src, unit)
    return arr if f == 1.0 else arr * f
