"""Background states for the constituent interface (`sasktran2.climatology`): the U.S. Standard Atmosphere 1976.

The reference interpolates a 21-point table of the standard (src/sasktran2/climatology/us76.py:8-116); here the standard's own defining
equations are evaluated (NOAA-S/T 76-1562, part 1: seven layers of constant lapse rate in geopotential height, hydrostatic pressure), which
agree with that table to its printed precision (0.01 K, 4 digits of pressure) below 80 km.  Above 86 km the temperature is held and the
pressure continues with its last scale height."""
from __future__ import annotations

import numpy as np

_R0 = 6356766.0          # m, effective earth radius of the standard
_G0 = 9.80665            # m / s^2
_M0 = 0.0289644          # kg / mol
_RSTAR = 8.31432         # J / (mol K), the value the standard uses
_HB = np.array([0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0, 84852.0])      # geopotential metres
_LB = np.array([-6.5e-3, 0.0, 1.0e-3, 2.8e-3, 0.0, -2.8e-3, -2.0e-3, 0.0])                 # K / m


def _base_values():
    tb, pb = [288.15], [101325.0]
    for b in range(len(_HB) - 1):
        dh = _HB[b + 1] - _HB[b]
        t1 = tb[b] + _LB[b] * dh
        if _LB[b] != 0.0:
            p1 = pb[b] * (tb[b] / t1) ** (_G0 * _M0 / (_RSTAR * _LB[b]))
        else:
            p1 = pb[b] * np.exp(-_G0 * _M0 * dh / (_RSTAR * tb[b]))
        tb.append(t1)
        pb.append(p1)
    return np.array(tb), np.array(pb)


_TB, _PB = _base_values()


class us76:
    @staticmethod
    def temperature_pressure(altitudes_m):
        """(temperature [K], pressure [Pa]) at geometric altitudes [m]"""
        z = np.asarray(altitudes_m, dtype=float)
        h = _R0 * z / (_R0 + z)
        b = np.clip(np.searchsorted(_HB, h, side="right") - 1, 0, len(_HB) - 1)
        dh = h - _HB[b]
        t = _TB[b] + _LB[b] * dh
        lapse = _LB[b] != 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            p = np.where(lapse, _PB[b] * (_TB[b] / t) ** (_G0 * _M0 / (_RSTAR * np.where(lapse, _LB[b], 1.0))),
                         _PB[b] * np.exp(-_G0 * _M0 * dh / (_RSTAR * _TB[b])))
        return t, p

    @staticmethod
    def add_us76_standard_atmosphere(atmo):
        """sets `atmo.temperature_k` and `atmo.pressure_pa` on the atmosphere's altitude grid"""
        t, p = us76.temperature_pressure(atmo.model_geometry.altitudes())
        atmo.temperature_k = t
        atmo.pressure_pa = p
