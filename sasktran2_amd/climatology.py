"""Background states for the constituent interface (`sasktran2.climatology`): the U.S. Standard Atmosphere 1976.

`us76.add_us76_standard_atmosphere` reproduces the reference (src/sasktran2/climatology/us76.py:87-116): temperature interpolated linearly
and pressure log-linearly in a 21-node table of the standard (-1 .. 80 km), clamped below the first node, temperature held and the
pressure continued with the last log-slope above the last one.  Between the coarse nodes (10 -> 15 km across the tropopause, 40 -> 50 km
across the stratopause) this differs from the standard's own equations by several kelvin / per cent, and radiances follow the table.
`us76.analytic_temperature_pressure` keeps the defining equations (NOAA-S/T 76-1562, part 1: seven layers of constant lapse rate in
geopotential height, hydrostatic pressure) for callers that want the continuous profile."""
from __future__ import annotations

import numpy as np

# the table's nodes: geometric altitude [m], temperature [deg C] and pressure [1e4 Pa] of the standard as the reference tabulates them
# (data; tests/golden/us76_nodes.json holds the same numbers transcribed by tests/golden/make_us76_golden.py)
_NODE_ALT_M = np.array([-1000.0, 0.0, 1000.0, 2000.0, 3000.0, 4000.0, 5000.0, 6000.0, 7000.0, 8000.0, 9000.0, 10000.0, 15000.0, 20000.0, 25000.0, 30000.0, 40000.0, 50000.0, 60000.0, 70000.0, 80000.0])
_NODE_T_C = np.array([21.5, 15.0, 8.5, 2.0, -4.49, -10.98, -17.47, -23.96, -30.45, -36.94, -43.42, -49.9, -56.5, -56.5, -51.6, -46.64, -22.8, -2.5, -26.13, -53.57, -74.51])
_NODE_P_1E4_PA = np.array([11.39, 10.13, 8.988, 7.95, 7.012, 6.166, 5.405, 4.722, 4.111, 3.565, 3.08, 2.65, 1.211, 0.5529, 0.2549, 0.1197, 0.0287, 0.007978, 0.002196, 0.00052, 0.00011])

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
        """(temperature [K], pressure [Pa]) at geometric altitudes [m] as the reference's `add_us76_standard_atmosphere` sets them:
        np.interp over the 21-node table (temperature; log pressure), clamped at both ends except for the pressure above 80 km, which
        continues with the log-slope of the last two nodes (us76.py:87-116)."""
        z = np.asarray(altitudes_m, dtype=float)
        t = np.interp(z, _NODE_ALT_M, _NODE_T_C + 273.15)
        logp = np.log(_NODE_P_1E4_PA * 1e4)
        lp = np.interp(z, _NODE_ALT_M, logp)
        top = z > _NODE_ALT_M[-1]
        slope = (logp[-1] - logp[-2]) / (_NODE_ALT_M[-1] - _NODE_ALT_M[-2])
        lp = np.where(top, logp[-1] + slope * (z - _NODE_ALT_M[-1]), lp)
        return t, np.exp(lp)

    @staticmethod
    def analytic_temperature_pressure(altitudes_m):
        """(temperature [K], pressure [Pa]) from the standard's defining equations (not what the reference uses between its table nodes);
        above 86 km the temperature is held and the pressure continues with its last scale height"""
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
