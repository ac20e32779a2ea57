"""Drop-in for ``hdp.measure`` (the step upstream of the hot path): same
function names, arguments, variable names, units handling and attrs; the heat-index ufunc runs in
a HIP kernel (``hdp_heat_index_f32``), the unit conversions stay float32 numpy as in the reference.

  heat_index                 <- hdp/measure.py:61-94   (array-level, GPU)
  apply_heat_index           <- hdp/measure.py:108-132
  convert_temp_units, kelvin_to_celsius, fahrenheit_to_celsius, celsius_to_fahrenheit
                             <- hdp/measure.py:10-58,135-149
  format_standard_measures   <- hdp/measure.py:152-203
"""
from __future__ import annotations

import numpy as np

from . import core
from ._xr import backend
from .utils import add_history, get_version

TEMPERATURE_UNITS = ['degC', 'degK', 'degF', 'C', 'K', 'F']
HUMIDITY_UNITS = ["%", "g/g"]


def _like(da, values, name=None, attrs=None):
    xr = backend()
    coords = {k: np.asarray(da.coords[k].values) for k in da.coords if k in da.dims}
    return xr.DataArray(values, dims=list(da.dims), coords=coords, name=da.name if name is None else name,
                        attrs=dict(da.attrs if attrs is None else attrs))


def kelvin_to_celsius(temp):
    """measure.py:10-24 (float32 array minus a Python float stays float32)."""
    out = _like(temp, np.asarray(temp.values) - np.asarray(273.15, dtype=temp.values.dtype))
    out.attrs["units"] = "degC"
    return add_history(out, "HDP converted units from Kelvin to Celsius.")


def fahrenheit_to_celsius(temp):
    """measure.py:27-41."""
    v = np.asarray(temp.values)
    out = _like(temp, (v - np.asarray(32, dtype=v.dtype)) / np.asarray(1.8, dtype=v.dtype))
    out.attrs["units"] = "degC"
    return add_history(out, "HDP converted units from Fahrenheit to Celsius.")


def celsius_to_fahrenheit(temp):
    """measure.py:44-58."""
    v = np.asarray(temp.values)
    out = _like(temp, (v * np.asarray(1.8, dtype=v.dtype)) + np.asarray(32, dtype=v.dtype))
    out.attrs["units"] = "degF"
    return add_history(out, "HDP converted units from Celsius to Fahrenheit.")


def heat_index(temp, rel_humid):
    """NWS heat-index regression (measure.py:61-94), element-wise on the GPU.
    temp in degrees Fahrenheit, rel_humid in percent; float32 in, float32 out."""
    return core.heat_index(temp, rel_humid)


def apply_heat_index(temp, rh):
    """measure.py:108-132: heat index DataArray named ``{temp.name}_hi`` (degrees Fahrenheit).
    As in the reference (apply_ufunc without keep_attrs) the input attrs are not carried over."""
    assert temp.attrs["units"] == "degF"
    assert rh.attrs["units"] == "%"
    t = np.asarray(temp.values, dtype=np.float32)
    r = np.asarray(rh.values, dtype=np.float32)
    if tuple(rh.dims) != tuple(temp.dims):  # align by name, as xarray broadcasting would
        r = np.transpose(r, [list(rh.dims).index(d) for d in temp.dims])
    name = f"{temp.name}_hi"
    hi = _like(temp, heat_index(t, r), name=name, attrs={})
    hi.attrs["baseline_variable"] = name
    return add_history(hi, f"Converted to heat index using '{rh.name}' relative humidity, "
                           f"renamed from '{temp.name}' to '{name}'.")


def convert_temp_units(temp_ds):
    """measure.py:135-149."""
    if temp_ds.attrs["units"] in ("K", "degK"):
        temp_ds = kelvin_to_celsius(temp_ds)
    elif temp_ds.attrs["units"] in ("F", "degF"):
        temp_ds = fahrenheit_to_celsius(temp_ds)
    return temp_ds


def format_standard_measures(temp_datasets, rh=None):
    """measure.py:152-203: float32 cast, hdp attrs, unit conversion to Celsius and, when ``rh`` is
    given, one heat-index measure per temperature measure; returns the merged Dataset."""
    xr = backend()
    measures = []
    for temp_ds in temp_datasets:
        assert "units" in temp_ds.attrs, f"Attribute 'units' not found in '{temp_ds.name}' dataset."
        assert temp_ds.attrs["units"] in TEMPERATURE_UNITS, \
            f"Units for '{temp_ds.name}' must be one of the following: {TEMPERATURE_UNITS}"
        t = _like(temp_ds, np.array(temp_ds.values, dtype=np.float32))
        t.attrs.update({"hdp_type": "measure", "input_variable": temp_ds.name, "baseline_variable": temp_ds.name})
        measures.append(convert_temp_units(t))

    if rh is not None:
        assert "units" in rh.attrs, "Attribute 'units' not found in rh dataset."
        assert rh.attrs["units"] in HUMIDITY_UNITS, f"Units for rh must be one of the following: {HUMIDITY_UNITS}"
        rh = _like(rh, np.array(rh.values, dtype=np.float32))
        if rh.attrs["units"] == "g/g":
            rh = _like(rh, np.asarray(rh.values) * np.float32(100))
            rh.attrs["units"] = "%"
        for measure in list(measures):
            hi_f = apply_heat_index(celsius_to_fahrenheit(measure), rh)
            hi_f.attrs["units"] = "degF"
            measures.append(fahrenheit_to_celsius(hi_f))

    agg = xr.merge([xr.Dataset({m.name: m}) for m in measures])
    agg.attrs = {
        "description": f"Heat measurement dataset generated by Heatwave Diagnostics Package (HDP v{get_version()})",
        "hdp_version": get_version(),
    }
    return add_history(agg, f"Dataset aggregated by HDP with measures: {[m.name for m in measures]}")
