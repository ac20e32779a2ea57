"""Shared test helpers: build HDP-formatted measure Datasets the way
hdp.measure.format_standard_measures stamps them (measure.py:166-173: float32 cast,
attrs hdp_type / input_variable / baseline_variable)."""
import numpy as np

from hdp_amd._xr import backend


def measure_dataset(values, lon, lat, dates, name="temp", dims=("lon", "lat", "time"), extra_coords=None):
    xr = backend()
    coords = {"lon": lon, "lat": lat, "time": dates}
    coords.update(extra_coords or {})
    coords = {k: v for k, v in coords.items() if k in dims}
    da = xr.DataArray(np.asarray(values, dtype=np.float32), dims=list(dims), coords=coords, name=name,
                      attrs={"units": "degC", "hdp_type": "measure", "input_variable": name,
                             "baseline_variable": name})
    return xr.Dataset({name: da})
