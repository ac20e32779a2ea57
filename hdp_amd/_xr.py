"""Pick the labelled-array backend: real xarray when importable, else hdp_amd.minixr."""


def backend():
    try:
        import xarray  # noqa: F401
        return xarray
    except Exception:  # not installed in the build/test images
        from . import minixr
        return minixr


def jan1_stamps(years, template_date):
    """Jan-1 timestamps of each year in the calendar of `template_date`
    (reference: cftime.datetime + xarray.date_range, metric.py:463-465)."""
    try:
        import cftime
        cal = template_date.calendar
        return [cftime.datetime(int(y), 1, 1, calendar=cal) for y in years]
    except Exception:
        cls = type(template_date)
        try:
            return [cls(int(y), 1, 1) for y in years]
        except Exception:
            from .utils import NoLeapDate
            return [NoLeapDate(int(y), 1, 1) for y in years]
