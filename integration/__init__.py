"""Integration of the HIP block backend with an UNMODIFIED cyten.

cyten is not importable in this repository's environments, so nothing here is imported by the product or the benches; the
modules import ``cyten`` lazily inside :func:`integration.cyten_hip.register`.  What CAN be checked without cyten is checked
in tests/test_integration_surface.py (against the text of the reference's sources when /root/reference exists) and
tests/test_gpu_array_api.py (the Array-API namespace against numpy on the device).
"""
