"""The reference's `methods` package surface (methods/oei.py, methods/sdp.py, methods/bo.py),
served by the HIP library.  Names, argument meaning and error behaviour follow the reference."""
