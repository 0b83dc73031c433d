"""disn_amd -- MI355X-native hot path of DISN's dense SDF-grid inference and
isosurface extraction.  Host-side mirror of the reference's demo surface
(reference: demo/demo.py) over the C ABI in include/disn.h.

The HIP library is mandatory: importing `disn_amd.api` without
disn_amd/libdisn_hip.so raises (there is no CPU fallback).
"""
__version__ = "0.1"
