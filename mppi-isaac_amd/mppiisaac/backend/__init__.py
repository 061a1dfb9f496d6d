"""ctypes binding of the C-ABI (include/mppi_hip.h) and the URDF -> packed-model compiler."""
